"""Host-side signal objects: they only PRODUCE the real coefficient table ``S[B][R][k]`` that is
shipped to the device; no device code here.

Mirrors the evaluation surface of the reference's ``signals/signals.py`` (``Signal`` :34-155,
``DiscreteSignal`` :257-311, ``SignalSum`` :505-577, ``SignalList`` :780-803): a signal is
``Re[f(t) exp(i(2 pi nu t + phi))]`` with an array-vectorised complex envelope ``f``.  The
arithmetic ORDER of the reference is kept (complex carrier argument, one ``exp``, sum over terms,
real part) so that tables are bit-identical to ``SignalList(...)(t)`` of the reference.
``+``, ``-``, unary ``-``, ``*`` (the two-term product rule of the reference, :838-1000),
``conjugate``, ``DiscreteSignal.from_Signal`` / ``add_samples``, ``DiscreteSignalSum`` (:612-777) and
``flatten`` are provided so that signals can be built the way the reference's users build them; the composite TYPE returned by a product may differ
from the reference's (always a ``SignalSum`` of plain ``Signal``s here), its VALUES do not.
RWA and transfer functions are out of scope for this path.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Union

import numpy as np

from ._lib import DynamicsError


class Signal:
    """``Re[f(t) e^{i(2 pi nu t + phi)}]`` with envelope ``f`` (callable or constant)."""

    def __init__(self, envelope: Union[Callable, complex, float], carrier_freq=0.0, phase=0.0,
                 name: Optional[str] = None):
        self._name = name
        self._is_constant = False
        if not callable(envelope):
            const = np.asarray(envelope)
            if np.ndim(carrier_freq) == 0 and carrier_freq == 0.0:
                self._is_constant = True
            self._envelope = lambda t: const * np.ones_like(t)
        else:
            self._envelope = envelope
        self.carrier_freq = carrier_freq
        self.phase = phase

    @property
    def name(self):
        return self._name

    @property
    def is_constant(self) -> bool:
        return self._is_constant

    @property
    def carrier_freq(self):
        return self._carrier_freq

    @carrier_freq.setter
    def carrier_freq(self, value):
        self._carrier_freq = np.asarray(value)
        self._carrier_arg = 1j * 2 * np.pi * self._carrier_freq

    @property
    def phase(self):
        return self._phase

    @phase.setter
    def phase(self, value):
        self._phase = np.asarray(value)
        self._phase_arg = 1j * self._phase

    def envelope(self, t):
        return self._envelope(t)

    def complex_value(self, t):
        return self.envelope(t) * np.exp(self._carrier_arg * t + self._phase_arg)

    def __call__(self, t):
        return np.real(self.complex_value(t))

    def __add__(self, other):
        return SignalSum(self, other)

    def __radd__(self, other):
        return SignalSum(other, self)

    def __neg__(self):
        return signal_multiply(-1.0, self)

    def __sub__(self, other):
        return SignalSum(self, signal_multiply(-1.0, other))

    def __rsub__(self, other):
        return SignalSum(other, signal_multiply(-1.0, self))

    def __mul__(self, other):
        return signal_multiply(self, other)

    def __rmul__(self, other):
        return signal_multiply(other, self)

    def conjugate(self):
        """Signal whose complex value is the conjugate of this one's."""
        return Signal(lambda t: np.conjugate(self.envelope(t)), -self.carrier_freq, -self.phase)

    def __str__(self):
        if self._name is not None:
            return str(self._name)
        if self._is_constant:
            return f"Constant({self(0.0)})"
        return f"Signal(carrier_freq={self.carrier_freq}, phase={self.phase})"


class DiscreteSignal(Signal):
    """Piecewise-constant envelope ``samples[floor((t - t0)/dt)]``, zero outside the sample window
    (the clipped indices -1 and len hit a zero pad)."""

    def __init__(self, dt: float, samples, start_time: float = 0.0, carrier_freq=0.0, phase=0.0,
                 name: Optional[str] = None):
        self._dt = dt
        samples = np.asarray(samples)
        if len(samples) == 0:
            pad = np.asarray([0])
        else:
            pad = np.expand_dims(np.zeros_like(samples[0]), 0)
        self._padded_samples = np.append(samples, pad, axis=0)
        self._start_time = start_time

        def envelope(t):
            t = np.asarray(t)
            idx = np.clip(np.array((t - self._start_time) // self._dt, dtype=int), -1,
                          len(self.samples))
            return self._padded_samples[idx]

        Signal.__init__(self, envelope=envelope, carrier_freq=carrier_freq, phase=phase, name=name)

    @classmethod
    def from_Signal(cls, signal: Signal, dt: float, n_samples: int, start_time: float = 0.0,
                    sample_carrier: bool = False):
        """Sample ``signal`` at the bin mid-points; with ``sample_carrier`` the carrier is folded
        into the samples (carrier frequency 0, phase kept)."""
        times = start_time + (np.arange(n_samples) + 0.5) * dt
        if sample_carrier:
            return cls(dt, signal(times), start_time=start_time, carrier_freq=0.0, phase=signal.phase,
                       name=signal.name)
        return cls(dt, signal.envelope(times), start_time=start_time, carrier_freq=signal.carrier_freq,
                   phase=signal.phase, name=signal.name)

    def conjugate(self):
        return DiscreteSignal(self._dt, np.conjugate(self.samples), start_time=self._start_time,
                              carrier_freq=-self.carrier_freq, phase=-self.phase)

    def add_samples(self, start_sample: int, samples):
        """Append ``samples`` starting at index ``start_sample``; a gap is filled with zeros
        (signals/signals.py:411-439)."""
        samples = np.asarray(samples)
        if len(samples) < 1:
            return
        if start_sample < len(self.samples):
            raise DynamicsError("Samples can only be added afer the last sample.")
        zero_pad = np.expand_dims(np.zeros_like(samples[0]), 0)
        new_samples = self.samples
        if len(self.samples) < start_sample:
            new_samples = np.append(new_samples, np.repeat(zero_pad, start_sample - len(self.samples)))
        new_samples = np.append(new_samples, samples)
        self._padded_samples = np.append(new_samples, zero_pad, axis=0)

    @property
    def dt(self):
        return self._dt

    @property
    def samples(self):
        return self._padded_samples[:-1]

    @property
    def start_time(self):
        return self._start_time

    @property
    def duration(self):
        return len(self.samples)


class SignalSum(Signal):
    """Sum of signals; ``complex_value`` sums the complex values of the terms."""

    def __init__(self, *signals, name: Optional[str] = None):
        components: List[Signal] = []
        for sig in signals:
            if isinstance(sig, list):
                sig = SignalSum(*sig)
            if isinstance(sig, SignalSum):
                components += sig.components
            elif isinstance(sig, Signal):
                components.append(sig)
            elif np.asarray(sig).ndim == 0 and np.asarray(sig).dtype.kind in "iufc":
                components.append(Signal(sig))
            else:
                raise DynamicsError(
                    "Components of a SignalSum must be instances of a Signal subclass or a scalar.")
        self._components = components

        def envelope(t):
            return np.moveaxis(np.asarray([s.envelope(t) for s in self._components]), 0, -1)

        super().__init__(envelope=envelope,
                         carrier_freq=[s.carrier_freq for s in components],
                         phase=[s.phase for s in components], name=name)

    @property
    def components(self):
        return self._components

    def __len__(self):
        return len(self._components)

    def __getitem__(self, idx):
        return self._components[idx]

    def __iter__(self):
        return iter(self._components)

    def complex_value(self, t):
        exp_phases = np.exp(np.expand_dims(t, -1) * self._carrier_arg + self._phase_arg)
        return np.sum(self.envelope(t) * exp_phases, axis=-1)

    def conjugate(self):
        return SignalSum(*[s.conjugate() for s in self._components])

    def flatten(self) -> Signal:
        """Merge into one ``Signal`` whose carrier is the average of the terms' carriers."""
        if len(self) == 0:
            return Signal(0.0)
        if len(self) == 1:
            return self._components[0]
        ave = np.sum(self.carrier_freq) / len(self)
        shifted = self._carrier_arg - (1j * 2 * np.pi * ave)

        def merged(t):
            return np.sum(self.envelope(t) * np.exp(np.expand_dims(t, -1) * shifted + self._phase_arg), axis=-1)

        return Signal(envelope=merged, carrier_freq=ave, name=str(self))


class DiscreteSignalSum(DiscreteSignal, SignalSum):
    """Sum of piecewise-constant signals that share dt, number of samples and start time
    (signals/signals.py:612-777): ``samples`` is 2-D, axis 0 = time, axis 1 = term."""

    def __init__(self, dt: float, samples, start_time: float = 0.0, carrier_freq=None, phase=None,
                 name: Optional[str] = None):
        samples = np.asarray(samples)
        if samples.ndim != 2:
            raise DynamicsError("DiscreteSignalSum samples must be a 2d array (time, term).")
        if carrier_freq is None:
            carrier_freq = np.zeros(samples.shape[-1], dtype=float)
        if phase is None:
            phase = np.zeros(samples.shape[-1], dtype=float)
        DiscreteSignal.__init__(self, dt=dt, samples=samples, start_time=start_time, carrier_freq=carrier_freq,
                                phase=phase, name=name)
        self._components = [
            DiscreteSignal(dt=self.dt, samples=row, start_time=self.start_time, carrier_freq=freq, phase=phi)
            for row, freq, phi in zip(self.samples.transpose(), np.atleast_1d(carrier_freq), np.atleast_1d(phase))]

    @classmethod
    def from_SignalSum(cls, signal_sum: SignalSum, dt: float, n_samples: int, start_time: float = 0.0,
                       sample_carrier: bool = False):
        """Sample every term of ``signal_sum`` at the bin mid-points (with ``sample_carrier`` the carriers
        are folded into the samples)."""
        times = start_time + (np.arange(n_samples) + 0.5) * dt
        freq = signal_sum.carrier_freq
        if sample_carrier:
            freq = 0.0 * freq
            samples = signal_sum.envelope(times) * np.exp(np.expand_dims(times, -1) * signal_sum._carrier_arg)
        else:
            samples = signal_sum.envelope(times)
        return cls(dt, samples, start_time=start_time, carrier_freq=freq, phase=signal_sum.phase,
                   name=signal_sum.name)

    def complex_value(self, t):
        return SignalSum.complex_value(self, t)

    def conjugate(self):
        return DiscreteSignalSum(self._dt, np.conjugate(self.samples), start_time=self._start_time,
                                 carrier_freq=-self.carrier_freq, phase=-self.phase)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            if idx >= len(self):
                raise IndexError("index out of range for DiscreteSignalSum of length " + str(len(self)))
            return self._components[idx]
        samples = self.samples[:, idx]
        return DiscreteSignalSum(self.dt, samples, start_time=self.start_time,
                                 carrier_freq=np.asarray(self.carrier_freq)[idx], phase=np.asarray(self.phase)[idx])


def _term_product(a: Signal, b: Signal) -> SignalSum:
    """Product of two elementary signals as a sum of two signals:
    Re[f e^{i x}] Re[g e^{i y}] = Re[(f g / 2) e^{i(x+y)}] + Re[(f conj(g) / 2) e^{i(x-y)}]."""
    if a.is_constant and b.is_constant:
        return SignalSum(Signal(a(0.0) * b(0.0)))
    if a.is_constant or b.is_constant:
        const, other = (a, b) if a.is_constant else (b, a)
        c = const(0.0)
        if type(other) is DiscreteSignal:
            return SignalSum(DiscreteSignal(other.dt, c * other.samples, start_time=other.start_time,
                                            carrier_freq=other.carrier_freq, phase=other.phase))
        return SignalSum(Signal(lambda t: c * other.envelope(t), other.carrier_freq, other.phase))
    plus = Signal(lambda t: 0.5 * a.envelope(t) * b.envelope(t), a.carrier_freq + b.carrier_freq,
                  a.phase + b.phase)
    minus = Signal(lambda t: 0.5 * a.envelope(t) * np.conjugate(b.envelope(t)),
                   a.carrier_freq - b.carrier_freq, a.phase - b.phase)
    return SignalSum(plus, minus)


def signal_multiply(sig1, sig2) -> "SignalSum":
    """Product of two signal-like objects (numbers count as constant signals)."""
    try:
        s1, s2 = to_SignalSum(sig1), to_SignalSum(sig2)
    except DynamicsError as err:
        raise DynamicsError("Only a number or a Signal instance can multiply a Signal.") from err
    terms = []
    for a in s1.components:
        for b in s2.components:
            terms += _term_product(a, b).components
    return SignalSum(*terms)


def to_SignalSum(sig) -> SignalSum:
    """Anything signal-like -> SignalSum (numbers become constant signals)."""
    if isinstance(sig, SignalSum):
        return sig
    if isinstance(sig, Signal):
        return SignalSum(sig)
    if np.asarray(sig).ndim == 0 and np.asarray(sig).dtype.kind in "iufc":
        return SignalSum(Signal(sig))
    raise DynamicsError("Input type incompatible with SignalSum.")


class SignalList:
    """List of signals evaluated together: ``SignalList(sigs)(t)`` has shape ``(*t.shape, k)``."""

    def __init__(self, signal_list):
        self._components = [to_SignalSum(s) for s in signal_list]

    @property
    def components(self):
        return self._components

    def __len__(self):
        return len(self._components)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return SignalList(self._components[idx])
        return self._components[idx]

    def __iter__(self):
        return iter(self._components)

    def complex_value(self, t):
        return np.moveaxis(np.asarray([s.complex_value(t) for s in self._components]), 0, -1)

    def __call__(self, t):
        return np.moveaxis(np.asarray([s(t) for s in self._components]), 0, -1)

    def flatten(self) -> "SignalList":
        """SignalList with every component merged into one Signal (signals/signals.py:805-814)."""
        return SignalList([sig.flatten() for sig in self._components])

    @property
    def drift(self):
        out = []
        for comp in self._components:
            val = 0.0
            for term in comp:
                if term.is_constant:
                    val += term(0.0)
            out.append(val)
        return np.asarray(out)

    def table(self, times) -> np.ndarray:
        """Real coefficient table (R, k) float64 at the 1-D array of times (vectorised over t)."""
        times = np.asarray(times, dtype=float)
        if len(self._components) == 0:
            return np.zeros((times.size, 0))
        return np.ascontiguousarray(self(times), dtype=np.float64)


def discrete_term_arrays(instances):
    """Flatten ``instances`` (B lists of k signal-likes) into the CSR arrays ``midyn_sigtable_create``
    takes: ``(term_ptr[B*k+1], term_params[T][4], sample_ptr[T][2],
    samples)``.  Returns ``None`` when any term cannot be evaluated on the device, i.e. is not a
    plain ``DiscreteSignal`` with 1-D samples or a constant ``Signal`` with scalar carrier/phase (a
    Python-callable envelope stays on the host, row a8)."""
    term_ptr = [0]
    params: List[tuple] = []
    ranges: List[tuple] = []
    pool: List[np.ndarray] = []
    seen = {}
    n_pool = 0
    for sigs in instances:
        for sig in sigs:
            # a sweep hands over B * k plain signals: no SignalSum is built around each of them (32768 of them for
            # the 4096-instance sweep of BASELINE cfg 3 cost 0.16 s of a 2.9 s solve)
            if type(sig) is DiscreteSignal or type(sig) is Signal:
                terms = (sig,)
            else:
                terms = to_SignalSum(sig).components
            for term in terms:
                freq, phase = term._carrier_freq, term._phase
                if freq.ndim != 0 or phase.ndim != 0:
                    return None
                if type(term) is DiscreteSignal:
                    smp = term._padded_samples
                    if smp.ndim != 1 or term._dt == 0:
                        return None
                    key = id(smp)
                    if key not in seen:
                        seen[key] = (n_pool, smp.shape[0] - 1, smp)  # keep smp alive: ids stay unique
                        pool.append(smp[:-1])
                        n_pool += smp.shape[0] - 1
                    ranges.append(seen[key][:2])
                    params.append((float(term._dt), float(term._start_time), float(freq), float(phase)))
                elif type(term) is Signal and term.is_constant:
                    val = np.asarray(term.envelope(0.0))
                    if val.ndim != 0:
                        return None
                    ranges.append((n_pool, 1))
                    pool.append(np.asarray([val]))
                    n_pool += 1
                    params.append((0.0, 0.0, float(term.carrier_freq), float(term.phase)))
                else:
                    return None
            term_ptr.append(len(params))
    samples = np.concatenate(pool).astype(np.complex128) if pool else np.zeros(1, dtype=np.complex128)
    if samples.shape[0] == 0:
        samples = np.zeros(1, dtype=np.complex128)
    return (np.asarray(term_ptr, dtype=np.int64), np.asarray(params, dtype=np.float64).reshape(-1, 4),
            np.asarray(ranges, dtype=np.int64).reshape(-1, 2), samples)
