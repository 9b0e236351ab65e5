import numpy as np
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import _lib
from qiskit_dynamics_amd.signals import discrete_term_arrays
rng = np.random.default_rng(11)
times = np.sort(rng.uniform(-1.0, 12.0, 211))
worst = 0
for trial in range(200):
    ns = int(rng.integers(1, 40))
    d = qd.DiscreteSignal(dt=rng.uniform(0.05, 0.7), samples=rng.normal(size=ns) + 1j * rng.normal(size=ns),
                          start_time=rng.uniform(-0.5, 2.0), carrier_freq=rng.uniform(-6, 6), phase=rng.uniform(-3, 3))
    d = qd.DiscreteSignal(dt=100.0, samples=np.array([1.0 + 0j]), start_time=-50.0, carrier_freq=d.carrier_freq, phase=d.phase)
    arr = discrete_term_arrays([[d]])
    got = _lib.SignalTable(_lib.default_context(), 1, 1, times, *arr).fetch()[0, :, 0]
    want = qd.SignalList([d]).table(times)[:, 0]
    i = np.argmax(np.abs(got - want))
    err = abs(got[i] - want[i])
    if err > worst:
        worst = err
        t = times[i]
        a = 6.283185307179586 * float(d.carrier_freq)
        arg = t * a + float(d.phase)
        print(trial, err, "t", t, "arg", arg, "ulp", np.spacing(arg), "cos host", np.cos(arg), got[i], want[i])
        carg = np.expand_dims(t, -1) * qd.SignalList([d])[0]._carrier_arg + qd.SignalList([d])[0]._phase_arg
        print("   numpy carg", carg, carg.imag - arg)
