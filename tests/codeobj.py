"""Kernel metadata of the gfx950 code object inside libmidyn.so (test infrastructure, no GPU needed).

hipcc embeds the device code as a clang offload bundle ("__CLANG_OFFLOAD_BUNDLE__" + entry table); the gfx950 entry is an
ELF whose NT_AMDGPU_METADATA note (name "AMDGPU", type 32) is a msgpack map with one record per kernel: `.name`,
`.vgpr_count`, `.vgpr_spill_count`, `.sgpr_spill_count`, `.private_segment_fixed_size` (scratch bytes per lane), ...
"""
import struct

import msgpack

BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def extract_code_objects(path, arch="gfx950"):
    """The gfx950 code object of every offload bundle in the file: libmidyn.so is linked from several translation units
    (midyn.hip + one per kernel family) and carries one bundle per unit."""
    data = open(path, "rb").read()
    found, at = [], data.find(BUNDLE_MAGIC)
    while at >= 0:
        p = at + len(BUNDLE_MAGIC)
        (count,) = struct.unpack_from("<Q", data, p)
        p += 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tlen].decode()
            p += tlen
            if arch in triple and size:
                found.append(data[at + off:at + off + size])
        at = data.find(BUNDLE_MAGIC, at + 1)
    if not found:
        raise ValueError(f"{path}: no {arch} entry in any clang offload bundle")
    return found


def extract_code_object(path, arch="gfx950"):
    """The first gfx950 code object (the unit linked first: midyn.hip)."""
    return extract_code_objects(path, arch)[0]


def disassembly(path, tmp_path, objdump="/opt/rocm/lib/llvm/bin/llvm-objdump"):
    """llvm-objdump -d of every gfx950 code object in the library, concatenated."""
    import subprocess

    text = []
    for i, elf in enumerate(extract_code_objects(path)):
        co = tmp_path / f"midyn_{i}.co"
        co.write_bytes(elf)
        text.append(subprocess.run([objdump, "-d", "--mcpu=gfx950", str(co)], capture_output=True, text=True, check=True).stdout)
    return "\n".join(text)


def kernel_metadata(elf):
    """{demangled-ish kernel symbol: metadata record} from the NT_AMDGPU_METADATA note of an AMDGPU ELF64 image."""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        raise ValueError("not an ELF64 image")
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        base = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, base + 4)
        sh_offset, sh_size = struct.unpack_from("<QQ", elf, base + 0x18)
        if sh_type != 7:          # SHT_NOTE
            continue
        p, end = sh_offset, sh_offset + sh_size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if name == b"AMDGPU" and ntype == 32:
                meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                return {k[".name"]: k for k in meta["amdhsa.kernels"]}
    raise ValueError("no NT_AMDGPU_METADATA note")


def library_kernels(path):
    kernels = {}
    for elf in extract_code_objects(path):
        unit = kernel_metadata(elf)
        twice = set(unit) & set(kernels)
        if twice:
            raise ValueError(f"kernels instantiated in two translation units: {sorted(twice)[:3]}")
        kernels.update(unit)
    return kernels
