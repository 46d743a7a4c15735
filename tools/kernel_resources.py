#!/usr/bin/env python3
"""Register / scratch / LDS use of every gfx950 kernel in the SHIPPED library, read from the code objects inside
lib/libofdmtools_hip.so (the .hip_fatbin section holds one clang offload bundle per translation unit; each bundle's
gfx950 entry is an ELF whose NT_AMDGPU_METADATA note lists the kernels).  No GPU needed.

  python tools/kernel_resources.py [path/to/lib.so] [--spills]      table (only kernels with scratch with --spills)

tests/test_abi_cpu.py::test_hot_kernels_have_no_scratch imports kernels() and asserts that the default builds of the
tuned kernels carry no scratch memory (a spilled register comes back at memory latency in every step: DESIGN 4.1c).
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, 'gr-ofdm_tools_amd', 'lib', 'libofdmtools_hip.so')
LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def _section(path, name):
    """Raw bytes of ELF section `name` (64-bit little-endian ELF, no external tool)."""
    data = open(path, 'rb').read()
    shoff, = struct.unpack_from('<Q', data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from('<HHH', data, 0x3A)
    secs = [struct.unpack_from('<IIQQQQIIQQ', data, shoff + i * shentsize) for i in range(shnum)]
    stroff = secs[shstrndx][4]
    for s in secs:
        end = data.index(b'\0', stroff + s[0])
        if data[stroff + s[0]:end].decode() == name:
            return data[s[4]:s[4] + s[5]]
    raise KeyError(name)


def code_objects(path=DEFAULT_LIB, arch='gfx950'):
    """-> list of ELF byte strings, one per bundle that has an entry for `arch`."""
    fat = _section(path, '.hip_fatbin')
    out, pos = [], 0
    while True:
        pos = fat.find(MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from('<Q', fat, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from('<QQQ', fat, q)
            triple = fat[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if arch in triple and size:
                out.append(fat[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def kernels(path=DEFAULT_LIB):
    """-> {demangled kernel name: dict(vgpr, agpr, sgpr, spill_vgpr, spill_sgpr, scratch, lds)} over all code objects."""
    res = {}
    for elf in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', f.name], stdout=subprocess.PIPE,
                                 check=True).stdout.decode()
        for blk in re.split(r'\n\s*- \.agpr_count:', '\n' + txt)[1:]:
            blk = '.agpr_count:' + blk
            get = lambda k, b=blk: re.search(r'\.%s:\s*(\S+)' % re.escape(k), b)      # noqa: E731
            name = get('name')
            if not name:
                continue
            num = lambda k: int(get(k).group(1)) if get(k) else 0      # noqa: E731
            one = dict(vgpr=num('vgpr_count'), agpr=num('agpr_count'), sgpr=num('sgpr_count'),
                       spill_vgpr=num('vgpr_spill_count'), spill_sgpr=num('sgpr_spill_count'),
                       scratch=num('private_segment_fixed_size'), lds=num('group_segment_fixed_size'))
            # two translation units may hold a kernel of the same name (welch4096.hip is built twice, `dpp` and `pipe`,
            # in an anonymous namespace): report the larger figure of each
            old = res.get(name.group(1))
            res[name.group(1)] = one if old is None else {k: max(v, old[k]) for k, v in one.items()}
    if res:
        names = list(res)
        dem = subprocess.run(['c++filt'] + names, stdout=subprocess.PIPE,
                             check=True).stdout.decode().splitlines()
        res = {d.replace('oth::(anonymous namespace)::', '').replace('void ', '', 1): res[m] for d, m in zip(dem, names)}
    return res


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    ks = kernels(args[0] if args else DEFAULT_LIB)
    only = '--spills' in sys.argv
    print('%-92s %5s %5s %6s %8s %7s' % ('kernel', 'VGPR', 'SGPR', 'spills', 'scratch B', 'LDS B'))
    for name in sorted(ks):
        k = ks[name]
        if only and not k['scratch']:
            continue
        short = re.sub(r'\((oth::)?\w+Args\)$', '', name)
        print('%-92s %5d %5d %6d %8d %7d' % (short[:92], k['vgpr'] + k['agpr'], k['sgpr'], k['spill_vgpr'], k['scratch'], k['lds']))
