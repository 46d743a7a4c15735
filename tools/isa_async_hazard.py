#!/usr/bin/env python3
"""usage: tools/isa_async_hazard.py file.s [kernel-substring]
       tools/isa_async_hazard.py --lib [libofdmtools_hip.so] [kernel-substring]      (the shipped code objects, disassembled)
The one-exchange kernels issue their sample / window loads from inline asm (pinned where they stand, spread over a
step) and wait for them with an explicit s_waitcnt.  The compiler does not know that the destination registers of
such a load are not valid until that wait: if it decides to COPY one of them in between (a register shuffle at a loop
edge, say) the copy reads a register whose load is still in flight - welch8kws_kernel's first build did exactly that
and returned garbage from its second segment on.  This tool walks the control-flow graph of every kernel in a
`hipcc -S` listing and reports any instruction that reads or overwrites a VGPR written by an INLINE-ASM global load
between the load and the s_waitcnt vmcnt that covers it (loads return in order: vmcnt(N) leaves the N youngest
pending; the compiler's own loads count towards N but are its own business) - and the same for inline-asm ds_read
and lgkmcnt.  With --lib the code objects inside the built library are disassembled (llvm-objdump) and EVERY load is
followed - the asm markers are gone there, and a load the compiler issued itself is waited for correctly by
construction, so it cannot produce a finding; this is the form tests/test_abi_cpu.py runs (seconds, and it is the
binary that ships).  Exit status 1 if anything is found."""
import re
import sys

import os
import subprocess
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return frozenset(out)


def kernels(lines):
    name, body = None, []
    for ln in lines:
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        body.append(ln.rstrip('\n'))
        if ln.strip().startswith('s_endpgm') and False:
            pass
        if re.match(r'^\t\.end_amdhsa_kernel|^\.Lfunc_end', ln):
            yield name, body
            name = None


def instrs(body):
    """-> list of (opcode, operands, inside inline asm), labels {name: index of the next instruction}"""
    out, labels, in_asm = [], {}, False
    for ln in body:
        if 'ASMSTART' in ln:
            in_asm = True
            continue
        if 'ASMEND' in ln:
            in_asm = False
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', ln)
        if m:
            labels[m.group(1)] = len(out)
            continue
        m = re.match(r'^\t([a-z_0-9]+)\s*(.*?)(?:\s*;.*)?$', ln)
        if m and not ln.startswith('\t.'):
            out.append((m.group(1), m.group(2), in_asm))
    return out, labels


LOADS = ('global_load', 'buffer_load', 'flat_load', 'scratch_load')
LGKM = ('ds_', 's_load', 's_buffer_load', 's_memtime', 's_memrealtime', 's_sendmsg')


def drop(pending, keep):
    return pending[len(pending) - keep:] if keep else ()


def walk(ins, labels, kname, found):
    """State per path: (vm, lg) - the destinations of the loads in flight per counter, oldest first; a load the compiler
    issued itself (it places its own waits) counts with an empty set.  LDS operations return in order; scalar loads
    share lgkmcnt and may not - the compiler waits lgkmcnt(0) for those, which clears everything here as well."""
    stack, seen = [(0, (), ())], set()
    while stack:
        i, vm, lg = stack.pop()
        while i < len(ins):
            key = (i, vm, lg)
            if key in seen:
                break
            seen.add(key)
            op, args, in_asm = ins[i]
            i += 1
            if op == 's_endpgm':
                break
            if op.startswith('s_waitcnt'):
                m = re.search(r'vmcnt\((\d+)\)', args)
                if m:
                    vm = drop(vm, int(m.group(1)))
                m = re.search(r'lgkmcnt\((\d+)\)', args)
                if m:
                    lg = drop(lg, int(m.group(1)))
                continue
            live = frozenset().union(*(vm + lg)) if (vm or lg) else frozenset()
            is_vm = op.startswith(LOADS) or (op.startswith(('global_atomic', 'buffer_atomic', 'flat_atomic')) and 'sc0' in args)
            is_lg = op.startswith(LGKM)
            if is_vm or is_lg:
                parts = args.split(',')
                reads = regs(','.join(parts[1:])) if (is_vm or op.startswith(('ds_read', 'ds_bpermute', 'ds_permute', 's_'))) else regs(args)
                hit = reads & live
                if hit:
                    found.append((kname, op + ' ' + args, sorted(hit)))
                writes = regs(parts[0]) if (in_asm and (is_vm or op.startswith('ds_read'))) else frozenset()
                if is_vm:
                    vm = (vm + (writes,))[-64:]
                else:
                    lg = (lg + (writes,))[-64:]
                continue
            is_store = op.startswith(('global_store', 'buffer_store', 'scratch_store', 'flat_store', 'global_atomic', 'buffer_atomic',
                                      'flat_atomic'))
            if live and (op.startswith('v_') or is_store):
                hit = regs(args) & live
                if hit:
                    found.append((kname, op + ' ' + args, sorted(hit)))
            if is_store:      # stores (and atomics without a return) count in vmcnt on gfx9 as well
                vm = (vm + (frozenset(),))[-64:]
            if op.startswith('s_cbranch'):
                tgt = args.strip()
                if tgt in labels:
                    stack.append((labels[tgt], vm, lg))
            elif op == 's_branch':
                tgt = args.strip()
                if tgt not in labels:
                    break
                i = labels[tgt]


def objdump_kernels(lib, skip_objects_with=()):
    """-> (name, instruction list, labels) per kernel of every code object in the library; a code object whose symbol
    names contain one of `skip_objects_with` (bytes) is left out whole (fft_any.hip's: twelve 20 000-instruction builds of
    one compiler-scheduled kernel without inline asm - llvm-objdump --symbolize-operands alone takes minutes on it)"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import kernel_resources
    for elf in kernel_resources.code_objects(lib):
        if any(tag in elf for tag in skip_objects_with):
            continue
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--symbolize-operands', f.name],
                                 stdout=subprocess.PIPE, check=True).stdout.decode()
        name, ins, labels = None, [], {}
        for ln in txt.splitlines():
            m = re.match(r'^[0-9a-f]+ <(\w+)>:', ln)
            if m:
                if re.fullmatch(r'L\d+', m.group(1)):
                    labels[m.group(1)] = len(ins)
                else:
                    if name and ins:
                        yield name, ins, labels
                    name, ins, labels = m.group(1), [], {}
                continue
            m = re.match(r'^\t([a-z_0-9]+)\s*(.*?)\s*//', ln)
            if m and name:
                ins.append((m.group(1), m.group(2), True))      # every load counts as "pinned" here
        if name and ins:
            yield name, ins, labels


def main():
    args = sys.argv[1:]
    bad, n = [], 0
    if args and args[0] == '--lib':
        lib = args[1] if len(args) > 1 and args[1].endswith('.so') else os.path.join(
            os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gr-ofdm_tools_amd', 'lib', 'libofdmtools_hip.so')
        want = args[-1] if len(args) > 1 and not args[-1].endswith('.so') else ''
        source = objdump_kernels(lib)
    else:
        want = args[1] if len(args) > 1 else ''
        source = ((k, *instrs(b)) for k, b in kernels(open(args[0]).readlines()))
    for kname, ins, labels in source:
        if want not in kname or not kname.startswith('_Z'):
            continue
        n += 1
        walk(ins, labels, kname, bad)
    seen = set()
    for kname, text, hit in bad:
        if (kname, text) in seen:
            continue
        seen.add((kname, text))
        print('%s: `%s` touches v%s while its load is in flight' % (kname[-60:], text[:70], hit))
    print('%d kernels scanned, %d findings' % (n, len(seen)))
    return n, len(seen)


if __name__ == '__main__':
    sys.exit(1 if main()[1] else 0)
