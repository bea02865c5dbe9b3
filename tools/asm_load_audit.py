#!/usr/bin/env python3
"""Audit of hand-issued vector-memory loads in the compiled kernels (gfx950 assembly).

csrc/conv3halo.hip issues its fp32 halo loads as inline asm (`buffer_load_dwordx4` into "=&v" outputs) and counts their
completion by hand (`s_waitcnt vmcnt(N)`, in-order retirement) because the compiler would drain the whole LDS-DMA ring in front
of a load it can see.  The compiler therefore believes the destination registers are valid right after the asm statement: any
instruction it places between the load and the counted wait that READS or WRITES one of those registers (a copy, a spill, an
early use) is a silent bug.  This script replays every kernel of an assembly listing in program order:

  * every VMEM instruction (buffer_/global_/scratch_ load, store, LDS-DMA) enters an in-order queue, loads with their
    destination VGPRs;
  * `s_waitcnt vmcnt(N)` retires all but the newest N entries;
  * any other instruction touching a VGPR that is still the destination of a queued load is reported.

Control flow is not modelled (a linear pass over the text: the unrolled nine-tap bodies are straight-line code, which is what
matters here).

    hipcc -S --cuda-device-only --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Istable-diffusion_amd/csrc \
          stable-diffusion_amd/csrc/conv3halo.hip -o /tmp/conv3halo.s
    python tools/asm_load_audit.py /tmp/conv3halo.s [--kernel gn]
"""
import argparse
import re
import sys

VREG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
ACC = re.compile(r'\ba(\d+)\b|\ba\[(\d+):(\d+)\]')


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def is_vmem(op):
    return op.startswith(('buffer_', 'global_', 'scratch_', 'flat_', 'tbuffer_'))


def audit(lines, name):
    queue = []          # (lineno, dst set or None)
    findings = []
    max_q = 0
    n_hand = 0
    for no, raw in lines:
        line = raw.split(';')[0].strip()
        if not line or line.startswith('.') or line.endswith(':'):
            continue
        parts = line.split(None, 1)
        op = parts[0]
        args = parts[1] if len(parts) > 1 else ''
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', args)
            if m:
                n = int(m.group(1))
                if len(queue) > n:
                    queue = queue[len(queue) - n:]
            continue
        touched = vregs(args)
        pending = set().union(*[d for _, d in queue if d]) if queue else set()
        if is_vmem(op):
            # address / data operands of a VMEM instruction are reads as well
            dst = None
            # (only the hand-issued kind carries destinations: hipcc's own loads are global_/scratch_ and it waits for them itself;
            # tracking them would only add if/else artefacts of the linear pass)
            if op == 'buffer_load_dwordx4' and ' lds' not in (' ' + args):
                first = args.split(',')[0]
                dst = vregs(first)
                reads = vregs(','.join(args.split(',')[1:]))
            else:
                reads = touched
            if dst is None and 'load' in op and ' lds' not in (' ' + args):
                reads = vregs(','.join(args.split(',')[1:])) | vregs(args.split(',')[0])      # a compiler load overwriting a pending register
            bad = (reads | (dst or set())) & pending
            if bad:
                findings.append((no, raw.strip(), sorted(bad)))
            queue.append((no, dst))
            max_q = max(max_q, len(queue))
            if dst and op == 'buffer_load_dwordx4':
                n_hand += 1
            continue
        bad = touched & pending
        if bad:
            findings.append((no, raw.strip(), sorted(bad)))
    print(f'{name}: {n_hand} buffer_load_dwordx4 to registers, deepest queue {max_q}, {len(findings)} unwaited touches')
    for no, text, regs in findings[:20]:
        print(f'   line {no}: {text}    <- pending v{regs[0]}..v{regs[-1]}')
    return len(findings)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('asm')
    ap.add_argument('--kernel', default='gn', help='substring of the (mangled) kernel names to audit')
    a = ap.parse_args()
    cur, body, total = None, [], 0
    with open(a.asm) as f:
        for no, raw in enumerate(f, 1):
            m = re.match(r'^(_Z\w+):', raw)
            if m:
                cur, body = m.group(1), []
                continue
            if cur is not None:
                if raw.startswith('.Lfunc_end'):
                    if a.kernel in cur:
                        total += audit(body, cur[:110])
                    cur = None
                else:
                    body.append((no, raw))
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main())
