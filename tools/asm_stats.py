#!/usr/bin/env python3
"""Per-function statistics of a gfx950 assembly listing: instruction count, multiplies, scratch / LDS / global accesses, calls, and
the scratch frame the compiler reports.  Used while tuning the pair kernels:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DZKV_FP_MUL_NOINLINE --cuda-device-only -S -o /tmp/k_pair.s k_pair.hip
    python tools/asm_stats.py /tmp/k_pair.s"""
import collections
import re
import subprocess
import sys


def main(path):
    lines = open(path).read().split('\n')
    funcs, cur = [], None
    for i, l in enumerate(lines):
        m = re.match(r'\s*\.type\s+(\S+),@function', l)
        if m:
            cur = m.group(1); funcs.append([cur, i, len(lines), None, None])
        if cur and l.strip().startswith('.size') and cur in l:
            funcs[-1][2] = i
        m = re.match(r'; ScratchSize: (\d+)', l)
        if m and funcs and funcs[-1][3] is None:
            funcs[-1][3] = int(m.group(1))
        m = re.match(r'; NumVgprs: (\d+)', l)
        if m and funcs and funcs[-1][4] is None:
            funcs[-1][4] = int(m.group(1))
    names = subprocess.run(['c++filt'], input='\n'.join(f[0] for f in funcs), capture_output=True, text=True).stdout.split('\n')
    print('%-66s %6s %5s %7s %4s %5s %5s %5s %5s' % ('function', 'instr', 'mad64', 'scratch', 'lds', 'glob', 'calls', 'frame', 'vgpr'))
    for (name, a, b, frame, vg), dn in zip(funcs, names):
        ins = [l.strip().split()[0] for l in lines[a:b] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        c = collections.Counter(ins)
        cnt = lambda p: sum(v for k, v in c.items() if k.startswith(p))
        print('%-66s %6d %5d %7d %4d %5d %5d %5s %5s' % (dn.replace('zkv::', '')[:66], len(ins), c.get('v_mad_u64_u32', 0), cnt('scratch_'), cnt('ds_'),
                                                          cnt('global_') + cnt('flat_'), c.get('s_swappc_b64', 0), frame, vg))


if __name__ == '__main__':
    main(sys.argv[1])
