#!/usr/bin/env python3
"""Instruction-class histogram of the marked regions of a gfx950 assembly listing (ZKV_MARK in csrc/zkv_field.h).

    cd stylus_zkvm_verifiers_amd/csrc
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DZKV_FP_MUL_NOINLINE -DZKV_ASM_MARKS --cuda-device-only -S -o /tmp/k_pair.s k_pair.hip
    python tools/asm_hist.py /tmp/k_pair.s k_finalexp2 [region=trips ...]

A region is the straight-line code between '; ZKVMARK begin X' and '; ZKVMARK end X' inside the named kernel (the hot Fp12 bodies are
straight-line code plus calls of the leaf multipliers); a call adds the callee's histogram.  With region=trips arguments the tool also
prints the trip-weighted totals per wavefront, which is what SQ_INSTS_VALU counts."""
import collections
import re
import sys

CLASSES = ['mad64', 'vop3', 'vop2', 'carry', 'dpp', 'lds', 'vmem', 'salu', 'wait', 'other']
CARRY = ('v_addc_co', 'v_subb_co', 'v_add_co', 'v_sub_co', 'v_subbrev_co', 'v_subrev_co')


def classify(mn):
    if mn.startswith(('v_mad_u64_u32', 'v_mad_i64_i32')):
        return 'mad64'
    if mn.endswith('_dpp'):
        return 'dpp'
    if mn.startswith(CARRY):
        return 'carry'
    if mn.startswith('ds_'):
        return 'lds'
    if mn.startswith(('global_', 'flat_', 'buffer_', 'scratch_')):
        return 'vmem'
    if mn.startswith(('s_waitcnt', 's_nop', 's_sleep')):
        return 'wait'
    if mn.startswith('s_'):
        return 'salu'
    if mn.startswith('v_'):
        return 'vop2' if mn.endswith('_e32') else 'vop3'
    return 'other'


def functions(lines):
    out, cur, start = {}, None, 0
    for i, l in enumerate(lines):
        m = re.match(r'\s*\.type\s+(\S+),@function', l)
        if m:
            cur, start = m.group(1), i
        if cur and l.strip().startswith('.size') and cur in l:
            out[cur] = (start, i); cur = None
    return out


def instrs(lines, a, b):
    for l in lines[a:b]:
        if l.startswith('\t') and not l.strip().startswith(('.', ';')):
            yield l.strip()


def walk(lines, start, labels, funcs, memo, stop_mark=None, end=None):
    """Instruction stream from line `start` along the executed path: unconditional branches are followed, conditional ones fall
    through (the marked bodies are straight-line code; their only conditional branches skip lane-masked sections; s_cbranch_execnz is
    followed), calls add the
    callee.  Stops at `stop_mark` (a '; ZKVMARK end X' comment) or at line `end`."""
    h, mn_h = collections.Counter(), collections.Counter()
    pending, i, steps = None, start, 0
    while True:
        if end is not None and i >= end:
            break
        l = lines[i]
        if stop_mark and stop_mark in l:
            break
        i += 1; steps += 1
        if steps > 400000:
            raise RuntimeError('no end mark on the path')
        if not l.startswith('\t') or l.strip().startswith(('.', ';')):
            continue
        ins = l.strip()
        mn = ins.split()[0]
        h[classify(mn)] += 1; mn_h[mn] += 1
        m = re.search(r'(_ZN\w+)@rel32@lo', ins)
        if m:
            pending = m.group(1)
        if mn == 's_swappc_b64' and pending:
            if pending not in memo:
                fa, fb = funcs[pending]
                memo[pending] = walk(lines, fa, labels, funcs, memo, end=fb)
            ch, cm = memo[pending]
            h.update(ch); mn_h.update(cm)
        if mn in ('s_branch', 's_cbranch_execnz'):      # execnz: taken whenever a lane is active (lane-masked sections placed out of line)
            i = labels[ins.split()[1]]
    return h, mn_h


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    trips = dict((a.split('=')[0], int(a.split('=')[1])) for a in sys.argv[3:])
    lines = open(path).read().split('\n')
    funcs = functions(lines)
    kname = [f for f in funcs if kernel in f][0]
    ka, kb = funcs[kname]
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'(\.LBB\w+):', l)
        if m:
            labels[m.group(1)] = i
    memo, regions = {}, {}
    for i in range(ka, kb):
        m = re.search(r'; ZKVMARK begin (\S+)', lines[i])
        if m:
            regions.setdefault(m.group(1), []).append(i)
    print('%-14s %7s ' % ('region', 'instr') + ' '.join('%6s' % c for c in CLASSES))
    total = collections.Counter()
    for name, spans in regions.items():
        for n, a in enumerate(spans):
            try:
                h, mn_h = walk(lines, a + 1, labels, funcs, memo, stop_mark='; ZKVMARK end ' + name)
            except (RuntimeError, KeyError):
                print('%-14s (region holds a loop or a two-way branch: not a single path)' % name)
                continue
            tot = sum(h.values())
            print('%-14s %7d ' % (name if len(spans) == 1 else '%s#%d' % (name, n), tot) + ' '.join('%6d' % h[c] for c in CLASSES))
            top = ', '.join('%s %d' % kv for kv in mn_h.most_common(14))
            print('    ' + top)
            if name in trips and n == 0:
                for c in CLASSES:
                    total[c] += h[c] * trips[name]
    if trips:
        tot = sum(total.values())
        print('%-14s %7d ' % ('weighted', tot) + ' '.join('%6d' % total[c] for c in CLASSES))
        valu = sum(total[c] for c in ('mad64', 'vop3', 'vop2', 'carry', 'dpp'))
        print('VALU instructions in the weighted regions: %d (mad64 %.1f %%; others per mad64 %.2f)' % (valu, 100.0 * total['mad64'] / valu, (valu - total['mad64']) / total['mad64']))


if __name__ == '__main__':
    main()
