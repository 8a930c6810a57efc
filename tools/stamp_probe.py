#!/usr/bin/env python3
"""Where does a launch of the two big kernels spend its time?  Runs a RISC Zero batch (default 2^16 proofs = one round of 2,048 wavefronts)
through a -DZKV_STAMPS build of the library (tools/ab_build.py stamps:k_pair:-DZKV_STAMPS) and reads back, per wavefront, the constant
100 MHz clock at its first and last instruction and the place it ran:
    ZKV_LIB_PATH=stylus_zkvm_verifiers_amd/csrc/build/ab/libzkv_stamps.so python tools/stamp_probe.py [--proofs N] [--repeat K]
Prints, per kernel: HIP-event stage time, first start -> last end, the spread of the starts, the distribution of the wavefronts' own
durations, the spread of the ends, and the same per XCD."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--proofs', type=int, default=1 << 16)
    ap.add_argument('--repeat', type=int, default=4)
    ap.add_argument('--vm', default='risc0')
    args = ap.parse_args()
    import bench
    g = bench.golden()
    host = bench.synthesize(args.vm, args.proofs, 0x5A4B5601, g, 64)
    import torch
    from stylus_zkvm_verifiers_amd import _lib
    dev = torch.device('cuda', 0)
    sh = bench.Shard(host, dev, g)
    L = _lib.lib()
    L.zkv_diag_set_stamps.argtypes = [C.c_void_p]
    n_waves = (2 * args.proofs + 63) // 64
    buf = torch.zeros((2, n_waves, 8), dtype=torch.int64, device=dev)
    assert L.zkv_diag_set_stamps(buf.data_ptr()) == 0
    stream = torch.cuda.current_stream().cuda_stream
    out = []
    for rep in range(args.repeat):
        buf.zero_()
        torch.cuda.synchronize()
        sh.enqueue(stream)
        sh.ctx.synchronize()
        torch.cuda.synchronize()
        st = sh.ctx.last_stage_ms()
        b = buf.cpu().numpy().astype(np.uint64)
        rec = {'rep': rep, 'stage_ms': {k: round(float(v), 3) for k, v in zip(bench.STAGES, st)}}
        for k, name in enumerate(('k_miller2', 'k_finalexp2')):
            r = b[k]
            live = r[:, 1] > 0                        # wavefronts that ran to the end (rejected proofs leave early)
            t0, t1 = r[:, 0].astype(np.float64) / 100.0, r[:, 1].astype(np.float64) / 100.0      # microseconds
            base = t0.min()
            dur = (t1 - t0)[live]
            cyc = (r[:, 5].astype(np.float64) - r[:, 4].astype(np.float64))[live]
            xcc = (r[:, 3] & 0xf).astype(int)
            hw = r[:, 2]
            cu = ((hw >> 8) & 0xf).astype(int) + 16 * ((hw >> 12) & 0x1).astype(int) + 32 * ((hw >> 13) & 0x7).astype(int)
            simd = (hw >> 4) & 0x3
            key = (xcc.astype(np.int64) << 20) | (cu.astype(np.int64) << 4) | simd.astype(np.int64)
            wid = (hw & 0xf).astype(int)
            import collections
            groups = collections.defaultdict(list)
            for kk, ww in zip(key.tolist(), wid.tolist()):
                groups[kk].append(ww)
            slot_sets = collections.Counter(tuple(sorted(set(v))) for v in groups.values())
            q = lambda a, p: float(np.percentile(a, p))
            rec[name] = {
                'first_start_to_last_end_us': round(float(t1[live].max() - base), 1),
                'start_spread_us': {'p50': round(q(t0 - base, 50), 1), 'p99': round(q(t0 - base, 99), 1), 'max': round(float((t0 - base).max()), 1)},
                'wave_duration_us': {'min': round(float(dur.min()), 1), 'p1': round(q(dur, 1), 1), 'p50': round(q(dur, 50), 1), 'p99': round(q(dur, 99), 1), 'max': round(float(dur.max()), 1)},
                'end_us': {'p1': round(q(t1[live] - base, 1), 1), 'p50': round(q(t1[live] - base, 50), 1), 'p99': round(q(t1[live] - base, 99), 1), 'max': round(float((t1[live] - base).max()), 1)},
                'shader_clock_ghz_p50': round(q(cyc / dur, 50) / 1000.0, 3),
                'waves_alive_to_the_end': int(live.sum()), 'waves': int(n_waves), 'simds_seen': len(groups),
                'wave_slot_ids_per_simd': {str(k): v for k, v in slot_sets.most_common(6)},
                'per_xcd': {str(x): {'waves': int((xcc == x).sum()), 'dur_p50_us': round(q((t1 - t0)[live & (xcc == x)], 50), 1) if (live & (xcc == x)).any() else None,
                                     'last_end_us': round(float((t1 - base)[live & (xcc == x)].max()), 1) if (live & (xcc == x)).any() else None,
                                     'first_start_us': round(float((t0 - base)[xcc == x].min()), 1) if (xcc == x).any() else None,
                                     'distinct_cu_ids': int(len(set(cu[xcc == x])))} for x in sorted(set(xcc))},
            }
        out.append(rec)
        print(json.dumps(rec))
    return out


if __name__ == '__main__':
    main()
