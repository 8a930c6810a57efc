"""Aggregate check on / off on device-resident batches: ms per batch, stage times, sub-batch counters, parity with the generator's labels.
    python tools/bench_aggregate.py [--vm sp1] [--log2 20] [--mutate 64,0] [--steps 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--vm', default='sp1')
    ap.add_argument('--log2', default='20')
    ap.add_argument('--mutate', default='64,0')
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--sub', default='64')
    ap.add_argument('--no-baseline', action='store_true', help='skip the run with the aggregate check off')
    args = ap.parse_args()
    g = bench.golden()
    hosts = {}
    for lg in [int(x) for x in args.log2.split(',')]:
        for mu in [int(x) for x in args.mutate.split(',')]:
            hosts[(lg, mu)] = (bench.synthesize_plonk(1 << lg, 0x5A4B5605, mu) if args.vm == 'plonk' else
                               bench.synthesize(args.vm, 1 << lg, 0x5A4B5602 if args.vm == 'sp1' else 0x5A4B5601, g, mu))
    import torch
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream().cuda_stream
    for (lg, mu), h in hosts.items():
        s = bench.Shard(h, dev, g)
        for on in ([] if args.no_baseline else [0]) + [int(x) for x in args.sub.split(',')]:
            s.ctx.set_aggregate_check(bool(on), seed=bytes(range(32)) if on else None, sub_batch=on or 64)
            s.ctx.reserve(s.n); s.ctx.synchronize()
            c0 = s.ctx.aggregate_counters()
            s.enqueue(stream); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                s.enqueue(stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / args.steps
            st = s.d_status.cpu().numpy()
            c1 = s.ctx.aggregate_counters()
            print(json.dumps({'vm': args.vm, 'n': s.n, 'mutate_every': mu, 'aggregate': on, 'ms': round(ms, 3), 'proofs_per_s': round(s.n / ms * 1e3),
                              'stage_ms': [round(float(x), 3) for x in s.ctx.last_stage_ms()], 'parity': bool(((st == 0) == ~s.mutated).all()),
                              'sub_batches': [(c1[0] - c0[0]) // (args.steps + 1), (c1[1] - c0[1]) // (args.steps + 1)]}), flush=True)
        s.ctx.close()
        del s


if __name__ == '__main__':
    main()
