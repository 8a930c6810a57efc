#!/usr/bin/env python3
"""Time per batch against batch size, in ONE process: n = 1024 ... 65536 RISC Zero proofs resident in HBM, through the automatic mapping
(zkv_ctx_set_lanes_per_proof 0) and through each forced mapping (2 = lane pairs, 16, 64 lanes per proof).  Median of --steps timed steps
(HIP events around the call on the caller's stream).  Prints one line per n and, for the automatic mapping, whether proofs/s is monotone.
    python tools/batch_sweep.py [--max 65536] [--step 1024] [--steps 7] [--lanes 0,2,16,64]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--max', type=int, default=65536)
    ap.add_argument('--step', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=7)
    ap.add_argument('--lanes', default='0,2,16,64')
    ap.add_argument('--sizes', default='')
    args = ap.parse_args()
    import bench
    g = bench.golden()
    sizes = [int(x) for x in args.sizes.split(',')] if args.sizes else [1, 256, 512, 768] + list(range(args.step, args.max + 1, args.step))
    host = bench.synthesize('risc0', max(sizes), 0x5A4B5601, g, 64)          # the device rows cover the largest size: never read past them
    import torch
    dev = torch.device('cuda', 0)
    sh = bench.Shard(host, dev, g)
    ts_stream = torch.cuda.Stream()            # a stream of its own: handle 0 (the default stream) would mean "the context's stream" to the library
    torch.cuda.set_stream(ts_stream)
    stream = ts_stream.cuda_stream
    lanes = [int(x) for x in args.lanes.split(',')]
    assert max(sizes) <= sh.n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rows = []
    for n in sizes:
        rec = {'n': n}
        for l in lanes:
            if l == 64 and n > 8192 or l == 16 and n > 32768:
                continue
            sh.ctx.set_lanes_per_proof(l)
            ts = []
            for k in range(args.steps + 2):
                sh.d_status[:n].fill_(255)
                e0.record()
                sh.ctx.verify_batch_dev(n, sh.d_seals.data_ptr(), sh.d_a.data_ptr(), sh.d_b.data_ptr(), sh.d_status.data_ptr(), 0, stream)
                e1.record()
                torch.cuda.synchronize()
                if k >= 2:
                    ts.append(e0.elapsed_time(e1))
            ok = bool(((sh.d_status[:n].cpu().numpy() == 0) == ~host['mutated'][:n]).all())
            rec['ms_%d' % l] = round(float(np.median(ts)), 3)
            rec['ok_%d' % l] = ok
        rows.append(rec)
        print(json.dumps(rec), flush=True)
    sh.ctx.set_lanes_per_proof(0)
    if 0 in lanes:
        worst = None
        for a, b in zip(rows, rows[1:]):
            ra, rb = a['n'] / a['ms_0'], b['n'] / b['ms_0']
            if rb < ra and (worst is None or rb / ra < worst[2]):
                worst = (a['n'], b['n'], rb / ra)
        print('automatic mapping: proofs/s %s' % ('is monotone over these sizes' if worst is None else
              'drops most from n=%d to n=%d (x%.3f)' % worst))
    assert all(v for r in rows for k, v in r.items() if k.startswith('ok_'))


if __name__ == '__main__':
    main()
