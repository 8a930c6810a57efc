#!/usr/bin/env python3
"""Event timeline of a host-buffer batch call (ZKV_HOST_TRACE=1: run_host_batch prints, per pass, when each segment's copy and kernels were
done) for three first-segment sizes, next to the HBM-resident time of the same batch: what SURVEY 8(d)'s C-ABI figure pays on top of the
resident one.  2^20 SP1 proofs; profiles/round4_host_boundary_timeline.txt.
    python tools/host_boundary_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    g = bench.golden()
    host = bench.synthesize('sp1', 1 << 20, 0x5A4B5602, g, 64)
    import torch
    dev = torch.device('cuda', 0)
    sh = bench.Shard(host, dev, g)
    os.environ['ZKV_HOST_TRACE'] = '1'
    for fs in ('', '32768', '131072'):
        if fs:
            os.environ['ZKV_HOST_FIRST_SEGMENT'] = fs
        print('first segment', fs or 'default (65536)', bench.host_boundary_rate(sh, repeats=2), flush=True)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        t0 = time.perf_counter()
        sh.enqueue(stream)
        torch.cuda.synchronize()
        print('resident ms %.2f' % ((time.perf_counter() - t0) * 1e3), flush=True)


if __name__ == '__main__':
    main()
