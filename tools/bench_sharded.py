#!/usr/bin/env python3
"""Sharded (multi-device) contexts behind the C ABI, measured: the same SP1 batch through a single-device context and through
`shard([...])` over the given devices, host buffers (one host thread and PCIe link per shard) and HBM-resident rows (peer copies in
two pieces).  On a one-GPU box `--devices 0,0` maps two logical shards to device 0 (with --force-staging the rows take the staging
path a second GPU would take): that measures the overhead of the machinery, not a scaling figure.

    python tools/bench_sharded.py --devices 0,1,2,3,4,5,6,7 --proofs 1048576"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = bytes.fromhex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--devices', default='0,0')
    ap.add_argument('--proofs', type=int, default=1 << 18)
    ap.add_argument('--force-staging', action='store_true')
    ap.add_argument('--repeats', type=int, default=3)
    a = ap.parse_args()
    if a.force_staging:
        os.environ['ZKV_SHARD_FORCE_STAGING'] = '1'
    devs = [int(x) for x in a.devices.split(',')]
    import bench
    g = bench.golden()
    host = bench.synthesize('sp1', a.proofs, 0x5A4B5607, g, 64)          # forks its workers before the GPU is touched
    import torch
    import stylus_zkvm_verifiers_amd as z
    from stylus_zkvm_verifiers_amd import _lib
    L = _lib.lib()
    n = a.proofs
    seals, vk, pv, mut = host['seals'], host['a'], host['b'], host['mutated']
    off = np.arange(n + 1, dtype=np.uint64) * 260
    pvoff = np.arange(n + 1, dtype=np.uint64) * pv.shape[1]
    out = {'devices': devs, 'proofs': n, 'forced_staging': bool(a.force_staging)}
    for name, ver in (('single', z.Sp1Verifier(devs[0])), ('sharded', z.shard([z.Sp1Verifier(d) for d in devs]))):
        ver.reserve(n); ver.synchronize()
        st = np.zeros(n, dtype=np.uint8)
        best = None
        for _ in range(a.repeats):
            t0 = time.perf_counter()
            _lib.check(L.zkv_sp1_verify_batch(ver._h, n, vk.ctypes.data, pv.ctypes.data, pvoff.ctypes.data, seals.ctypes.data, off.ctypes.data, st.ctypes.data, None), 'host batch')
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        ok = bool(((st == 0) == ~mut).all())
        dev = torch.device('cuda', devs[0])
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (vk, pv, seals)]
        d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        bestd = None
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            for _ in range(a.repeats):
                d_st.fill_(255); torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                ver.verify_batch_dev(n, d[0].data_ptr(), d[1].data_ptr(), pv.shape[1], d[2].data_ptr(), d_st.data_ptr(), 0, stream)
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - t0
                bestd = dt if bestd is None else min(bestd, dt)
        okd = bool(((d_st.cpu().numpy() == 0) == ~mut).all())
        out[name] = {'host_buffers_proofs_per_s': n / best, 'host_buffers_ms': best * 1e3, 'hbm_resident_proofs_per_s': n / bestd, 'hbm_resident_ms': bestd * 1e3,
                     'statuses_match_construction': ok and okd}
        ver.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
