"""Probe: would running the two halves of a 2^20-proof aggregate-mode chunk on two streams hide the latency-bound kernels (reduction,
pseudo-proofs, second pass) of one half behind the other half's full-size kernels?  Two contexts, two streams, half the batch each."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def main():
    g = bench.golden()
    out = []
    hosts = {mu: bench.synthesize('sp1', 1 << 20, 0x5A4B5602, g, mu) for mu in (64, 0)}
    import torch
    dev = torch.device('cuda', 0)
    for mu, h in hosts.items():
        n = h['n']
        halves = []
        for k in range(2):
            sl = slice(k * n // 2, (k + 1) * n // 2)
            hh = dict(h, n=n // 2, seals=h['seals'][sl], a=h['a'][sl], b=h['b'][sl], mutated=h['mutated'][sl])
            halves.append(bench.Shard(hh, dev, g))
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for sub in (64, 16):
            for s in halves:
                s.ctx.set_aggregate_check(True, seed=bytes(range(32)), sub_batch=sub); s.ctx.reserve(s.n); s.ctx.synchronize()
            for mode in ('one stream', 'two streams'):
                def step():
                    for k, s in enumerate(halves):
                        st = streams[k if mode == 'two streams' else 0]
                        with torch.cuda.stream(st):
                            s.enqueue(st.cuda_stream)
                step(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3): step()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3 / 3
                ok = all(bool(((s.d_status.cpu().numpy() == 0) == ~s.mutated).all()) for s in halves)
                print(json.dumps({'mutate_every': mu, 'sub': sub, 'mode': mode, 'ms': round(ms, 3), 'proofs_per_s': round(n / ms * 1e3), 'parity': ok}), flush=True)
        for s in halves: s.ctx.close()


if __name__ == '__main__':
    main()
