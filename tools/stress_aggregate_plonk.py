"""Soak of the aggregate check on a PLONK context: random sizes, damage densities, sub-batch sizes and secrets; statuses == per-proof path
== accept <=> undamaged.  python tools/stress_aggregate_plonk.py [--seconds 180]"""
import argparse, json, os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = bytes.fromhex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=180)
    args = ap.parse_args()
    import torch
    import stylus_zkvm_verifiers_amd as zkv
    pool = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'plonk_pool.json')))
    rows = lambda key: np.stack([np.frombuffer(H(p[key]), dtype=np.uint8) for p in pool['proofs']])
    P0, V0, W0 = rows('proof'), rows('vkey'), rows('public_values')
    rng = random.Random(0x91A); nrng = np.random.default_rng(0x91A)
    dev = torch.device('cuda', 0)
    os.environ['ZKV_AGG_MIN'] = '64'
    v = zkv.Sp1PlonkVerifier(H(pool['vk']), H(pool['verifier_hash']))
    stats = {'batches': 0, 'proofs': 0, 'damaged': 0, 'sub_batches_failed': 0, 'mismatches': 0}
    t_end = time.time() + args.seconds
    while time.time() < t_end:
        n = rng.choice((64, 65, 127, 129, 1000, rng.randrange(64, 20000), rng.randrange(64, 2000)))
        density = rng.choice((0.0, 0.002, 0.02, 0.2, 1.0))
        src = nrng.integers(0, len(P0), n)
        P, V, W = P0[src].copy(), V0[src].copy(), W0[src].copy()
        mut = nrng.random(n) < density
        for i in np.flatnonzero(mut):
            kd = rng.randrange(3)
            if kd == 0: P[i, 4 + 32 * rng.randrange(27) + 31] ^= 1
            elif kd == 1: W[i, -1] ^= 1
            else: P[i, 0] ^= 1
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (V, W, P)]
        def run():
            st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
            v.verify_batch_dev(n, d[0].data_ptr(), d[1].data_ptr(), 96, d[2].data_ptr(), st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            return st.cpu().numpy()
        v.set_aggregate_check(False)
        plain = run()
        v.set_aggregate_check(True, seed=rng.randbytes(32) if rng.random() < 0.5 else None, sub_batch=rng.choice((16, 32, 64, 128, 256)))
        c0 = v.aggregate_counters(); agg = run(); c1 = v.aggregate_counters()
        bad = int((agg != plain).sum()) + int(((plain == 0) != ~mut).sum())
        stats['batches'] += 1; stats['proofs'] += n; stats['damaged'] += int(mut.sum()); stats['sub_batches_failed'] += c1[1] - c0[1]; stats['mismatches'] += bad
        if bad: print('MISMATCH', n, density, flush=True)
    print(json.dumps(dict(stats, seconds=args.seconds, verdict='ok' if stats['mismatches'] == 0 else 'MISMATCH')), flush=True)
    sys.exit(0 if stats['mismatches'] == 0 else 1)


if __name__ == '__main__':
    main()
