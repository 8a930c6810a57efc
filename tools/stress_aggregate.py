"""Soak test of the aggregate check: random batch sizes, reject densities and placements (shuffled, clustered), sub-batch and group sizes and
secrets, RISC Zero and SP1; every status must equal the per-proof path's.  python tools/stress_aggregate.py [--seconds 300]"""
import argparse, json, os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = bytes.fromhex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=300)
    ap.add_argument('--seed', type=int, default=0xA66)
    args = ap.parse_args()
    import torch
    import stylus_zkvm_verifiers_amd as zkv
    from stylus_zkvm_verifiers_amd import synth
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'real_proofs.json')))
    rng = random.Random(args.seed)
    dev = torch.device('cuda', 0)
    os.environ['ZKV_AGG_MIN'] = '64'
    r, s = g['risc0'], g['sp1']
    pools = {}
    for vm, base in (('risc0', H(r['seal'])), ('sp1', H(s['proof']))):
        seals, mut, mclass, flip = synth.make_batch(vm, base, 4096, 0x5A4B56F0 + len(pools), pool=16, mutate_every=2)
        pools[vm] = (seals, mut, flip)
    v0 = zkv.RiscZeroVerifier(); v0.initialize(H(r['control_root']), H(r['bn254_control_id']))
    v1 = zkv.Sp1Verifier()
    stats = {'batches': 0, 'proofs': 0, 'rejected': 0, 'sub_batches_failed': 0, 'mismatches': 0}
    t_end = time.time() + args.seconds
    last = time.time()
    while time.time() < t_end:
        vm = rng.choice(('risc0', 'sp1'))
        seals, mut, flip = pools[vm]
        n = rng.choice((64, 65, 100, 127, 128, 129, 1000, 4097, rng.randrange(64, 30000), rng.randrange(64, 3000)))
        density = rng.choice((0.0, 0.001, 0.01, 0.05, 0.3, 1.0))
        good = np.flatnonzero(~mut); bad = np.flatnonzero(mut)
        pick = np.where(np.array([rng.random() < density for _ in range(n)]), np.array([bad[rng.randrange(len(bad))] for _ in range(n)]),
                        np.array([good[rng.randrange(len(good))] for _ in range(n)]))
        if rng.random() < 0.3:                                    # clustered rejects: sort so that the rejects sit together somewhere
            order = np.argsort(~mut[pick], kind='stable'); k = rng.randrange(n); pick = np.roll(pick[order], k)
        S = np.ascontiguousarray(seals[pick]); m = mut[pick]; fl = flip[pick]
        if vm == 'risc0':
            A = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n, 1)); B = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n, 1)); B[fl, 0] ^= 1
        else:
            A = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (n, 1)); B = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (n, 1)); B[fl, -1] ^= 1
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (S, A, B)]
        v = v0 if vm == 'risc0' else v1
        def run():
            st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
            if vm == 'risc0': v.verify_batch_dev(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
            else: v.verify_batch_dev(n, d[1].data_ptr(), d[2].data_ptr(), 96, d[0].data_ptr(), st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            return st.cpu().numpy()
        v.set_aggregate_check(False)
        plain = run()
        os.environ['ZKV_AGG_GROUP'] = str(rng.choice((1, 2, 4, 8)))
        v.set_aggregate_check(True, seed=rng.randbytes(32) if rng.random() < 0.5 else None, sub_batch=rng.choice((16, 32, 64, 128, 256)))
        c0 = v.aggregate_counters()
        agg = run()
        c1 = v.aggregate_counters()
        bad_n = int((agg != plain).sum()) + int((((plain == 0) != ~m)).sum())
        stats['batches'] += 1; stats['proofs'] += n; stats['rejected'] += int(m.sum()); stats['sub_batches_failed'] += c1[1] - c0[1]; stats['mismatches'] += bad_n
        if bad_n:
            print('MISMATCH', vm, n, density, os.environ['ZKV_AGG_GROUP'], flush=True)
        if time.time() - last > 50:
            print(json.dumps(stats), flush=True); last = time.time()
    print(json.dumps(dict(stats, seconds=args.seconds, verdict='ok' if stats['mismatches'] == 0 else 'MISMATCH')), flush=True)
    sys.exit(0 if stats['mismatches'] == 0 else 1)


if __name__ == '__main__':
    main()
