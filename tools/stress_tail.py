#!/usr/bin/env python3
"""Soak test of the tail split (zkv_capi.hip tail_of_chunk): chunks of 32,768 k + r proofs whose last r proofs run through the small-batch
kernels beside (k odd) or after (k even) the lane-pair kernels of the others.  Random k in 1..4 and r in 1..12,288, calls enqueued back to
back on two caller streams without synchronisation in between, for --seconds; every status must equal the construction's (accept <=> the
proof was not damaged).  Looks for races between the context's second stream and the next call's kernels on the shared workspace."""
import argparse, json, os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = bytes.fromhex


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--seconds', type=float, default=60.0); ap.add_argument('--seed', type=int, default=7); a = ap.parse_args()
    from stylus_zkvm_verifiers_amd import synth
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'real_proofs.json')))
    r = g['risc0']
    N = 4 * 32768 + 12288
    seals, mut, _, flip = synth.make_batch_parallel('risc0', H(r['seal']), N, 0x7A11, mutate_every=7)
    import torch
    import stylus_zkvm_verifiers_amd as z
    dev = torch.device('cuda', 0)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (N, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (N, 1)); jds[flip, 0] ^= 1
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (seals, ids, jds)]
    v = z.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    rng = random.Random(a.seed)
    t0 = time.time(); calls = proofs = bad = 0
    while time.time() - t0 < a.seconds:
        batch = []
        for j in range(rng.randrange(1, 4)):                      # up to three calls in flight on alternating streams
            k = rng.randrange(1, 5); rr = rng.choice([1, 33, 700, 2000, 2048, 3072, 4096, 4097, 8192, 8193, 12288]) if rng.random() < 0.5 else rng.randrange(1, 12289)
            n = 32768 * k + rr
            off = rng.randrange(0, N - n + 1)
            st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
            s = streams[j & 1]
            v.verify_batch_dev(n, d[0][off:].data_ptr(), d[1][off:].data_ptr(), d[2][off:].data_ptr(), st.data_ptr(), 0, s.cuda_stream)
            batch.append((n, off, st))
        torch.cuda.synchronize()
        for n, off, st in batch:
            got = st.cpu().numpy()
            bad += int(((got == 0) != ~mut[off:off + n]).sum()); calls += 1; proofs += n
    print(json.dumps({'calls': calls, 'proofs': proofs, 'mismatches': bad, 'seconds': round(time.time() - t0, 1), 'verdict': 'ok' if bad == 0 else 'FAILED'}))
    sys.exit(0 if bad == 0 else 1)


if __name__ == '__main__':
    main()
