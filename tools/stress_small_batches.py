#!/usr/bin/env python3
"""Soak test of the small-batch kernels (two wavefronts per proof, one wavefront per proof, 16 lanes per proof): random slices of a pool
of 8,192 synthetic RISC Zero and SP1 proofs (1/5 damaged, every mutation class) of random sizes 1..2,200 through the automatic
mapping, for --seconds; every status must equal what the lane-pair kernels returned for the same proof.  Looks for rare races in the
producer / consumer hand-over of k_miller_w64d (LDS table + step counter); a lost hand-over would fail the proof, never hang."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = bytes.fromhex


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--seconds', type=float, default=120.0); a = ap.parse_args()
    from stylus_zkvm_verifiers_amd import synth
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'real_proofs.json')))
    r, s = g['risc0'], g['sp1']
    N = 8192
    s0, m0, _, f0 = synth.make_batch('risc0', H(r['seal']), N, 0x5A4B56E1, pool=8, mutate_every=5)
    s1, m1, _, f1 = synth.make_batch('sp1', H(s['proof']), N, 0x5A4B56E2, pool=8, mutate_every=5)
    import torch
    import stylus_zkvm_verifiers_amd as z
    dev = torch.device('cuda', 0)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (N, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (N, 1)); jds[f0, 0] ^= 1
    vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (N, 1))
    pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (N, 1)); pv[f1, -1] ^= 1
    up = lambda *xs: [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in xs]
    d0, d1 = up(s0, ids, jds), up(vk, pv, s1)
    v0 = z.RiscZeroVerifier(0); v0.initialize(H(r['control_root']), H(r['bn254_control_id']))
    v1 = z.Sp1Verifier(0)
    stream = torch.cuda.current_stream().cuda_stream
    st = torch.full((N,), 255, dtype=torch.uint8, device=dev)
    ref = []
    for v, d, kind in ((v0, d0, 0), (v1, d1, 1)):
        v.set_lanes_per_proof(2)
        st.fill_(255)
        if kind == 0: v.verify_batch_dev(N, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), st.data_ptr(), 0, stream)
        else: v.verify_batch_dev(N, d[0].data_ptr(), d[1].data_ptr(), 96, d[2].data_ptr(), st.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        ref.append(st.cpu().numpy().copy())
        v.set_lanes_per_proof(0)
    assert ((ref[0] == 0) == ~m0).all() and ((ref[1] == 0) == ~m1).all()
    rng = np.random.default_rng(0x5A4B56E3)
    t0 = time.time(); it = 0; proofs = 0; by = {'two_wavefronts': 0, 'one_wavefront': 0, 'sixteen_lanes': 0}
    while time.time() - t0 < a.seconds:
        n = int(rng.integers(1, 2201)); off = int(rng.integers(0, N - n + 1)); kind = int(rng.integers(0, 2))
        st[:n].fill_(255)
        if kind == 0:
            v0.verify_batch_dev(n, d0[0][off:].data_ptr(), d0[1][off:].data_ptr(), d0[2][off:].data_ptr(), st.data_ptr(), 0, stream)
        else:
            v1.verify_batch_dev(n, d1[0][off:].data_ptr(), d1[1][off:].data_ptr(), 96, d1[2][off:].data_ptr(), st.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        got = st[:n].cpu().numpy()
        if not (got == ref[kind][off:off + n]).all():
            bad = np.nonzero(got != ref[kind][off:off + n])[0]
            print(json.dumps({'ok': False, 'iteration': it, 'n': n, 'off': off, 'vm': kind, 'first_bad': int(bad[0]), 'got': int(got[bad[0]]), 'want': int(ref[kind][off + bad[0]])}))
            sys.exit(1)
        it += 1; proofs += n
        by['two_wavefronts' if n <= 768 else 'one_wavefront' if n <= 2048 else 'sixteen_lanes'] += 1
    print(json.dumps({'ok': True, 'iterations': it, 'proofs': proofs, 'seconds': round(time.time() - t0, 1), 'batches_by_mapping': by}))


if __name__ == '__main__':
    main()
