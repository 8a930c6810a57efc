"""Parity of the HIP path (through the C ABI) against the golden vectors and the CPU oracle.  Needs an MI355X.

Pinned by the reference: the two real proofs (examples/*/examples/interact.rs) must ACCEPT.
Everything else is agreement with the oracle / spec model => "parity unpinned" (SURVEY.md 8c)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
H = bytes.fromhex


@pytest.fixture(scope='module')
def zkv():
    import stylus_zkvm_verifiers_amd as z
    assert z.device_count() >= 1, 'no gfx950 device visible'
    return z


@pytest.fixture(scope='module')
def r0(zkv, real_proofs):
    v = zkv.RiscZeroVerifier()
    v.initialize(H(real_proofs['risc0']['control_root']), H(real_proofs['risc0']['bn254_control_id']))
    yield v
    v.close()


@pytest.fixture(scope='module')
def sp1(zkv):
    v = zkv.Sp1Verifier()
    yield v
    v.close()


def test_real_risc0_proof_accepts(r0, real_proofs):
    r = real_proofs['risc0']
    assert r0.get_selector().hex() == r['selector']
    assert r0.verify(H(r['seal']), H(r['image_id']), H(r['journal_digest'])) is True
    assert r0.verify_integrity(H(r['seal']), H(r['claim_digest'])) is True


def test_real_sp1_proof_accepts(sp1, real_proofs):
    s = real_proofs['sp1']
    assert sp1.verify_proof(H(s['vkey']), H(s['public_values']), H(s['proof'])) is None


def test_verify_corpus_matches_golden_and_oracle(zkv, r0, sp1, verify_corpus):
    import oracle_lib as ol
    orc = ol.Risc0Oracle()
    orc.initialize(H(verify_corpus['risc0_ctx']['control_root']), H(verify_corpus['risc0_ctx']['bn254_control_id']))
    rc = [c for c in verify_corpus['cases'] if c['vm'] == 'risc0']
    sc = [c for c in verify_corpus['cases'] if c['vm'] == 'sp1']
    st, rv = r0.verify_batch([H(c['seal']) for c in rc], [H(c['image_id']) for c in rc], [H(c['journal_digest']) for c in rc])
    for c, s, r in zip(rc, st, rv):
        ost, orecv = orc.verify(H(c['seal']), H(c['image_id']), H(c['journal_digest']))
        assert int(s) == c['status'] == ost, c['name']
        assert bytes(r).hex() == (c['received'] or '00000000'), c['name']
    st, rv = sp1.verify_batch([H(c['vkey']) for c in sc], [H(c['public_values']) for c in sc], [H(c['proof']) for c in sc])
    for c, s, r in zip(sc, st, rv):
        ost, orecv = ol.sp1_verify_proof(H(c['vkey']), H(c['public_values']), H(c['proof']))
        assert int(s) == c['status'] == ost, c['name']
        assert bytes(r).hex() == (c['received'] or '00000000'), c['name']


def test_single_call_errors_carry_revert_bytes(zkv, r0, sp1, verify_corpus, revert_vectors):
    for c in verify_corpus['cases']:
        if c['status'] == 0:
            continue
        with pytest.raises(zkv.VerifierError) as ei:
            if c['vm'] == 'risc0':
                r0.verify(H(c['seal']), H(c['image_id']), H(c['journal_digest']))
            else:
                sp1.verify_proof(H(c['vkey']), H(c['public_values']), H(c['proof']))
        assert ei.value.status == c['status'], c['name']
        if c['status'] == 5:
            assert ei.value.received.hex() == c['received']
            assert ei.value.revert[4:8].hex() == c['received']


def test_context_cases(zkv, verify_corpus):
    for c in verify_corpus['ctx_cases']:
        v = zkv.RiscZeroVerifier()
        if c['name'] == 'second initialize':
            cr = H(verify_corpus['risc0_ctx']['control_root']); cid = H(verify_corpus['risc0_ctx']['bn254_control_id'])
            v.initialize(cr, cid)
            with pytest.raises(zkv.VerifierError) as ei:
                v.initialize(cr, cid)
            assert ei.value.status == c['status']
            continue
        if c['control_root'] is not None:
            v.initialize(H(c['control_root']), H(c['bn254_control_id']))
            assert v.get_selector().hex() == c['selector']
        st, _ = v.verify_batch([H(c['seal'])], [H(c['image_id'])], [H(c['journal_digest'])])
        assert int(st[0]) == c['status'], c['name']
        v.close()


def test_no_wavefront_ever_gave_up_waiting(zkv, r0, real_proofs):
    """zkv_diag_wait_faults: the two-wavefront kernels fail closed when a consumer's bounded wait for its producer runs out, and count it.
    After a few hundred single and small-batch verifications (the kernels in question) the count is still zero: every reject in this suite
    is a verdict, none a timeout."""
    import ctypes as C
    from stylus_zkvm_verifiers_amd import _lib
    r = real_proofs['risc0']
    for _ in range(8):
        assert r0.verify(H(r['seal']), H(r['image_id']), H(r['journal_digest'])) is True
    st, _ = r0.verify_batch([H(r['seal'])] * 300, [H(r['image_id'])] * 300, [H(r['journal_digest'])] * 300)
    assert (st == 0).all()
    out = C.c_uint64(123)
    _lib.check(_lib.lib().zkv_diag_wait_faults(0, C.byref(out)), 'zkv_diag_wait_faults')
    assert out.value == 0


def test_empty_batch(r0, sp1):
    st, rv = r0.verify_batch([], [], [])
    assert len(st) == 0
    st, rv = sp1.verify_batch([], [], [])
    assert len(st) == 0


def test_precompile_seam_matches_kats_and_oracle(zkv, precompile_kats):
    """Inner seam: batched ecAdd / ecMul / ecPairing with EIP-196/197 semantics against the golden KATs
    (spec model) and the C oracle on seeded random inputs."""
    import random
    import oracle_lib as ol
    pc = zkv.Bn254Precompiles()
    adds = [H(c['input']) for c in precompile_kats['ecadd'] if len(c['input']) == 256]
    want = [c['output'] for c in precompile_kats['ecadd'] if len(c['input']) == 256]
    got = pc.ecadd(adds)
    assert [g.hex() if g is not None else None for g in got] == want
    muls = [c for c in precompile_kats['ecmul']]
    got = pc.ecmul([H(c['input']) for c in muls])
    assert [g.hex() if g is not None else None for g in got] == [c['output'] for c in muls]
    for dual in ('768', '0'):             # few calls: one call per workgroup of two wavefronts (k_pairing_pair_w64d); '0': the lane-pair kernels
        os.environ['ZKV_DUAL_BELOW'] = dual
        for k in (1, 2):
            cases = [c for c in precompile_kats['pairing'] if len(c['input']) == 384 * k]
            got = pc.pairing([H(c['input']) for c in cases], k)
            for c, g in zip(cases, got):
                exp = None if c['output'] is None else (int(c['output'], 16) != 0)
                assert g == exp, (dual, c['name'])
        assert pc.pairing([b''], 0) == [True]                       # empty input: product of no pairings is 1
    del os.environ['ZKV_DUAL_BELOW']
    # random ecMul / ecAdd against the C oracle
    rng = random.Random(99)
    G = (1).to_bytes(32, 'big') + (2).to_bytes(32, 'big')
    pts = [ol.ecmul(G + rng.randrange(1 << 256).to_bytes(32, 'big')) for _ in range(24)]
    muls = [p + rng.randrange(1 << 256).to_bytes(32, 'big') for p in pts]
    assert pc.ecmul(muls) == [ol.ecmul(m) for m in muls]
    adds = [pts[i] + pts[(i * 5 + 1) % len(pts)] for i in range(len(pts))] + [pts[0] + pts[0], pts[1] + bytes(64)]
    assert pc.ecadd(adds) == [ol.ecadd(a) for a in adds]
    pc.close()


def test_groth16_pairing_through_the_precompile_seam(zkv, real_proofs):
    """The reference's own 768-byte ecPairing calldata for the real RISC Zero proof (groth16.rs:109-119) -> 1."""
    import oracle_lib as ol
    import spec_model as m
    r = real_proofs['risc0']
    seal = H(r['seal'])
    w = [seal[4 + 32 * i:36 + 32 * i] for i in range(8)]
    ax, ay = m.negate_g1_words(int.from_bytes(w[0], 'big'), int.from_bytes(w[1], 'big'))
    vk = m.RISC0_VK
    g2 = lambda q: b''.join(m.be32(v) for v in (q[0][0], q[0][1], q[1][0], q[1][1]))
    data = (m.be32(ax) + m.be32(ay) + b''.join(w[2:6]) + m.be32(vk['alpha1'][0]) + m.be32(vk['alpha1'][1]) + g2(vk['beta2'])
            + H(r['vk_x'][0]) + H(r['vk_x'][1]) + g2(vk['gamma2']) + w[6] + w[7] + g2(vk['delta2']))
    pc = zkv.Bn254Precompiles()
    bad = bytearray(data); bad[100] ^= 1
    inf_g1 = bytes(64) + data[64:]                           # first pair with A = infinity: still a valid input, the product changes
    inf_g2 = data[:64] + bytes(128) + data[192:]
    off_g2 = data[:64] + data[64:100] + bytes([data[100] ^ 1]) + data[101:]
    want = lambda x: (lambda r: None if r is None else bool(r[-1]))(ol.ecpairing(x))
    for dual in ('768', '0'):             # the two-wavefront kernels of small batches, then the lane-pair kernels
        os.environ['ZKV_DUAL_BELOW'] = dual
        assert pc.pairing([data], 4) == [True], dual
        batch = [bytes(bad), inf_g1, inf_g2, off_g2, data, data[:192] + data[:192] + data[384:]]
        assert pc.pairing(batch, 4) == [want(x) for x in batch], dual
    del os.environ['ZKV_DUAL_BELOW']
    # calls of 1, 2, 3, 4 and 9 pairs cut from the same inputs (2 .. 8 pairs run ONE Miller loop per call, 1 and 9 one loop per pair), a batch
    # large enough for the lane-pair kernels, host buffers and device-resident buffers: equal to the oracle
    import torch
    dev = torch.device('cuda', 0)
    base = [bytes(bad), inf_g1, inf_g2, off_g2, data, data[:192] + data[:192] + data[384:], data[192:] + data[:192]]
    for k in (4, 1, 2, 3, 9):
        calls = [(x * 3)[:192 * k] for x in base] * 150          # 1,050 calls
        exp = [want(x) for x in calls[:len(base)]] * 150
        assert pc.pairing(calls, k) == exp, k
        n = len(calls)
        d_in = torch.from_numpy(np.frombuffer(b''.join(calls), dtype=np.uint8).copy()).to(dev)
        d_res = torch.full((n,), 255, dtype=torch.uint8, device=dev); d_ok = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        pc.pairing_dev(n, k, d_in.data_ptr(), d_res.data_ptr(), d_ok.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        res, ok = d_res.cpu().numpy(), d_ok.cpu().numpy()
        assert [bool(res[i]) if ok[i] else None for i in range(n)] == exp, k
    pc.close()


def test_device_resident_fast_path_matches_oracle(zkv, r0, sp1, real_proofs):
    """Fixed-stride, HBM-resident entry points (coalesced LDS staging of the seals) on a seeded synthetic batch with every
    mutation class: statuses equal the CPU oracle and the ragged host-pointer path; partial last wavefront included."""
    import torch
    import oracle_lib as ol
    from stylus_zkvm_verifiers_amd import synth
    dev = torch.device('cuda', 0)
    r = real_proofs['risc0']
    n = 200                                                    # not a multiple of 64
    seals, mut, mclass, flip = synth.make_batch('risc0', H(r['seal']), n, 0x5A4B5611, pool=4, mutate_every=5)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n, 1)); jds[flip, 0] ^= 1
    d_seals, d_ids, d_jds = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (seals, ids, jds))
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev); d_rv = torch.zeros((n, 4), dtype=torch.uint8, device=dev)
    r0.verify_batch_dev(n, d_seals.data_ptr(), d_ids.data_ptr(), d_jds.data_ptr(), d_st.data_ptr(), d_rv.data_ptr(),
                        torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy(); rv = d_rv.cpu().numpy()
    orc = ol.Risc0Oracle(); orc.initialize(H(r['control_root']), H(r['bn254_control_id']))
    ost, orv = orc.verify_batch([x.tobytes() for x in seals], [x.tobytes() for x in ids], [x.tobytes() for x in jds], threads=8)
    assert (st == ost).all()
    assert (rv.reshape(-1) == orv).all()
    hst, hrv = r0.verify_batch([x.tobytes() for x in seals], [x.tobytes() for x in ids], [x.tobytes() for x in jds])
    assert (hst == st).all() and (hrv == rv).all()
    assert ((st == 0) == ~mut).all() and set(st[mut]) == {1, 5}
    # SP1, 96-byte public values
    s = real_proofs['sp1']
    n = 130
    proofs, mut, mclass, flip = synth.make_batch('sp1', H(s['proof']), n, 0x5A4B5612, pool=4, mutate_every=5)
    vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (n, 1))
    pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (n, 1)); pv[flip, -1] ^= 1
    d_p, d_vk, d_pv = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (proofs, vk, pv))
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    sp1.verify_batch_dev(n, d_vk.data_ptr(), d_pv.data_ptr(), 96, d_p.data_ptr(), d_st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy()
    ost, _ = ol.sp1_verify_batch([x.tobytes() for x in vk], [x.tobytes() for x in pv], [x.tobytes() for x in proofs], threads=8)
    assert (st == ost).all() and ((st == 0) == ~mut).all()


def test_both_kernel_mappings_agree_on_a_2p13_batch(zkv, real_proofs):
    """8,192 seeded proofs (every mutation class, 1/16 mutated) through the lane-pair kernels and the 16-lanes-per-proof kernels:
    identical statuses, accept <=> not mutated on the whole batch, and equality with the CPU oracle on a 1,024-proof sample."""
    import torch
    import oracle_lib as ol
    from stylus_zkvm_verifiers_amd import synth
    dev = torch.device('cuda', 0)
    r = real_proofs['risc0']
    n = 1 << 13
    seals, mut, mclass, flip = synth.make_batch('risc0', H(r['seal']), n, 0x5A4B5621, pool=8, mutate_every=16)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n, 1)); jds[flip, 0] ^= 1
    d_seals, d_ids, d_jds = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (seals, ids, jds))
    out = {}
    for lanes in (2, 16, 64, 128):
        v = zkv.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id'])); v.set_lanes_per_proof(lanes)
        d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        v.verify_batch_dev(n, d_seals.data_ptr(), d_ids.data_ptr(), d_jds.data_ptr(), d_st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out[lanes] = d_st.cpu().numpy()
        v.close()
    assert (out[16] == out[2]).all() and (out[64] == out[2]).all() and (out[128] == out[2]).all()
    assert ((out[2] == 0) == ~mut).all()
    k = 1024
    orc = ol.Risc0Oracle(); orc.initialize(H(r['control_root']), H(r['bn254_control_id']))
    ost, _ = orc.verify_batch([x.tobytes() for x in seals[:k]], [x.tobytes() for x in ids[:k]], [x.tobytes() for x in jds[:k]], threads=8)
    assert (out[2][:k] == ost).all()


def test_both_window_widths_of_the_vk_x_stage_agree(zkv, real_proofs, monkeypatch):
    """The vk_x stage of a batch above ZKV_MSM_WAVE_BELOW walks 16-bit window rows by default and the 8-bit rows with
    ZKV_MSM_WINDOW_BITS=8 (read when a context is set up): the same statuses on 8,192 seeded RISC Zero proofs (accept <=> untouched),
    and compute_vk_x itself byte for byte on random signals, all different."""
    import random
    import torch
    import spec_model as m
    from stylus_zkvm_verifiers_amd import synth
    dev = torch.device('cuda', 0)
    r = real_proofs['risc0']
    n = 1 << 13
    seals, mut, mclass, flip = synth.make_batch('risc0', H(r['seal']), n, 0x16B175, pool=8, mutate_every=16)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n, 1)); jds[flip, 0] ^= 1
    d_seals, d_ids, d_jds = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (seals, ids, jds))
    rng = random.Random(99)
    pairs = [(m.be32(rng.randrange(1 << 128)), m.be32(rng.randrange(1 << 128))) for _ in range(64)]
    out, vkx = {}, {}
    for bits in ('16', '8'):
        monkeypatch.setenv('ZKV_MSM_WINDOW_BITS', bits)
        v = zkv.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
        d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        v.verify_batch_dev(n, d_seals.data_ptr(), d_ids.data_ptr(), d_jds.data_ptr(), d_st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out[bits] = d_st.cpu().numpy()
        vkx[bits] = list(v.vk_x_batch(pairs))
        v.close()
    assert (out['16'] == out['8']).all() and ((out['16'] == 0) == ~mut).all()
    assert vkx['16'] == vkx['8'] and len(set(vkx['16'])) == len(pairs)


def test_entries_of_every_16_bit_window_row_against_the_oracle(zkv, r0, sp1, real_proofs):
    """The device-built 16-bit window rows entry by entry: a signal that is one 16-bit digit d at window w makes compute_vk_x return
    base + d * 65536^w * IC_b, i.e. exactly one table entry added to the base point.  Per row: the corners of the 64-entry build
    chunks (lower byte 0, 1, 63, 64, 255; upper byte 0, 1, 255) and 600 random digits, against the oracle's ecMul / ecAdd chain."""
    import random
    import oracle_lib as ol
    import spec_model as m
    rng = random.Random(1616)
    fixed = [H(x) for x in real_proofs['risc0']['signals']]
    for vm, ctx, bits in ((0, r0, 128), (1, sp1, 256)):
        sigs = []
        for b in range(2):
            for w in range(bits // 16):
                digs = [(hi << 8) | lo for hi in (0, 1, 255) for lo in (0, 1, 63, 64, 255)] + [rng.randrange(1, 1 << 16) for _ in range(600)]
                for d in digs:
                    v = d << (16 * w)
                    if vm == 1 and v >= (m.R if b == 0 else 1 << 253):
                        continue                                  # not a signal the SP1 entry point can carry
                    sigs.append((v, 0) if b == 0 else (0, v))
        got = ctx.vk_x_batch([(m.be32(a), m.be32(b)) for a, b in sigs])
        for (a, b), g in zip(sigs, got):
            want = ol.groth16_vk_x(0, [fixed[0], fixed[1], m.be32(a), m.be32(b), fixed[4]]) if vm == 0 else ol.groth16_vk_x(1, [m.be32(a), m.be32(b)])
            assert g == want, (vm, hex(a), hex(b))


def test_vk_x_batch_matches_oracle(zkv, r0, sp1, real_proofs):
    """compute_vk_x on the GPU (windowed fixed-base tables built by the set-up kernels) against the oracle's ecMul/ecAdd chain
    for random and edge-case signals -- the valid proofs of the corpus all share one set of public inputs."""
    import random
    import oracle_lib as ol
    import spec_model as m
    r = real_proofs['risc0']
    fixed = [H(x) for x in r['signals']]
    rng = random.Random(12)
    pairs = [(0, 0), (1, 0), (0, 1), ((1 << 128) - 1, (1 << 128) - 1), (15, 1 << 124)] + [(rng.randrange(1 << 128), rng.randrange(1 << 128)) for _ in range(120)]
    got = r0.vk_x_batch([(m.be32(a), m.be32(b)) for a, b in pairs])
    for (a, b), g in zip(pairs, got):
        assert g == ol.groth16_vk_x(0, [fixed[0], fixed[1], m.be32(a), m.be32(b), fixed[4]]), (hex(a), hex(b))
    assert got[len(pairs) - 1] != got[0]
    pairs = [(0, 0), (1, 1), (m.R - 1, m.R - 1), (m.R - 1, 0), (1 << 252, (1 << 253) - 1)] + [(rng.randrange(m.R), rng.randrange(1 << 253)) for _ in range(120)]
    got = sp1.vk_x_batch([(m.be32(a), m.be32(b)) for a, b in pairs])
    for (a, b), g in zip(pairs, got):
        assert g == ol.groth16_vk_x(1, [m.be32(a), m.be32(b)]), (hex(a), hex(b))


def test_generic_groth16_trapdoor_keys_on_gpu(zkv):
    """zkv_groth16_verify_batch (verify_proof_with_key for an arbitrary key): valid proofs with distinct public inputs for keys with
    1..6 IC points, wrong / out-of-range signals, the other VM convention, keys with infinity or invalid points -- equal to the
    spec model and the C oracle."""
    import random
    import oracle_lib as ol
    import spec_model as m
    from trapdoor_cases import generic_cases
    rng = random.Random(78)
    cases = generic_cases(rng)
    # group by (vm, key) so that each context verifies a small batch
    groups = {}
    for name, vm, vk, prf, sig, expect in cases:
        groups.setdefault((vm, m.vk_to_words(vk), len(vk['ic'])), []).append((name, prf, sig, expect, vk))
    for (vm, vkb, n_ic), items in groups.items():
        v = zkv.Groth16Verifier(vkb, n_ic, zkv.errors.VM_RISC0 if vm == 'risc0' else zkv.errors.VM_SP1)
        got = v.verify_batch([m.proof_to_words(*it[1]) for it in items], [[m.be32(s) for s in it[2]] for it in items])
        for it, g in zip(items, got):
            name, prf, sig, expect, vk = it
            assert bool(g) == expect == ol.groth16_verify_vk(0 if vm == 'risc0' else 1, vkb, n_ic, m.proof_to_words(*prf), [m.be32(s) for s in sig]), name
        v.close()
    # a larger batch of distinct public inputs on one key
    vk, td = m.trapdoor_vk(rng, 6)
    proofs, sigs, exp = [], [], []
    for i in range(96):
        sig = [rng.randrange(m.R) for _ in range(5)]
        prf = m.trapdoor_prove(rng, td, sig, 'risc0')
        if i % 5 == 4:
            sig[i % 5 - 1] ^= 1
        proofs.append(m.proof_to_words(*prf)); sigs.append([m.be32(s) for s in sig]); exp.append(i % 5 != 4)
    v = zkv.Groth16Verifier(m.vk_to_words(vk), 6, zkv.errors.VM_RISC0)
    assert list(v.verify_batch(proofs, sigs)) == exp
    A, B, Cc = m.trapdoor_prove(rng, td, [1, 2, 3, 4, 5], 'risc0')
    assert v.verify_proof_with_key(A, B, Cc, [1, 2, 3, 4, 5]) is True
    assert v.verify_proof_with_key(A, B, Cc, [1, 2, 3, 4]) is False          # length mismatch, groth16.rs:32
    v.close()


def test_wire_layer_cases_match_golden_and_oracle(zkv, r0, sp1, wire_cases):
    """eth_call calldata decoded on the device (csrc/k_wire.hip): return / revert data and status of every golden case equal
    the fixture (spec model) and the C oracle.  The whole layer is UNPINNED by the reference (the Stylus router is a
    dependency): it is pinned by the ABI specification and the two real proofs travelling through it."""
    import oracle_lib as ol
    from wire_util import calldata_of
    enc = (zkv.wire.encode_risc0_verify, zkv.wire.encode_risc0_verify_integrity, zkv.wire.encode_sp1_verify_proof)
    new = zkv.RiscZeroVerifier()
    ctx = wire_cases['risc0_ctx']
    o_init = ol.Risc0Oracle(); o_init.initialize(H(ctx['control_root']), H(ctx['bn254_control_id']))
    o_new = ol.Risc0Oracle()
    groups = {('risc0', 'init'): (r0, o_init.eth_call), ('risc0', 'new'): (new, o_new.eth_call), ('sp1', 'init'): (sp1, ol.sp1_eth_call)}
    for key, (ver, oracle_call) in groups.items():
        cases = [c for c in wire_cases['cases'] if (c['vm'], c['ctx']) == key]
        cds = [calldata_of(c, *enc) for c in cases]
        for order, host_chunk in ((cds, None), (cds[::-1], None), (cds, '7')):
            # both orders: request offsets get every alignment; 7-request chunks: the double-buffered H2D pipeline rotates
            if host_chunk:
                os.environ['ZKV_WIRE_HOST_CHUNK'] = host_chunk
            try:
                rev, ret, st = zkv.wire.eth_call_batch(ver, order)
            finally:
                os.environ.pop('ZKV_WIRE_HOST_CHUNK', None)
            cs = cases[::-1] if order is not cds else cases
            for c, cd, rv, rd, s in zip(cs, order, rev, ret, st):
                assert (bool(rv), rd.hex()) == (c['reverted'], c['returndata']), (key, c['name'])
                assert int(s) == (6 if c['status'] is None else c['status']), (key, c['name'])
                orev, oret, ost = oracle_call(cd)
                assert (bool(rv), rd) == (orev, oret), (key, c['name'])
    new.close()


def test_wire_layer_device_fast_path_2p12(zkv, r0, sp1, real_proofs):
    """4,096 seeded eth_calls per verifier with calldata resident in HBM: statuses equal the fixed-stride seal path on the
    same proofs (which the other tests tie to the oracle), calldata-level damage is reported as BAD_CALLDATA, and a sample
    equals the oracle's eth_call."""
    import torch
    import oracle_lib as ol
    from stylus_zkvm_verifiers_amd import synth
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream().cuda_stream
    r = real_proofs['risc0']
    n = 1 << 12
    seals, mut, mclass, flip = synth.make_batch('risc0', H(r['seal']), n, 0x5A4B5631, pool=8, mutate_every=8)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n, 1)); jds[flip, 0] ^= 1
    cd = synth.calldata_risc0_verify(seals, ids, jds)
    bad = np.zeros(n, dtype=bool)
    bad[5::97] = True
    cd[5::97, 132 + 32 * 17 + 9] = 0x40                        # element 17 is not a uint8
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(cd.shape[1])
    d_cd, d_off = torch.from_numpy(cd).to(dev), torch.from_numpy(off.view(np.int64)).to(dev)
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev); d_rv = torch.zeros((n, 4), dtype=torch.uint8, device=dev)
    zkv.wire.eth_call_batch_dev(r0, n, d_cd.data_ptr(), d_off.data_ptr(), cd.size, d_st.data_ptr(), d_rv.data_ptr(), stream)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy()
    d_seals, d_ids, d_jds = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (seals, ids, jds))
    d_st2 = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    r0.verify_batch_dev(n, d_seals.data_ptr(), d_ids.data_ptr(), d_jds.data_ptr(), d_st2.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    st2 = d_st2.cpu().numpy()
    assert (st[bad] == 6).all() and (st[~bad] == st2[~bad]).all()
    assert ((st == 0) == (~mut & ~bad)).all()
    assert zkv.wire.last_wire_ms(r0) > 0
    orc = ol.Risc0Oracle(); orc.initialize(H(r['control_root']), H(r['bn254_control_id']))
    for i in list(range(0, 24)) + [102, 199]:
        orev, oret, ost = orc.eth_call(cd[i].tobytes())
        assert int(st[i]) == ost, i
    # SP1
    s = real_proofs['sp1']
    proofs, mut, mclass, flip = synth.make_batch('sp1', H(s['proof']), n, 0x5A4B5632, pool=8, mutate_every=8)
    vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (n, 1))
    pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (n, 1)); pv[flip, -1] ^= 1
    cd = synth.calldata_sp1_verify_proof(vk, pv, proofs)
    bad = np.zeros(n, dtype=bool)
    bad[7::101] = True
    cd[7::101, 68 + 31] ^= 0x20                                # second offset no longer canonical
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(cd.shape[1])
    d_cd, d_off = torch.from_numpy(cd).to(dev), torch.from_numpy(off.view(np.int64)).to(dev)
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    zkv.wire.eth_call_batch_dev(sp1, n, d_cd.data_ptr(), d_off.data_ptr(), cd.size, d_st.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy()
    assert (st[bad] == 6).all() and ((st == 0) == (~mut & ~bad)).all()
    for i in list(range(0, 16)) + [108]:
        orev, oret, ost = ol.sp1_eth_call(cd[i].tobytes())
        assert int(st[i]) == ost, i


def test_dev_calls_on_different_streams_do_not_race(zkv, r0, real_proofs):
    """Two device-resident batches enqueued back to back on two HIP streams share the context's workspace: the second must
    wait for the first (event ordering inside the library), so both status vectors are right."""
    import torch
    from stylus_zkvm_verifiers_amd import synth
    dev = torch.device('cuda', 0)
    r = real_proofs['risc0']
    n = 1 << 12
    batches = []
    for seed, every in ((0x5A4B5641, 3), (0x5A4B5642, 5)):
        seals, mut, mclass, flip = synth.make_batch('risc0', H(r['seal']), n, seed, pool=4, mutate_every=every)
        ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n, 1))
        jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n, 1)); jds[flip, 0] ^= 1
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (seals, ids, jds)]
        batches.append((d, mut, torch.full((n,), 255, dtype=torch.uint8, device=dev)))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    for rep in range(3):
        for (d, mut, d_st), st in zip(batches, streams):
            r0.verify_batch_dev(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_st.data_ptr(), 0, st.cuda_stream)
    torch.cuda.synchronize()
    for d, mut, d_st in batches:
        st = d_st.cpu().numpy()
        assert ((st == 0) == ~mut).all()


def test_full_size_configs_through_properties(zkv, r0, sp1, real_proofs):
    """BASELINE.json's single-GPU sizes (2^16 RISC Zero proofs, 2^20 SP1 proofs -- one 2^20-proof chunk) checked through
    size-independent properties: accept <=> not mutated by construction for every proof, permutation equivariance (the batch
    is a seeded shuffle of copies of a 2^12 base batch whose statuses the oracle-pinned tests cover: status[i] must equal
    base_status[source[i]]), and idempotence (a second run returns the same bytes)."""
    import torch
    from stylus_zkvm_verifiers_amd import synth
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream().cuda_stream
    nb = 1 << 12

    def run_r0(seals, ids, jds):
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (seals, ids, jds)]
        d_st = torch.full((len(seals),), 255, dtype=torch.uint8, device=dev)
        r0.verify_batch_dev(len(seals), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_st.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        return d_st.cpu().numpy()

    r = real_proofs['risc0']
    seals, mut, mclass, flip = synth.make_batch('risc0', H(r['seal']), nb, 0x5A4B5651, pool=8, mutate_every=16)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (nb, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (nb, 1)); jds[flip, 0] ^= 1
    base = run_r0(seals, ids, jds)
    assert ((base == 0) == ~mut).all()
    n = 1 << 16
    src = np.random.default_rng(0x5A4B5652).permutation(n) % nb
    st = run_r0(seals[src], ids[src], jds[src])
    assert (st == base[src]).all() and int((st == 0).sum()) == int((~mut[src]).sum())
    assert (run_r0(seals[src], ids[src], jds[src]) == st).all()

    s = real_proofs['sp1']
    proofs, mut, mclass, flip = synth.make_batch('sp1', H(s['proof']), nb, 0x5A4B5653, pool=8, mutate_every=16)
    vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (nb, 1))
    pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (nb, 1)); pv[flip, -1] ^= 1

    def run_sp1(proofs, vk, pv):
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (vk, pv, proofs)]
        d_st = torch.full((len(proofs),), 255, dtype=torch.uint8, device=dev)
        sp1.verify_batch_dev(len(proofs), d[0].data_ptr(), d[1].data_ptr(), pv.shape[1], d[2].data_ptr(), d_st.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        return d_st.cpu().numpy()

    base = run_sp1(proofs, vk, pv)
    assert ((base == 0) == ~mut).all()
    n = 1 << 20
    src = np.random.default_rng(0x5A4B5654).permutation(n) % nb
    st = run_sp1(proofs[src], vk[src], pv[src])
    assert (st == base[src]).all() and int((st == 0).sum()) == int((~mut[src]).sum())


def test_verifier_set_matches_per_instance_oracles(zkv, real_proofs):
    """A RiscZeroVerifierSet (many instances resident on the device, selectors derived by the set-up kernel):
    every proof's status equals what the oracle's verifier for ITS instance returns -- the real instance accepts the valid
    proofs, the others answer SelectorMismatch or (with their own selector spliced in) VerificationFailed, an instance
    with bn254_control_id >= R fails everything, an unknown index is an un-initialised verifier."""
    import random
    import torch
    import oracle_lib as ol
    from stylus_zkvm_verifiers_amd import synth
    rng = random.Random(0x5A4B5671)
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    r = real_proofs['risc0']
    roots = [H(r['control_root'])] + [rng.randbytes(32) for _ in range(3)] + [rng.randbytes(32), H(r['control_root'])]
    ids = [H(r['bn254_control_id'])] + [rng.randrange(R).to_bytes(32, 'big') for _ in range(3)] + [(R + 5).to_bytes(32, 'big'), H(r['bn254_control_id'])]
    vs = zkv.RiscZeroVerifierSet(roots, ids)
    oracles = []
    for cr, cid in zip(roots, ids):
        o = ol.Risc0Oracle(); o.initialize(cr, cid); oracles.append(o)
    assert len(vs) == 6
    for k, o in enumerate(oracles):
        assert vs.get_selector(k) == o.get_selector()
    n = 600
    seals, mut, mclass, flip = synth.make_batch('risc0', H(r['seal']), n, 0x5A4B5672, pool=4, mutate_every=7)
    iid = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n, 1)); jds[flip, 0] ^= 1
    inst = np.array([(i * 5 + i // 6) % 6 for i in range(n)], dtype=np.uint32)
    for i in range(0, n, 11):                                   # splice the instance's own selector in: reaches the pairing
        seals[i, :4] = np.frombuffer(oracles[inst[i]].get_selector(), dtype=np.uint8)
    inst[17] = 6; inst[300] = 0xFFFFFFFF                        # unknown instances
    st, rv = vs.verify_batch(inst, [x.tobytes() for x in seals], [x.tobytes() for x in iid], [x.tobytes() for x in jds])
    want_st, want_rv = [], []
    for i in range(n):
        if inst[i] >= 6:
            want_st.append(2); want_rv.append(bytes(4)); continue
        s, rcv = oracles[inst[i]].verify(seals[i].tobytes(), iid[i].tobytes(), jds[i].tobytes())
        want_st.append(s); want_rv.append(rcv or bytes(4))
    assert list(st) == want_st
    assert [bytes(x) for x in rv] == want_rv
    assert {0, 1, 2, 5} <= set(want_st)
    # device-resident entry point
    dev = torch.device('cuda', 0)
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (seals, iid, jds)]
    d_inst = torch.from_numpy(inst.view(np.int32)).to(dev)
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    vs.verify_batch_dev(n, d_inst.data_ptr(), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_st.data_ptr(), 0,
                        torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert list(d_st.cpu().numpy()) == want_st
    # compute_vk_x per instance
    sig = [(rng.randrange(1 << 128).to_bytes(32, 'big'), rng.randrange(1 << 128).to_bytes(32, 'big')) for _ in range(24)]
    which = [k % 4 for k in range(24)]
    got = vs.vk_x_batch(which, sig)
    for k, (a, b) in zip(which, sig):
        lo, hi = oracles[k].get_control_root()
        assert got.pop(0) == ol.groth16_vk_x(0, [lo.rjust(32, b'\0'), hi.rjust(32, b'\0'), a, b, ids[k]])
    vs.close()


def test_differential_fuzz_against_the_oracle(zkv, r0, sp1, real_proofs):
    """Seeded random damage -- byte flips, word swaps, coordinate edits near 0 / Q / 2^256, truncations -- applied to valid
    seals and to canonical calldata; every outcome (status, received selector, return / revert data) must equal the oracle's."""
    import random
    import oracle_lib as ol
    rng = random.Random(0x5A4B5691)
    Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    r, s = real_proofs['risc0'], real_proofs['sp1']
    orc = ol.Risc0Oracle(); orc.initialize(H(r['control_root']), H(r['bn254_control_id']))

    def damage(seal):
        b = bytearray(seal)
        for _ in range(rng.choice((1, 1, 1, 2, 3))):
            k = rng.randrange(8)
            if k == 0:
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            elif k == 1:                                     # replace one word by an edge value
                w = 4 + 32 * rng.randrange(8)
                v = rng.choice((0, 1, 2, Q - 1, Q, Q + 1, (1 << 256) - 1, rng.randrange(1 << 256), rng.randrange(Q)))
                b[w:w + 32] = v.to_bytes(32, 'big')
            elif k == 2:                                     # swap two words
                i, j = 4 + 32 * rng.randrange(8), 4 + 32 * rng.randrange(8)
                b[i:i + 32], b[j:j + 32] = b[j:j + 32], b[i:i + 32]
            elif k == 3:
                del b[rng.randrange(len(b) + 1):]
            elif k == 4:
                b += rng.randbytes(rng.randrange(1, 40))
            elif k == 5:                                     # zero a whole point
                w = rng.choice((4, 68, 196))
                n = 128 if w == 68 else 64
                b[w:w + n] = bytes(n)
            elif k == 6:                                     # negate a G1 y coordinate (valid point, wrong proof)
                w = rng.choice((36, 228))
                y = int.from_bytes(b[w:w + 32], 'big')
                if 0 < y < Q:
                    b[w:w + 32] = (Q - y).to_bytes(32, 'big')
            # k == 7: leave as is
        return bytes(b)

    n = 1500
    seals = [damage(H(r['seal'])) for _ in range(n)]
    ids = [H(r['image_id'])] * n
    jds = [H(r['journal_digest']) if rng.random() < 0.9 else rng.randbytes(32) for _ in range(n)]
    ost, orv = orc.verify_batch(seals, ids, jds, threads=8)
    for lanes in (0, 128, 64, 16, 2):        # every kernel mapping under the same damaged inputs (0 = automatic: one proof per wavefront at this size)
        r0.set_lanes_per_proof(lanes)
        st, rv = r0.verify_batch(seals, ids, jds)
        assert (st == ost).all() and (rv.reshape(-1) == orv).all(), lanes
    r0.set_lanes_per_proof(0)
    assert {0, 1, 4, 5} <= set(int(x) for x in st)
    proofs = [damage(H(s['proof'])) for _ in range(n)]
    vks = [H(s['vkey']) if rng.random() < 0.9 else rng.randbytes(32) for _ in range(n)]
    pvs = [H(s['public_values']) if rng.random() < 0.8 else rng.randbytes(rng.randrange(0, 200)) for _ in range(n)]
    ost, orv = ol.sp1_verify_batch(vks, pvs, proofs, threads=8)
    for lanes in (0, 128, 64, 16, 2):
        sp1.set_lanes_per_proof(lanes)
        st, rv = sp1.verify_batch(vks, pvs, proofs)
        assert (st == ost).all() and (rv.reshape(-1) == orv).all(), lanes
    sp1.set_lanes_per_proof(0)

    # calldata: damage anywhere in the canonical encoding
    def damage_cd(cd):
        b = bytearray(cd)
        for _ in range(rng.choice((1, 1, 2))):
            k = rng.randrange(6)
            if k == 0:
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            elif k == 1:
                del b[rng.randrange(len(b) + 1):]
            elif k == 2:
                b += bytes(rng.randrange(1, 70))
            elif k == 3:                                     # edit one of the first six words (heads, lengths)
                w = 4 + 32 * rng.randrange(6)
                b[w + 28:w + 32] = rng.choice((0, 0x20, 0x40, 0x60, 0x80, 259, 260, 261, 96, 1 << 31)).to_bytes(4, 'big')
            elif k == 4:                                     # drop or duplicate a whole word
                w = 4 + 32 * rng.randrange(max(1, (len(b) - 4) // 32))
                if rng.random() < 0.5:
                    del b[w:w + 32]
                else:
                    b[w:w] = b[w:w + 32]
        return bytes(b)

    m_ = 400
    base = zkv.wire.encode_risc0_verify(H(r['seal']), H(r['image_id']), H(r['journal_digest']))
    base_i = zkv.wire.encode_risc0_verify_integrity(H(r['seal']), H(r['claim_digest']))
    cds = [damage_cd(base if rng.random() < 0.7 else base_i) for _ in range(m_)]
    rev, ret, wst = zkv.wire.eth_call_batch(r0, cds)
    for cd, a, b in zip(cds, rev, ret):
        assert (bool(a), b) == orc.eth_call(cd)[:2]
    base = zkv.wire.encode_sp1_verify_proof(H(s['vkey']), H(s['public_values']), H(s['proof']))
    cds = [damage_cd(base) for _ in range(m_)]
    rev, ret, wst = zkv.wire.eth_call_batch(sp1, cds)
    for cd, a, b in zip(cds, rev, ret):
        assert (bool(a), b) == ol.sp1_eth_call(cd)[:2]


def test_two_contexts_from_two_host_threads(zkv, real_proofs, verify_corpus):
    """One context per host thread is the documented way to overlap work: two threads, two RISC Zero contexts and one SP1
    context on the same device, all verifying the corpus concurrently (ctypes releases the GIL during the calls)."""
    import threading
    r = real_proofs['risc0']
    rc = [c for c in verify_corpus['cases'] if c['vm'] == 'risc0'] * 3
    sc = [c for c in verify_corpus['cases'] if c['vm'] == 'sp1'] * 3
    out, errs = {}, []

    def risc0_worker(tag):
        try:
            v = zkv.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
            for rep in range(3):
                st, _ = v.verify_batch([H(c['seal']) for c in rc], [H(c['image_id']) for c in rc], [H(c['journal_digest']) for c in rc])
                out[(tag, rep)] = list(st)
            v.close()
        except Exception as e:                      # surfaced below: an exception in a thread must fail the test
            errs.append(e)

    def sp1_worker(tag):
        try:
            v = zkv.Sp1Verifier()
            for rep in range(3):
                st, _ = v.verify_batch([H(c['vkey']) for c in sc], [H(c['public_values']) for c in sc], [H(c['proof']) for c in sc])
                out[(tag, rep)] = list(st)
            v.close()
        except Exception as e:
            errs.append(e)

    ts = [threading.Thread(target=risc0_worker, args=('a',)), threading.Thread(target=risc0_worker, args=('b',)),
          threading.Thread(target=sp1_worker, args=('s',))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for rep in range(3):
        assert out[('a', rep)] == out[('b', rep)] == [c['status'] for c in rc]
        assert out[('s', rep)] == [c['status'] for c in sc]


def test_small_chunks_give_the_same_answers(zkv, r0, sp1, real_proofs):
    """Every entry point loops over workspace chunks (ZKV_CHUNK proofs).  Contexts created with a 192-proof chunk must return
    exactly what the default contexts return on 700-proof batches: host and device-resident seal paths, calldata paths,
    verifier sets, vk_x and the generic / precompile entry points that share the loop."""
    import torch
    from stylus_zkvm_verifiers_amd import synth
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream().cuda_stream
    r, s = real_proofs['risc0'], real_proofs['sp1']
    n = 700
    seals, mut, mclass, flip = synth.make_batch('risc0', H(r['seal']), n, 0x5A4B56A1, pool=4, mutate_every=5)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n, 1)); jds[flip, 0] ^= 1
    proofs, smut, _, sflip = synth.make_batch('sp1', H(s['proof']), n, 0x5A4B56A2, pool=4, mutate_every=5)
    vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (n, 1))
    pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (n, 1)); pv[sflip, -1] ^= 1
    rows = lambda a: [x.tobytes() for x in a]
    os.environ['ZKV_CHUNK'] = '192'
    try:
        r0s = zkv.RiscZeroVerifier(); r0s.initialize(H(r['control_root']), H(r['bn254_control_id']))
        sp1s = zkv.Sp1Verifier()
        sets = zkv.RiscZeroVerifierSet([H(r['control_root']), bytes(32)], [H(r['bn254_control_id']), bytes(32)])
        # host seal paths
        want, want_rv = r0.verify_batch(rows(seals), rows(ids), rows(jds))
        got, got_rv = r0s.verify_batch(rows(seals), rows(ids), rows(jds))
        assert (got == want).all() and (got_rv == want_rv).all() and ((want == 0) == ~mut).all()
        swant, _ = sp1.verify_batch(rows(vk), rows(pv), rows(proofs))
        sgot, _ = sp1s.verify_batch(rows(vk), rows(pv), rows(proofs))
        assert (sgot == swant).all() and ((swant == 0) == ~smut).all()
        # device-resident seal path
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (seals, ids, jds)]
        d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        r0s.verify_batch_dev(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_st.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        assert (d_st.cpu().numpy() == want).all()
        # calldata, host and device-resident
        cds = [zkv.wire.encode_risc0_verify(a, b, c) for a, b, c in zip(rows(seals), rows(ids), rows(jds))]
        rev, ret, wst = zkv.wire.eth_call_batch(r0s, cds)
        assert (wst == want).all()
        cd = synth.calldata_sp1_verify_proof(vk, pv, proofs)
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(cd.shape[1])
        d_cd, d_off = torch.from_numpy(cd).to(dev), torch.from_numpy(off.view(np.int64)).to(dev)
        d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        zkv.wire.eth_call_batch_dev(sp1s, n, d_cd.data_ptr(), d_off.data_ptr(), cd.size, d_st.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        assert (d_st.cpu().numpy() == swant).all()
        # verifier set: instance 0 is the real verifier, instance 1 another one
        inst = np.array([i % 2 for i in range(n)], dtype=np.uint32)
        st_set, _ = sets.verify_batch(inst, rows(seals), rows(ids), rows(jds))
        assert (st_set[inst == 0] == want[inst == 0]).all() and set(st_set[inst == 1]) <= {4, 5}
        # vk_x
        sig = [(bytes(16) + bytes([i % 251] * 16), bytes(16) + bytes([(7 * i) % 253] * 16)) for i in range(n)]
        assert r0s.vk_x_batch(sig) == r0.vk_x_batch(sig)
        # precompile seam (same chunk loop)
        pc = zkv.Bn254Precompiles()
        G = (1).to_bytes(32, 'big') + (2).to_bytes(32, 'big')
        muls = [G + (i * 0x9E3779B97F4A7C15 % (1 << 200)).to_bytes(32, 'big') for i in range(n)]
        small = pc.ecmul(muls)
        pc.close()
        for v in (r0s, sp1s, sets):
            v.close()
    finally:
        os.environ.pop('ZKV_CHUNK', None)
    pc2 = zkv.Bn254Precompiles()
    assert pc2.ecmul(muls) == small
    pc2.close()


def test_host_batches_in_several_passes_and_segments(zkv, r0, sp1, real_proofs, verify_corpus):
    """The host-buffer driver stages a batch pass by pass (ZKV_HOST_PASS proofs) and, inside a pass, segment by segment on the copy
    stream (a short first segment, then workspace-sized ones).  With a 150-proof pass, a 40-proof first segment and a 64-proof
    workspace a 700-proof RAGGED batch (corpus seals of every length spliced in, so passes start at non-zero blob offsets) must
    return exactly what the default contexts return; same for SP1 with ragged public values."""
    from stylus_zkvm_verifiers_amd import synth
    r, s = real_proofs['risc0'], real_proofs['sp1']
    n = 700
    seals, mut, _, flip = synth.make_batch('risc0', H(r['seal']), n, 0x5A4B56B1, pool=4, mutate_every=7)
    ids = [H(r['image_id'])] * n
    jds = [H(r['journal_digest'])[:-1] + bytes([H(r['journal_digest'])[-1] ^ (1 if f else 0)]) for f in flip]
    ragged = [H(c['seal']) for c in verify_corpus['cases'] if c['vm'] == 'risc0']
    sl = [x.tobytes() for x in seals]
    for k, x in enumerate(ragged):
        sl[(37 * k + 5) % n] = x                                  # lengths 0 .. 292 sprinkled through the batch
    proofs, smut, _, sflip = synth.make_batch('sp1', H(s['proof']), n, 0x5A4B56B2, pool=4, mutate_every=7)
    pvs = [H(s['public_values'])[:(96 if i % 3 else 40 + i % 50)] for i in range(n)]             # ragged public values (most proofs then fail)
    want, want_rv = r0.verify_batch(sl, ids, jds)
    swant, swant_rv = sp1.verify_batch([H(s['vkey'])] * n, pvs, [x.tobytes() for x in proofs])
    os.environ.update(ZKV_HOST_PASS='150', ZKV_HOST_FIRST_SEGMENT='40', ZKV_CHUNK='64')
    try:
        r0s = zkv.RiscZeroVerifier(); r0s.initialize(H(r['control_root']), H(r['bn254_control_id']))
        sp1s = zkv.Sp1Verifier()
        got, got_rv = r0s.verify_batch(sl, ids, jds)
        assert (got == want).all() and (got_rv == want_rv).all()
        sgot, sgot_rv = sp1s.verify_batch([H(s['vkey'])] * n, pvs, [x.tobytes() for x in proofs])
        assert (sgot == swant).all() and (sgot_rv == swant_rv).all()
        r0s.close(); sp1s.close()
    finally:
        for k in ('ZKV_HOST_PASS', 'ZKV_HOST_FIRST_SEGMENT', 'ZKV_CHUNK'):
            os.environ.pop(k, None)
    assert (want == 0).sum() > 500 and (swant == 0).sum() > 300 and len(set(want)) >= 3


def test_pinned_host_buffers_give_the_same_statuses(zkv, r0, real_proofs):
    """zkv_host_register: a 3,000-proof RISC Zero batch handed over from buffers the caller pinned beforehand (direct DMA staging) returns
    the bytes the same call returns from pageable memory."""
    from stylus_zkvm_verifiers_amd import _lib, synth
    L = _lib.lib()
    r = real_proofs['risc0']
    n = 3000
    seals, mut, _, flip = synth.make_batch('risc0', H(r['seal']), n, 0x5A4B56C1, pool=4, mutate_every=9)
    seals = np.ascontiguousarray(seals)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(seals.shape[1])
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n, 1)).copy()
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n, 1)).copy()
    jds[flip, 0] ^= 1
    st_a, st_b = np.full(n, 255, dtype=np.uint8), np.full(n, 255, dtype=np.uint8)
    rv_a, rv_b = np.zeros(4 * n, dtype=np.uint8), np.zeros(4 * n, dtype=np.uint8)
    call = lambda st, rv: _lib.check(L.zkv_risc0_verify_batch(r0._h, n, seals.ctypes.data, off.ctypes.data, ids.ctypes.data, jds.ctypes.data,
                                                              st.ctypes.data, rv.ctypes.data), 'batch')
    call(st_a, rv_a)
    bufs = [seals, off, ids, jds, st_b, rv_b]
    for x in bufs:
        zkv.host_register(x)
    try:
        call(st_b, rv_b)
    finally:
        for x in bufs:
            zkv.host_unregister(x)
    assert (st_a == st_b).all() and (rv_a == rv_b).all()
    assert ((st_a == 0) == ~mut).all() and 0 < mut.sum() < n


def test_device_calldata_with_corrupt_offsets_is_never_read(zkv, r0, real_proofs):
    """zkv_eth_call_batch_dev takes its offsets from device memory: requests whose offsets run backwards or leave the blob get
    BAD_CALLDATA without being dereferenced; their neighbours are unaffected."""
    import torch
    dev = torch.device('cuda', 0)
    r = real_proofs['risc0']
    cd = np.frombuffer(zkv.wire.encode_risc0_verify(H(r['seal']), H(r['image_id']), H(r['journal_digest'])), dtype=np.uint8)
    n, L = 6, len(cd)
    blob = np.tile(cd, n)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    off[2] = np.uint64(1) << np.uint64(62)                      # request 1 ends far outside, request 2 starts there
    off[5] = off[4] - np.uint64(8)                              # request 4 runs backwards; request 5 then is 8 bytes too long
    d_cd, d_off = torch.from_numpy(blob).to(dev), torch.from_numpy(off.view(np.int64)).to(dev)
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    zkv.wire.eth_call_batch_dev(r0, n, d_cd.data_ptr(), d_off.data_ptr(), blob.size, d_st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert list(d_st.cpu().numpy()) == [0, 6, 6, 0, 6, 6]


def test_sixteen_lanes_per_proof_kernels_match_the_oracle(zkv, real_proofs, verify_corpus):
    """The small-batch kernels (one proof per 16 lanes, coefficient-parallel Fp12 arithmetic, csrc/zkv_tower_wide.h) against the
    golden corpus, the oracle on a seeded batch with every mutation class (partial last wavefront: 4 proofs per wave), and the
    lane-pair kernels on the same inputs; the automatic choice (small chunks -> 16 lanes) returns the same bytes."""
    import oracle_lib as ol
    from stylus_zkvm_verifiers_amd import synth
    r, s = real_proofs['risc0'], real_proofs['sp1']
    rc = [c for c in verify_corpus['cases'] if c['vm'] == 'risc0']
    sc = [c for c in verify_corpus['cases'] if c['vm'] == 'sp1']
    n = 301
    seals, mut, mclass, flip = synth.make_batch('risc0', H(r['seal']), n, 0x5A4B56C1, pool=4, mutate_every=4)
    ids = [H(r['image_id'])] * n
    jds = [H(r['journal_digest']) if not f else bytes([H(r['journal_digest'])[0] ^ 1]) + H(r['journal_digest'])[1:] for f in flip]
    orc = ol.Risc0Oracle(); orc.initialize(H(r['control_root']), H(r['bn254_control_id']))
    want, _ = orc.verify_batch([x.tobytes() for x in seals], ids, jds, threads=8)
    proofs, smut, _, sflip = synth.make_batch('sp1', H(s['proof']), n, 0x5A4B56C2, pool=4, mutate_every=4)
    pvs = [H(s['public_values']) if not f else H(s['public_values'])[:-1] + bytes([H(s['public_values'])[-1] ^ 1]) for f in sflip]
    swant, _ = ol.sp1_verify_batch([H(s['vkey'])] * n, pvs, [x.tobytes() for x in proofs], threads=8)
    got = {}
    for lanes in (16, 64, 128, 2, 0):        # 0 = automatic: these batches are below ZKV_DUAL_BELOW, so two wavefronts per proof in the Miller loop
        v = zkv.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id'])); v.set_lanes_per_proof(lanes)
        st, rv = v.verify_batch([H(c['seal']) for c in rc], [H(c['image_id']) for c in rc], [H(c['journal_digest']) for c in rc])
        assert [int(x) for x in st] == [c['status'] for c in rc], lanes
        st, _ = v.verify_batch([x.tobytes() for x in seals], ids, jds)
        assert (st == want).all(), lanes
        sp = zkv.Sp1Verifier(); sp.set_lanes_per_proof(lanes)
        st2, _ = sp.verify_batch([H(c['vkey']) for c in sc], [H(c['public_values']) for c in sc], [H(c['proof']) for c in sc])
        assert [int(x) for x in st2] == [c['status'] for c in sc], lanes
        st3, _ = sp.verify_batch([H(s['vkey'])] * n, pvs, [x.tobytes() for x in proofs])
        assert (st3 == swant).all(), lanes
        got[lanes] = (st.tobytes(), st3.tobytes(), v.last_stage_ms())
        v.close(); sp.close()
    assert got[16][:2] == got[2][:2] == got[0][:2] == got[64][:2] == got[128][:2]
    assert ((want == 0) == ~mut).all() and ((swant == 0) == ~smut).all()


def test_last_small_chunk_switches_kernels_inside_one_batch(zkv, real_proofs):
    """A batch of 16,384 + 300 proofs on a context with 16,384-proof chunks: the first chunk runs on the lane-pair kernels, the
    300-proof remainder on the 16-lane kernels; statuses must equal the base batch they were tiled from."""
    import torch
    from stylus_zkvm_verifiers_amd import synth
    dev = torch.device('cuda', 0)
    r = real_proofs['risc0']
    nb = 1024
    seals, mut, _, flip = synth.make_batch('risc0', H(r['seal']), nb, 0x5A4B56D1, pool=4, mutate_every=6)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (nb, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (nb, 1)); jds[flip, 0] ^= 1
    os.environ['ZKV_CHUNK'] = '16384'
    try:
        v = zkv.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
        n = 16384 + 300
        src = np.random.default_rng(0x5A4B56D2).integers(0, nb, n)
        d = [torch.from_numpy(np.ascontiguousarray(x[src])).to(dev) for x in (seals, ids, jds)]
        d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        v.verify_batch_dev(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        st = d_st.cpu().numpy()
        v.close()
    finally:
        os.environ.pop('ZKV_CHUNK', None)
    assert ((st == 0) == ~mut[src]).all()
    assert len(set(st[16384:])) >= 2                             # the remainder holds accepted and rejected proofs


def test_chunks_that_only_just_start_a_new_layer_split_off_their_tail(zkv, r0, real_proofs):
    """32,768 k + r proofs with a small r: the last r proofs run through the small-batch mapping their number selects -- BESIDE the lane-pair
    kernels of the others on the context's second stream when k is odd, after them when k is even (zkv_capi.hip tail_of_chunk).  Statuses
    must equal the construction's and the unsplit run's (ZKV_TAIL_SPLIT_BELOW=0) at the split points: k = 1 with r = 1 and 700 (two
    wavefronts per proof), 2,000 (one wavefront per proof), 8,192 (16 lanes per proof) and 8,193 (no split); k = 2 with r = 3,072, 12,288 and
    12,289 (no split); k = 3 with r = 4,096 and 4,097 (no split).  Mutations of every class, so the tail holds early rejects, subgroup failures
    and pairing failures.  Twice, so that the second stream's kernels of one call meet the next call's."""
    import torch
    from stylus_zkvm_verifiers_amd import synth
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream().cuda_stream
    r = real_proofs['risc0']
    nmax = 98304 + 4097
    seals, mut, _, flip = synth.make_batch_parallel('risc0', H(r['seal']), nmax, 0x5A4B56F1, mutate_every=5)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (nmax, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (nmax, 1)); jds[flip, 0] ^= 1
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (seals, ids, jds)]
    def run(n, times=1):
        sts = [torch.full((n,), 255, dtype=torch.uint8, device=dev) for _ in range(times)]
        for st in sts:                                            # back to back, no synchronisation in between
            r0.verify_batch_dev(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), st.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        return [st.cpu().numpy() for st in sts]
    for n in (32768 + 1, 32768 + 700, 32768 + 2000, 32768 + 8192, 32768 + 8193, 65536 + 3072, 65536 + 12288, 65536 + 12289, 98304 + 4096, 98304 + 4097):
        got, again = run(n, 2)
        os.environ['ZKV_TAIL_SPLIT_BELOW'] = '0'
        try:
            plain = run(n)[0]
        finally:
            del os.environ['ZKV_TAIL_SPLIT_BELOW']
        assert (got == plain).all() and (again == plain).all(), (n, np.flatnonzero(got != plain)[:8], np.flatnonzero(again != plain)[:8])
        assert ((got == 0) == ~mut[:n]).all() and len(set(got[-700:])) >= 2, n


def test_small_order_g2_points_through_the_pairing_kernels(zkv, r0, sp1, precompile_kats, g2_membership_points, real_proofs):
    """The 14 + 36 G2 known-answer points (in-subgroup, random twist points, points of exact order 10069 / 5864401 / their product,
    G2 points with such a component, cofactor-only points) go
    through the GPU's subgroup checks where the reference would meet them (the ecPairing call, groth16.rs:121-125): as the G2 input of
    `zkv_bn254_pairing_batch` (`ok` = the precompile succeeds) and as `B` of a RISC Zero seal and of an SP1 proof (lane-pair kernels
    k_g2chk2 / k_miller2, and the 16-lane kernels for this small batch).  Expected: the fixture's flags, the C oracle, and the
    oracle's verifiers."""
    import oracle_lib as ol
    pts = precompile_kats['g2_subgroup'] + [dict(point=c['point'], on_twist=True, in_subgroup=c['in_subgroup']) for c in g2_membership_points]
    G1 = (1).to_bytes(32, 'big') + (2).to_bytes(32, 'big')
    calls = [G1 + H(''.join(c['point'])) for c in pts]
    pc = zkv.Bn254Precompiles()
    for dual in ('768', '0'):             # the two-wavefront kernels of small batches (verdict from the producer's final point), then the lane pairs
        os.environ['ZKV_DUAL_BELOW'] = dual
        got = pc.pairing(calls, 1)
        for c, call, g in zip(pts, calls, got):
            want = ol.ecpairing(call)
            assert (g is not None) == (c['on_twist'] and c['in_subgroup']) == (want is not None), (dual, c['point'][0][:16])
            if g is not None:
                assert g == bool(want[-1])
        # the same points paired with infinity on the G1 side: validation happens regardless of the skip (EIP-197)
        got = pc.pairing([bytes(64) + H(''.join(c['point'])) for c in pts], 1)
        assert [g is not None for g in got] == [c['on_twist'] and c['in_subgroup'] for c in pts], dual
    del os.environ['ZKV_DUAL_BELOW']
    pc.close()
    r, s = real_proofs['risc0'], real_proofs['sp1']
    orc = ol.Risc0Oracle(); orc.initialize(H(r['control_root']), H(r['bn254_control_id']))
    seal, proof = H(r['seal']), H(s['proof'])
    seals = [seal[:68] + H(''.join(c['point'])) + seal[196:] for c in pts]
    proofs = [proof[:68] + H(''.join(c['point'])) + proof[196:] for c in pts]
    for lanes in (2, 16, 64, 128):
        r0.set_lanes_per_proof(lanes); sp1.set_lanes_per_proof(lanes)
        st, _ = r0.verify_batch(seals, [H(r['image_id'])] * len(pts), [H(r['journal_digest'])] * len(pts))
        for x, got_st in zip(seals, st):
            assert int(got_st) == orc.verify(x, H(r['image_id']), H(r['journal_digest']))[0] == 1      # a foreign B never verifies
        st, _ = sp1.verify_batch([H(s['vkey'])] * len(pts), [H(s['public_values'])] * len(pts), proofs)
        for x, got_st in zip(proofs, st):
            assert int(got_st) == ol.sp1_verify_proof(H(s['vkey']), H(s['public_values']), x)[0] == 1
    r0.set_lanes_per_proof(0); sp1.set_lanes_per_proof(0)


def test_vk_x_on_generic_trapdoor_keys(zkv):
    """zkv_ctx_vk_x_batch on ZKV_VM_GROTH16 contexts: all n_ic - 1 signals per proof (1..5), random and edge values, against the
    oracle's ecMul / ecAdd chain (groth16.rs:51-58) on the same trapdoor key."""
    import random
    import oracle_lib as ol
    import spec_model as m
    rng = random.Random(4711)
    for n_ic in (1, 2, 3, 4, 6):
        vk, td = m.trapdoor_vk(rng, n_ic)
        vkb = m.vk_to_words(vk)
        v = zkv.Groth16Verifier(vkb, n_ic, zkv.errors.VM_SP1)
        sigs = [[rng.choice([0, 1, m.R - 1, rng.randrange(m.R)]) for _ in range(n_ic - 1)] for _ in range(70)]
        got = v.vk_x_batch([[m.be32(x) for x in sg] for sg in sigs])
        for sg, g in zip(sigs, got):
            assert g == ol.groth16_vk_x_vk(vkb, n_ic, [m.be32(x) for x in sg]), (n_ic, sg)
        v.close()
