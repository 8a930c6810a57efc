"""The opt-in aggregate check (csrc/zkv_agg.h, k_agg.hip): the arithmetic compiled for the host against the spec model's integers, and
on the GPU the statuses of aggregate-mode batches against the deterministic ones (= the oracle's).  The reference has no such mode
(it verifies one proof per call); what is pinned here is that switching it on never changes a status."""
import ctypes as C
import hashlib
import os
import random
import subprocess

import pytest

import spec_model as m

HERE = os.path.dirname(os.path.abspath(__file__))
LAMBDA = int.from_bytes(b''.join(w.to_bytes(4, 'little') for w in
                                 (0x36636f23, 0xb8ca0b2d, 0xec2bc5e9, 0xcc37a73f, 0x3fd84104, 0x048b6e19, 0xe131a029, 0x30644e72)), 'little')


@pytest.fixture(scope='module')
def hsa():
    src = os.path.join(HERE, 'host_sim', 'host_sim_agg.cpp')
    lib = os.path.join(HERE, 'host_sim', 'libhost_sim_agg.so')
    csrc = os.path.join(HERE, '..', 'stylus_zkvm_verifiers_amd', 'csrc')
    deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith('.h')]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-Wno-unknown-pragmas', '-o', lib, src])
    L = C.CDLL(lib)
    L.hsa_mul.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_char_p]
    L.hsa_e.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_char_p]
    L.hsa_coeff.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.hsa_norm3.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.hsa_norm3.restype = C.c_uint32
    return L


def _xy(pt): return b'\0' * 64 if pt is None else m.be32(pt[0]) + m.be32(pt[1])
def _pt(b): return (int.from_bytes(b[:32], 'big'), int.from_bytes(b[32:64], 'big'))


def test_lambda_is_a_cube_root_of_unity_and_the_short_pairs_are_injective():
    assert (LAMBDA * LAMBDA + LAMBDA + 1) % m.R == 0 and LAMBDA != 1
    # (r1, r2) -> r1 + r2 lambda is injective on [0, 2^64)^2 iff the lattice {(a, b): a + b lambda = 0 mod R} has no non-zero vector with
    # |a|, |b| < 2^64: Gauss-reduce its basis (R, 0), (-lambda, 1) and look at the shortest vector
    u, v = (m.R, 0), (-LAMBDA % m.R, 1)
    n2 = lambda w: w[0] * w[0] + w[1] * w[1]
    while True:
        if n2(u) < n2(v): u, v = v, u
        q = (u[0] * v[0] + u[1] * v[1] + n2(v) // 2) // n2(v)
        if q == 0: break
        u = (u[0] - q * v[0], u[1] - q * v[1])
    # a vector with both entries below 2^64 is shorter than 2^64.5; every non-zero lattice vector is at least as long as v (~2^127)
    assert n2(v) > 2 * (1 << 128)


def test_coefficients_are_sha256_of_seed_call_index(hsa):
    seed = bytes(range(32))
    for call, idx in ((0, 0), (1, 5), (0xFFFFFFFF, 0x03FFFFFF), (7, 1 << 20)):
        r = (C.c_uint64 * 2)()
        hsa.hsa_coeff(seed, call, idx, r)
        h = hashlib.sha256(seed + call.to_bytes(4, 'big') + idx.to_bytes(4, 'big')).digest()
        assert (r[0], r[1]) == (int.from_bytes(h[:8], 'big'), int.from_bytes(h[8:16], 'big'))


def test_glv_scalar_multiplication_matches_the_spec_model(hsa):
    rng = random.Random(0xA66)
    g = (1, 2)
    pts = [g, m.g1_mul(g, 5), m.g1_mul(g, rng.randrange(m.R)), m.g1_mul(g, m.R - 1)]
    pairs = [(0, 0), (1, 0), (0, 1), (1, 1), (2, 3), ((1 << 64) - 1, (1 << 64) - 1), (1 << 63, 1), (0x8000000000000001, 0xFFFFFFFF00000000)]
    pairs += [(rng.randrange(1 << 64), rng.randrange(1 << 64)) for _ in range(12)]
    for p in pts:
        for r1, r2 in pairs:
            o = C.create_string_buffer(64)
            inf = hsa.hsa_mul(_xy(p), r1, r2, o)
            want = m.g1_mul(p, (r1 + r2 * LAMBDA) % m.R)
            assert (inf == 1) == (want is None)
            if want is not None: assert _pt(o.raw) == want


def test_lane_shares_sum_to_e(hsa):
    rng = random.Random(0xE5)
    alpha = m.g1_mul((1, 2), rng.randrange(1, m.R))
    cases = [(0, 0, 1), (1, 0, 1), (5, 7, 3), ((1 << 70) - 1, (1 << 70) - 1, 65), (64 * ((1 << 64) - 1), 0, 65), (0, 64 * ((1 << 64) - 1), 1)]
    cases += [(rng.randrange(1 << 70), rng.randrange(1 << 70), rng.randrange(1, 66)) for _ in range(10)]
    for k, (s1, s2, c) in enumerate(cases):
        for sub in ((16, 32, 64) if k < 8 else (64,)):
            o = C.create_string_buffer(64)
            inf = hsa.hsa_e(_xy(alpha), sub, s1 & ((1 << 64) - 1), s1 >> 64, s2 & ((1 << 64) - 1), s2 >> 64, c, o)
            want = m.g1_mul(alpha, (s1 - c + s2 * LAMBDA) % m.R)
            assert (inf == 1) == (want is None)
            if want is not None: assert _pt(o.raw) == want


def test_three_point_normalisation(hsa):
    rng = random.Random(0x303)
    g = (1, 2)
    for mask in range(8):
        pts = [None if mask >> k & 1 else m.g1_mul(g, rng.randrange(1, m.R)) for k in range(3)]
        z = [rng.randrange(1, m.P) for _ in range(3)]
        o = C.create_string_buffer(192)
        fl = hsa.hsa_norm3(b''.join(_xy(p) for p in pts), b''.join(m.be32(v) for v in z), o)
        assert fl == 1 | (2 if pts[0] is None else 0) | (16 if pts[1] is None else 0) | (8 if pts[2] is None else 0)
        for k, p in enumerate(pts):
            xs, ys = int.from_bytes(o.raw[64 * k:64 * k + 32], 'big'), int.from_bytes(o.raw[64 * k + 32:64 * k + 64], 'big')
            if p is None: assert (xs, ys) == (0, 0)
            else: assert (xs, ys) == (p[0] * pow(p[1], -1, m.P) % m.P, pow(p[1], -1, m.P))


def test_coefficient_as_field_element(hsa):
    rng = random.Random(0xF7)
    for r1, r2 in [(0, 0), (1, 0), (0, 1), ((1 << 64) - 1, (1 << 64) - 1)] + [(rng.randrange(1 << 64), rng.randrange(1 << 64)) for _ in range(20)]:
        o = C.create_string_buffer(32)
        hsa.hsa_coeff_fr(C.c_uint64(r1), C.c_uint64(r2), o)
        assert int.from_bytes(o.raw, 'big') == (r1 + r2 * LAMBDA) % m.R


def test_window_shares_sum_to_the_weighted_vk_x(hsa, real_proofs):
    """U = R base + T_0 IC_a + T_1 IC_b from the keys' window tables (all 32 windows of each per-proof signal, although a RISC Zero
    signal itself has 16), for 16, 32 and 64 lanes per sub-batch; R = 1 and T = the signals gives compute_vk_x itself."""
    rng = random.Random(0x77)
    r = real_proofs['risc0']
    cr, cid = bytes.fromhex(r['control_root']), bytes.fromhex(r['bn254_control_id'])
    v = m.Risc0Verifier(); v.initialize(cr, cid)
    fixed = v.signals(bytes(32))                       # control root halves, zero claim halves, control id
    keys = [(0, m.RISC0_VK, fixed, (2, 3)), (1, m.SP1_VK, [0, 0], (0, 1))]
    hsa.hsa_u.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p]
    for vm, vk, sig0, var in keys:
        base = m.compute_vk_x(vk, sig0)
        ics = [vk['ic'][1 + var[0]], vk['ic'][1 + var[1]]]
        cases = [(1, 5, 7), (0, 0, 0), (m.R - 1, m.R - 1, m.R - 1), (1 << 253, 255, 256)] + [tuple(rng.randrange(m.R) for _ in range(3)) for _ in range(3)]
        for k, (R_, t0, t1) in enumerate(cases):
            for sub in ((16, 32, 64) if k < 2 else (64,)):
                o = C.create_string_buffer(64)
                inf = hsa.hsa_u(vm, cr, cid, sub, m.be32(R_), m.be32(t0) + m.be32(t1), o)
                want = m.g1_add(m.g1_add(m.g1_mul(base, R_), m.g1_mul(ics[0], t0)), m.g1_mul(ics[1], t1))
                assert (inf == 1) == (want is None), (vm, k)
                if want is not None: assert _pt(o.raw) == want, (vm, k, sub)


def test_two_proofs_sharing_the_miller_accumulator(real_proofs, verify_corpus):
    """miller_loop_pg (two or four proofs of a lane pair, one accumulator, running points in memory rows) == the product of the proofs' own
    Miller values, for both / one proof taking part, with A at infinity, and a B outside the subgroup reported in its bit."""
    csrc = os.path.join(HERE, '..', 'stylus_zkvm_verifiers_amd', 'csrc')
    def build(name, extra):
        src = os.path.join(HERE, 'host_sim', name + '.cpp'); lib = os.path.join(HERE, 'host_sim', 'lib' + name + '.so')
        deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith('.h')]
        if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(d) for d in deps):
            subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-Wno-unknown-pragmas'] + extra + ['-o', lib, src])
        return C.CDLL(lib)
    hs, hp = build('host_sim', []), build('host_sim_paired', ['-pthread'])
    hs.hs_prepare.restype = C.c_void_p
    hp.hs2_miller2.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    r0 = real_proofs['risc0']
    cr, cid = H(r0['control_root']), H(r0['bn254_control_id'])
    v = m.Risc0Verifier(); v.initialize(cr, cid)
    rows = []
    for c in verify_corpus['cases']:
        if c['vm'] != 'risc0' or c['status'] not in (0, 1): continue
        sig = v.signals(m.receipt_claim_ok_digest(H(c['image_id']), H(c['journal_digest'])))
        fl = C.c_uint32(0); norm = (C.c_uint32 * 48)(); b = (C.c_uint32 * 32)()
        t = hs.hs_prepare(0, cr, cid, H(c['seal'])[4:], m.be32(sig[2]), m.be32(sig[3]), C.byref(fl), norm, b)
        if t and not (fl.value & (2 | 4)): rows.append((c['name'], t, list(norm), list(b)))
        if len(rows) >= 4: break
    assert len(rows) >= 3
    t = rows[0][1]
    def run(i, k, mask, abmask):
        norm96 = (C.c_uint32 * 96)(*(rows[i][2] + rows[k][2]))
        b96 = (C.c_uint32 * 96)(*(rows[i][3] + [0] * 16 + rows[k][3] + [0] * 16))
        fine = C.c_uint32(0)
        return hp.hs2_miller2(t, 2, mask, abmask, norm96, b96, C.byref(fine)), fine.value
    def run4(idx, mask, abmask):
        norm = (C.c_uint32 * 192)(*sum((rows[i][2] for i in idx), []))
        b = (C.c_uint32 * 192)(*sum((rows[i][3] + [0] * 16 for i in idx), []))
        fine = C.c_uint32(0)
        return hp.hs2_miller2(t, 4, mask, abmask, norm, b, C.byref(fine)), fine.value
    assert run4((0, 1, 2, 0), 15, 15) == (1, 15)
    assert run4((2, 1, 0, 1), 0b1011, 0b0011) == (1, 15)
    assert run(0, 1, 3, 3) == (1, 3)
    assert run(1, 2, 3, 3) == (1, 3)
    assert run(0, 0, 3, 3) == (1, 3)                   # the same proof twice
    assert run(0, 1, 1, 1) == (1, 3) and run(0, 1, 2, 2) == (1, 3)          # one proof of the pair only
    assert run(0, 1, 3, 1) == (1, 3) and run(0, 1, 3, 0) == (1, 3)          # A at infinity: the point is stepped, no line products


def test_ecmul_glv_walk_matches_the_spec_model_for_any_256_bit_scalar(hsa):
    """g1_mul_glv (k_ecmul, the ecMul seam): scalars 0, 1, r - 1, r, r + 1, 2^256 - 1, multiples of r, values around 2^128 and lambda,
    and random ones -- the EIP-196 answer is (k mod r) P."""
    rng = random.Random(0xEC)
    g = (1, 2)
    pts = [g, m.g1_mul(g, 7), m.g1_mul(g, rng.randrange(m.R))]
    ks = [0, 1, 2, m.R - 1, m.R, m.R + 1, 2 * m.R, 5 * m.R + 3, (1 << 256) - 1, (1 << 128) - 1, 1 << 128, (1 << 128) + 1, LAMBDA, LAMBDA + 1, m.R - LAMBDA,
          (1 << 255), (1 << 254) + 12345] + [rng.randrange(1 << 256) for _ in range(40)] + [rng.randrange(1 << 64) for _ in range(5)]
    for p in pts:
        for k in ks:
            o = C.create_string_buffer(64)
            inf = hsa.hsa_ecmul_glv(_xy(p), m.be32(k), o)
            want = m.g1_mul(p, k % m.R)
            assert (inf == 1) == (want is None), hex(k)
            if want is not None: assert _pt(o.raw) == want, hex(k)


# ---------------------------------------------------------------------------------------------------------------- GPU
H = bytes.fromhex


def _risc0_inputs(real_proofs, n, seed, mutate_every, classes=None):
    import numpy as np
    from stylus_zkvm_verifiers_amd import synth
    r = real_proofs['risc0']
    kw = {} if classes is None else {'classes': classes}
    seals, mut, mclass, flip = synth.make_batch('risc0', H(r['seal']), n, seed, pool=8, mutate_every=mutate_every, **kw)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n, 1)); jds[flip, 0] ^= 1
    return seals, ids, jds, mut, mclass


def _sub_batches(n, sub, g=4):
    """Sub-batches counted for a chunk of n proofs.  g proofs share a Miller accumulator (default 4; members l, l + 64/g, ... of a 64-proof
    block): a sub-batch is sub / g consecutive proofs of the block's first 64 / g and their partners; g = 1: `sub` consecutive proofs."""
    if sub > 64: return (n + sub - 1) // sub                     # 2 or 4 whole 64-proof blocks
    full, rem = divmod(n, 64)
    return full * (64 // sub) + (min(rem, 64 // g) + sub // g - 1) // (sub // g)


def _run_risc0_dev(v, seals, ids, jds):
    import numpy as np
    import torch
    dev = torch.device('cuda', 0)
    n = len(seals)
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (seals, ids, jds)]
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    v.verify_batch_dev(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_st.cpu().numpy()


@pytest.mark.gpu
def test_aggregate_check_gives_the_deterministic_statuses(real_proofs, monkeypatch):
    """RISC Zero, 4,160 proofs (65 sub-batches, the last one partial...), every mutation class, one proof in 7 mutated: statuses with
    the aggregate check on == statuses with it off == the oracle's on a sample; the counters show sub-batches that passed and
    sub-batches that were verified again."""
    import numpy as np
    import oracle_lib as ol
    import stylus_zkvm_verifiers_amd as zkv
    monkeypatch.setenv('ZKV_AGG_MIN', '64')
    r = real_proofs['risc0']
    n = 4160 + 17
    seals, ids, jds, mut, mclass = _risc0_inputs(real_proofs, n, 0x5A4B56A1, 7)
    v = zkv.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
    plain = _run_risc0_dev(v, seals, ids, jds)
    total = 0
    for g in (4, 1, 2, 8):
        monkeypatch.setenv('ZKV_AGG_GROUP', str(g))
        for sub in (16, 32, 64, 128, 256):
            v.set_aggregate_check(True, seed=bytes(range(32)), sub_batch=sub)
            agg = _run_risc0_dev(v, seals, ids, jds)
            checked, failed = v.aggregate_counters()
            assert (agg == plain).all(), (g, sub)
            assert ((agg == 0) == ~mut).all()
            total += _sub_batches(n, sub, g)
            assert checked == total and 0 < failed <= checked
    monkeypatch.delenv('ZKV_AGG_GROUP')
    # a second run draws other coefficients (the per-chunk counter): same statuses
    assert (_run_risc0_dev(v, seals, ids, jds) == plain).all()
    k = 512
    orc = ol.Risc0Oracle(); orc.initialize(H(r['control_root']), H(r['bn254_control_id']))
    ost, _ = orc.verify_batch([x.tobytes() for x in seals[:k]], [x.tobytes() for x in ids[:k]], [x.tobytes() for x in jds[:k]], threads=8)
    assert (agg[:k] == ost).all()
    v.set_aggregate_check(False)
    assert (_run_risc0_dev(v, seals, ids, jds) == plain).all()
    with pytest.raises(Exception): v.set_aggregate_check(True, sub_batch=48)
    v.close()


@pytest.mark.gpu
def test_aggregate_check_all_valid_and_only_early_rejects(real_proofs, monkeypatch):
    """All-valid batch: no sub-batch fails.  Rejects that never reach the pairing (bad selector, length, coordinate, a point off the
    curve) or whose B fails the subgroup test leave their sub-batch's aggregate check untouched: still no failure."""
    import stylus_zkvm_verifiers_amd as zkv
    from stylus_zkvm_verifiers_amd import synth
    monkeypatch.setenv('ZKV_AGG_MIN', '64')
    r = real_proofs['risc0']
    v = zkv.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
    v.set_aggregate_check(True, seed=b'\x07' * 32, sub_batch=64)
    n = 2048
    seals, ids, jds, mut, _ = _risc0_inputs(real_proofs, n, 0x5A4B56A2, 0)
    st = _run_risc0_dev(v, seals, ids, jds)
    assert (st == 0).all() and v.aggregate_counters() == (32, 0)
    v.set_aggregate_check(True, seed=b'\x08' * 32, sub_batch=256)
    st = _run_risc0_dev(v, seals, ids, jds)
    assert (st == 0).all() and v.aggregate_counters() == (32 + 8, 0)
    v.set_aggregate_check(True, seed=b'\x07' * 32, sub_batch=64)
    early = tuple(c for c in synth.MUTATION_CLASSES if c != 'flip_input')
    seals, ids, jds, mut, _ = _risc0_inputs(real_proofs, n, 0x5A4B56A3, 5, classes=early)
    st = _run_risc0_dev(v, seals, ids, jds)
    assert ((st == 0) == ~mut).all() and mut.sum() > 300
    assert v.aggregate_counters() == (64 + 8, 0)
    # one wrong public input in the whole batch: exactly one sub-batch is verified again, and only that proof is rejected
    seals, ids, jds, mut, _ = _risc0_inputs(real_proofs, n, 0x5A4B56A4, 0)
    jds[777, 3] ^= 0x10
    st = _run_risc0_dev(v, seals, ids, jds)
    assert st[777] == 1 and (st == 0).sum() == n - 1
    assert v.aggregate_counters() == (96 + 8, 1)
    v.close()


@pytest.mark.gpu
def test_aggregate_check_sp1_and_mixed(real_proofs, monkeypatch):
    import numpy as np
    import torch
    import stylus_zkvm_verifiers_amd as zkv
    from stylus_zkvm_verifiers_amd import synth
    monkeypatch.setenv('ZKV_AGG_MIN', '64')
    dev = torch.device('cuda', 0)
    s = real_proofs['sp1']
    n = 1500
    proofs, mut, mclass, flip = synth.make_batch('sp1', H(s['proof']), n, 0x5A4B56A5, pool=4, mutate_every=9)
    vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (n, 1))
    pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (n, 1)); pv[flip, -1] ^= 1
    d_p, d_vk, d_pv = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (proofs, vk, pv))
    out = []
    v = zkv.Sp1Verifier()
    for on in (False, True):
        if on: v.set_aggregate_check(True, seed=b'\x31' * 32, sub_batch=64)
        d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        v.verify_batch_dev(n, d_vk.data_ptr(), d_pv.data_ptr(), 96, d_p.data_ptr(), d_st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out.append(d_st.cpu().numpy())
    assert (out[0] == out[1]).all() and ((out[1] == 0) == ~mut).all()
    checked, failed = v.aggregate_counters()
    assert checked == _sub_batches(n, 64) and failed > 0
    # host-pointer entry point (ragged blobs, several segments)
    hst, _ = v.verify_batch([x.tobytes() for x in vk], [x.tobytes() for x in pv], [x.tobytes() for x in proofs])
    assert (np.asarray(hst) == out[0]).all()
    v.close()


@pytest.mark.gpu
def test_aggregate_check_on_generic_keys_and_a_verifier_set(real_proofs, monkeypatch):
    """Trapdoor keys with 1, 2, 3 and 6 IC points (0, 1, 2 per-proof signals: vk_x from summed scalars; 5: per-proof r vk_x), both VM
    conventions, every fifth proof with a wrong signal; and a RISC Zero verifier set (a vk_x constant per instance: per-proof form),
    the real instance among others: answers with the aggregate check on == off."""
    import numpy as np
    import stylus_zkvm_verifiers_amd as zkv
    from stylus_zkvm_verifiers_amd import synth
    monkeypatch.setenv('ZKV_AGG_MIN', '64')
    rng = random.Random(0xA6A6)
    for n_ic, vm in ((1, 'sp1'), (2, 'risc0'), (3, 'sp1'), (6, 'risc0')):
        vk, td = m.trapdoor_vk(rng, n_ic)
        proofs, sigs, exp = [], [], []
        for i in range(96):
            sig = [rng.randrange(m.R) for _ in range(n_ic - 1)]
            prf = m.trapdoor_prove(rng, td, sig, vm)
            bad = n_ic > 1 and i % 5 == 4
            if bad: sig[i % (n_ic - 1)] ^= 2
            if n_ic == 1 and i % 7 == 3:                                 # no signals to damage: damage C instead (another curve point)
                prf = (prf[0], prf[1], m.g1_mul(prf[2], 3)); bad = True
            proofs.append(m.proof_to_words(*prf)); sigs.append([m.be32(s) for s in sig]); exp.append(not bad)
        v = zkv.Groth16Verifier(m.vk_to_words(vk), n_ic, zkv.errors.VM_RISC0 if vm == 'risc0' else zkv.errors.VM_SP1)
        plain = list(v.verify_batch(proofs, sigs))
        assert plain == exp, n_ic
        v.set_aggregate_check(True, seed=b'\x55' * 32, sub_batch=32)
        assert list(v.verify_batch(proofs, sigs)) == exp, n_ic
        checked, failed = v.aggregate_counters()
        assert checked == _sub_batches(96, 32) and 0 < failed <= checked
        v.close()
    r = real_proofs['risc0']
    roots = [H(r['control_root'])] + [rng.randbytes(32) for _ in range(2)]
    ids = [H(r['bn254_control_id'])] + [rng.randrange(m.R).to_bytes(32, 'big') for _ in range(2)]
    vs = zkv.RiscZeroVerifierSet(roots, ids)
    n = 700
    seals, mut, mclass, flip = synth.make_batch('risc0', H(r['seal']), n, 0x5A4B56A7, pool=4, mutate_every=9)
    iid = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n, 1)); jds[flip, 0] ^= 1
    inst = np.array([0 if i % 4 else 1 + (i // 4) % 2 for i in range(n)], dtype=np.uint32)
    for i in range(0, n, 8):                                             # another instance's own selector spliced in: reaches the pairing and fails
        seals[i, :4] = np.frombuffer(vs.get_selector(int(inst[i])), dtype=np.uint8)
    args = (inst, [x.tobytes() for x in seals], [x.tobytes() for x in iid], [x.tobytes() for x in jds])
    st0, rv0 = vs.verify_batch(*args)
    vs.set_aggregate_check(True, seed=b'\x66' * 32, sub_batch=64)
    st1, rv1 = vs.verify_batch(*args)
    assert (np.asarray(st0) == np.asarray(st1)).all() and (np.asarray(rv0) == np.asarray(rv1)).all()
    assert {0, 1, 5} <= set(int(x) for x in st1)
    assert vs.aggregate_counters()[0] == _sub_batches(n, 64)
    vs.close()


@pytest.mark.gpu
def test_aggregate_check_behind_the_mixed_entry_point(real_proofs, monkeypatch):
    import numpy as np
    import stylus_zkvm_verifiers_amd as zkv
    from stylus_zkvm_verifiers_amd import synth
    monkeypatch.setenv('ZKV_AGG_MIN', '64')
    r, s = real_proofs['risc0'], real_proofs['sp1']
    n = 900
    s0, mut0, _, f0 = synth.make_batch('risc0', H(r['seal']), n, 0x5A4B56A8, pool=4, mutate_every=11)
    s1, mut1, _, f1 = synth.make_batch('sp1', H(s['proof']), n, 0x5A4B56A9, pool=4, mutate_every=13)
    vm, seals, ia, ib, want = [], [], [], [], []
    for i in range(n):
        for tag, ss, mm, ff in ((0, s0, mut0, f0), (1, s1, mut1, f1)):
            vm.append(tag); seals.append(ss[i].tobytes()); want.append(not mm[i])
            if tag == 0:
                jd = bytearray(H(r['journal_digest'])); jd[0] ^= int(ff[i]); ia.append(H(r['image_id'])); ib.append(bytes(jd))
            else:
                pv = bytearray(H(s['public_values'])); pv[-1] ^= int(ff[i]); ia.append(H(s['vkey'])); ib.append(bytes(pv))
    v = zkv.MixedVerifier(H(r['control_root']), H(r['bn254_control_id']))
    st0, rv0 = v.verify_batch(vm, seals, ia, ib)
    v.set_aggregate_check(True, seed=b'\x77' * 32, sub_batch=16)
    st1, rv1 = v.verify_batch(vm, seals, ia, ib)
    assert (st0 == st1).all() and (rv0 == rv1).all() and [x == 0 for x in st1] == want
    checked, failed = v.aggregate_counters()
    assert checked == 2 * _sub_batches(n, 16) and failed > 0
    v.close()


@pytest.mark.gpu
def test_aggregate_check_on_a_sharded_verifier(real_proofs, monkeypatch):
    """Three SP1 verifiers on one GPU behind one sharded context: the switch reaches every shard (each with its own derived secret), and a
    batch split over them gives the deterministic statuses."""
    import numpy as np
    import stylus_zkvm_verifiers_amd as zkv
    from stylus_zkvm_verifiers_amd import sharded, synth
    monkeypatch.setenv('ZKV_AGG_MIN', '64')
    monkeypatch.setenv('ZKV_SHARD_MIN', '64')
    s = real_proofs['sp1']
    n = 3000
    proofs, mut, _, flip = synth.make_batch('sp1', H(s['proof']), n, 0x5A4B56AA, pool=4, mutate_every=17)
    vk = [H(s['vkey'])] * n
    pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (n, 1)); pv[flip, -1] ^= 1
    v = sharded.shard([zkv.Sp1Verifier() for _ in range(3)])
    args = (vk, [x.tobytes() for x in pv], [x.tobytes() for x in proofs])
    st0, _ = v.verify_batch(*args)
    v.set_aggregate_check(True, seed=b'\x42' * 32, sub_batch=16)
    st1, _ = v.verify_batch(*args)
    assert (np.asarray(st0) == np.asarray(st1)).all() and ((np.asarray(st1) == 0) == ~mut).all()
    checked, failed = v.aggregate_counters()
    assert checked == 3 * _sub_batches(1000, 16) and failed > 0
    v.close()


@pytest.mark.gpu
def test_aggregate_check_engages_at_its_default_threshold(real_proofs, monkeypatch):
    """No environment overrides: a 2^17-proof SP1 chunk takes the aggregate check with the default sub-batch (32) and group (4) sizes, a
    chunk one proof short of the threshold does not; statuses == the per-proof path == accept <=> not mutated on all 131,072 proofs."""
    import numpy as np
    import torch
    import stylus_zkvm_verifiers_amd as zkv
    from stylus_zkvm_verifiers_amd import synth
    for k in ('ZKV_AGG_MIN', 'ZKV_AGG_GROUP', 'ZKV_CHUNK'):
        monkeypatch.delenv(k, raising=False)
    dev = torch.device('cuda', 0)
    s = real_proofs['sp1']
    n = 1 << 17
    proofs, mut, _, flip = synth.make_batch_parallel('sp1', H(s['proof']), n, 0x5A4B56AB, mutate_every=64, cache=False)
    vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (n, 1))
    pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (n, 1)); pv[flip, -1] ^= 1
    d_p, d_vk, d_pv = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (proofs, vk, pv))
    v = zkv.Sp1Verifier()
    def run(m):
        d_st = torch.full((m,), 255, dtype=torch.uint8, device=dev)
        v.verify_batch_dev(m, d_vk.data_ptr(), d_pv.data_ptr(), 96, d_p.data_ptr(), d_st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return d_st.cpu().numpy()
    plain = run(n)
    v.set_aggregate_check(True)                                 # secret from the operating system
    assert (run(n - 1) == plain[:n - 1]).all() and v.aggregate_counters() == (0, 0)
    agg = run(n)
    checked, failed = v.aggregate_counters()
    assert (agg == plain).all() and ((agg == 0) == ~mut).all()
    assert checked == n // 32 and 0 < failed < checked // 4
    v.close()


@pytest.mark.gpu
def test_aggregate_check_on_the_golden_corpus(real_proofs, verify_corpus, monkeypatch):
    """Every case of the golden corpus (points at infinity in A, B, C, a B outside the subgroup, a valid but wrong B, selector and length
    errors, ...), the RISC Zero and the SP1 ones each as one batch of several shuffled copies, through the aggregate check with every
    sub-batch and group size: status and received selector as in the fixture."""
    import numpy as np
    import stylus_zkvm_verifiers_amd as zkv
    monkeypatch.setenv('ZKV_AGG_MIN', '64')
    r = real_proofs['risc0']
    rng = random.Random(0xC0)
    r0 = [c for c in verify_corpus['cases'] if c['vm'] == 'risc0']
    s1 = [c for c in verify_corpus['cases'] if c['vm'] == 'sp1']
    order0 = [k % len(r0) for k in range(6 * len(r0))]; rng.shuffle(order0)
    order1 = [k % len(s1) for k in range(8 * len(s1))]; rng.shuffle(order1)
    v = zkv.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
    sp = zkv.Sp1Verifier()
    a0 = ([H(r0[k]['seal']) for k in order0], [H(r0[k]['image_id']) for k in order0], [H(r0[k]['journal_digest']) for k in order0])
    a1 = ([H(s1[k]['vkey']) for k in order1], [H(s1[k]['public_values']) for k in order1], [H(s1[k]['proof']) for k in order1])
    want0 = [r0[k]['status'] for k in order0]; want1 = [s1[k]['status'] for k in order1]
    st, rv = v.verify_batch(*a0); assert [int(x) for x in st] == want0
    st, rv1 = sp.verify_batch(*a1); assert [int(x) for x in st] == want1
    for g in (4, 1, 2, 8):
        monkeypatch.setenv('ZKV_AGG_GROUP', str(g))
        for sub in (16, 32, 64, 128, 256):
            v.set_aggregate_check(True, seed=bytes([sub % 251 + g]) * 32, sub_batch=sub)
            sp.set_aggregate_check(True, seed=bytes([sub % 251 + g + 1]) * 32, sub_batch=sub)
            st, rv_ = v.verify_batch(*a0)
            assert [int(x) for x in st] == want0 and (np.asarray(rv_) == np.asarray(rv)).all(), (g, sub)
            st, rv_ = sp.verify_batch(*a1)
            assert [int(x) for x in st] == want1 and (np.asarray(rv_) == np.asarray(rv1)).all(), (g, sub)
    assert v.aggregate_counters()[0] > 0 and sp.aggregate_counters()[0] > 0
    v.close(); sp.close()


@pytest.mark.gpu
def test_automatic_sub_batch_size_follows_the_failure_rate(real_proofs, monkeypatch):
    """enable = 1: sub-batches of 32 at first; once the context is idle at the start of a call the counters decide -- one proof in 100 failing
    at the pairing brings 16, a run of valid batches 128.  Statuses stay the per-proof ones throughout."""
    import stylus_zkvm_verifiers_amd as zkv
    monkeypatch.setenv('ZKV_AGG_MIN', '64')
    monkeypatch.setenv('ZKV_AGG_GROUP', '1')                     # contiguous sub-batches: the counts below are then n / sub exactly
    r = real_proofs['risc0']
    v = zkv.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
    v.set_aggregate_check(True)                                  # sub_batch = None: automatic
    n = 16384
    seals, ids, jds, mut, _ = _risc0_inputs(real_proofs, n, 0x5A4B56B1, 100, classes=('flip_input',))
    def run(s_, i_, j_):
        before = v.aggregate_counters()[0]
        st = _run_risc0_dev(v, s_, i_, j_)
        v.synchronize()
        return st, v.aggregate_counters()[0] - before
    st, k = run(seals, ids, jds); assert ((st == 0) == ~mut).all() and k == n // 32
    st, k = run(seals, ids, jds); assert ((st == 0) == ~mut).all() and k == n // 16
    seals, ids, jds, mut, _ = _risc0_inputs(real_proofs, n, 0x5A4B56B2, 0)
    st, k = run(seals, ids, jds); assert (st == 0).all() and k == n // 16          # decided on what the last call showed
    st, k = run(seals, ids, jds); assert (st == 0).all() and k == n // 128
    v.close()


@pytest.mark.gpu
def test_automatic_mode_pauses_the_check_while_most_sub_batches_fail(real_proofs, monkeypatch):
    """enable = 1 under a stream of bad proofs (one in 4 fails at the pairing: 99 % of the sub-batches of 16 would be verified twice): the
    check switches itself off for 8 chunks, probes once with size 16, stays off for 16 more, ... -- and comes back for good once the
    proofs are valid again.  Statuses are the per-proof ones in every call."""
    import stylus_zkvm_verifiers_amd as zkv
    monkeypatch.setenv('ZKV_AGG_MIN', '64')
    monkeypatch.setenv('ZKV_AGG_GROUP', '1')
    r = real_proofs['risc0']
    v = zkv.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
    v.set_aggregate_check(True)
    n = 16384
    bad = _risc0_inputs(real_proofs, n, 0x5A4B56D1, 4, classes=('flip_input',))
    good = _risc0_inputs(real_proofs, n, 0x5A4B56D2, 0)
    def run(b):
        before = v.aggregate_counters()[0]
        st = _run_risc0_dev(v, b[0], b[1], b[2])
        v.synchronize()
        assert ((st == 0) == ~b[3]).all()
        return v.aggregate_counters()[0] - before
    assert run(bad) == n // 32                                   # no evidence yet
    ks = [run(bad) for _ in range(8 + 1 + 16 + 1)]
    assert ks[:8] == [0] * 8 and ks[8] == n // 16, ks            # eight chunks without the check, then the probe
    assert ks[9:25] == [0] * 16 and ks[25] == n // 16, ks        # still bad: twice as long a pause
    ks = [run(good) for _ in range(32 + 1 + 3)]
    assert ks[:32] == [0] * 32 and ks[32] == n // 16, ks         # the pause that was decided on the bad stream, then the probe on the good one
    assert all(k > 0 for k in ks[33:]), ks                       # the check is back (and grows its sub-batches: fewer of them per call)
    v.close()


@pytest.mark.gpu
def test_aggregate_check_rekeys_an_os_drawn_secret(real_proofs, monkeypatch):
    """seed32 = NULL: the secret comes from getrandom and is drawn afresh every ZKV_AGG_REKEY chunks (default 1,024; 2 here, with 64-proof
    chunks, so that a 1,000-proof batch passes through eight secrets): statuses as without the check."""
    import stylus_zkvm_verifiers_amd as zkv
    monkeypatch.setenv('ZKV_AGG_MIN', '64')
    monkeypatch.setenv('ZKV_AGG_REKEY', '2')
    monkeypatch.setenv('ZKV_CHUNK', '64')
    r = real_proofs['risc0']
    v = zkv.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
    n = 1000
    seals, ids, jds, mut, _ = _risc0_inputs(real_proofs, n, 0x5A4B56D3, 9)
    plain = _run_risc0_dev(v, seals, ids, jds)
    v.set_aggregate_check(True, sub_batch=16)                    # seed = None: operating system
    agg = _run_risc0_dev(v, seals, ids, jds)
    assert (agg == plain).all() and ((agg == 0) == ~mut).all() and v.aggregate_counters()[0] == (n // 64) * 4       # the last, 40-proof chunk is below ZKV_AGG_MIN
    v.close()


@pytest.mark.gpu
def test_cancelling_pairs_are_rejected_because_every_proof_has_its_own_coefficient(real_proofs, monkeypatch):
    """What the soundness of the aggregate check rests on: DISTINCT secret coefficients per proof.  Two copies of a valid proof with C + D and
    C - D in place of C are both invalid, but e(C + D, delta) e(C - D, delta) = e(C, delta)^2: with EQUAL coefficients the pair's errors
    cancel and any aggregate check passes.  Such pairs are placed in the same sub-batch and in the same Miller group -- members l and
    l + 32 of a 64-proof block (one group for 2, 4 or 8 proofs per accumulator) and neighbours 2k, 2k + 1 (one sub-batch for contiguous
    sub-batches) -- for every group size and sub-batch size: all of them must be rejected, the valid proofs between them accepted, and
    the counters must show failed sub-batches."""
    import numpy as np
    import stylus_zkvm_verifiers_amd as zkv
    from stylus_zkvm_verifiers_amd import synth
    monkeypatch.setenv('ZKV_AGG_MIN', '64')
    r = real_proofs['risc0']
    n = 512
    seals, ids, jds, mut, _ = _risc0_inputs(real_proofs, n, 0x5A4B56D4, 0)
    seals = np.ascontiguousarray(seals).copy()
    rng = synth.SplitMix64(0xD1FF)
    bad = np.zeros(n, dtype=bool)
    def twist(i, j):
        # proofs i and j become copies of proof i with C + D and C - D
        a, b, c = synth.parse_seal(seals[i].tobytes())
        d = synth.g1_mul((1, 2), rng.scalar())
        nd = (d[0], (-d[1]) % synth.P)
        for row, cc in ((i, synth.g1_add(c, d)), (j, synth.g1_add(c, nd))):
            synth._write(seals, row, seals[i, :4].tobytes(), synth.seal_words(a, b, cc))
            bad[row] = True
    for blk in range(0, n, 64):
        for l in (1, 5, 12, 30):
            twist(blk + l, blk + l + 32)
        for k in (8, 9, 20):
            twist(blk + 2 * k, blk + 2 * k + 1)
    v = zkv.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
    plain = _run_risc0_dev(v, seals, ids, jds)
    assert ((plain == 0) == ~bad).all() and (plain[bad] == 1).all() and bad.sum() == 8 * 14
    for g in (1, 2, 4, 8):
        monkeypatch.setenv('ZKV_AGG_GROUP', str(g))
        for sub in (16, 32, 64, 128, 256):
            before = v.aggregate_counters() if g != 1 or sub != 16 else (0, 0)
            v.set_aggregate_check(True, seed=bytes([g, sub & 255]) * 16, sub_batch=sub)
            agg = _run_risc0_dev(v, seals, ids, jds)
            after = v.aggregate_counters()
            assert (agg == plain).all(), (g, sub, np.flatnonzero(agg != plain)[:8])
            assert after[1] > before[1], (g, sub)
    v.close()
