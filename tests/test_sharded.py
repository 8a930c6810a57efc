"""Sharded (multi-device) contexts, SURVEY.md 8(b) `device_mask` / 8(e): creation rules and the C++ harness on the CPU; on the GPU
two logical shards mapped to device 0 (and, with ZKV_SHARD_FORCE_STAGING, the peer-copy path a second GPU would take) against a
single-device context of the same verifier.  The reference verifies one proof per call (risc0/verifier.rs:78-92,
sp1/verifier.rs:39-46): splitting a batch into contiguous ranges changes nothing a proof can observe."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = bytes.fromhex


def _harness(real_proofs, n, env=None):
    from stylus_zkvm_verifiers_amd import build
    build.build(verbose=False)
    src = os.path.join(ROOT, 'tests', 'host_cpp', 'test_sharded.cpp')
    exe = os.path.join(ROOT, 'tests', 'host_cpp', 'test_sharded')
    libdir = os.path.join(ROOT, 'stylus_zkvm_verifiers_amd')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-o', exe, src, '-L' + libdir, '-lzkv_mi355x', '-Wl,-rpath,' + libdir])
    r = real_proofs['risc0']
    out = subprocess.check_output([exe, r['control_root'], r['bn254_control_id'], r['seal'], r['image_id'], r['journal_digest'], str(n)],
                                  timeout=600, env=dict(os.environ, **(env or {}))).decode()
    return dict(kv.split('=', 1) for kv in out.split())


def test_sharded_creation_rules_through_the_c_abi(real_proofs):
    kv = _harness(real_proofs, 8)
    for k in ('refuse_mixed_kinds', 'refuse_duplicate', 'refuse_uninitialised', 'refuse_different_params', 'refuse_empty', 'sharded', 'no_mask'):
        assert kv[k] == '1', k
    assert (kv['shards'], kv['dev1'], kv['multi_shards'], kv['plain_shards']) == ('2', '0', '1', '0')
    assert (kv['peer0'], kv['peer1'], kv['peer_bad_index'], kv['peer_plain']) == ('2', '2', '1', '1')       # zkv_ctx_shard_peer_access
    assert kv['selector'] == real_proofs['risc0']['selector'] and kv['initialized'] == '1'
    if 'rc_no_device' in kv:
        assert kv['rc_no_device'] == '-2'                 # ZKV_ERR_NO_DEVICE from the shards: no CPU fallback


def test_python_shard_wrapper_host_side(real_proofs):
    import stylus_zkvm_verifiers_amd as z
    r = real_proofs['risc0']
    mk = lambda: (lambda v: (v.initialize(H(r['control_root']), H(r['bn254_control_id'])), v)[1])(z.RiscZeroVerifier(0))
    a, b = mk(), mk()
    s = z.shard([a, b])
    assert a._h is None and b._h is None and z.shard_count(s) == 2 and z.shard_devices(s) == [0, 0] and z.shard_peer_access(s) == [2, 2]
    assert s.get_selector().hex() == r['selector'] and s.is_initialized()
    with pytest.raises(ValueError):
        z.shard([mk(), z.Sp1Verifier(0)])
    with pytest.raises(ValueError):
        z.shard([])
    m2 = z.shard([z.MixedVerifier(H(r['control_root']), H(r['bn254_control_id']), 0) for _ in range(3)])
    assert z.shard_count(m2) == 3
    s.close(); m2.close()


@pytest.mark.gpu
@pytest.mark.parametrize('env', [{}, {'ZKV_SHARD_MIN': '1'}])
def test_cpp_harness_two_logical_shards_on_one_gpu(real_proofs, env):
    kv = _harness(real_proofs, 5000, env)
    assert (kv['rc'], kv['rc_single'], kv['rc_multi']) == ('0', '0', '0')
    assert kv['ok'] == kv['want_ok'] and kv['same'] == kv['n'] == '5000' and kv['status3'] == '4'
    assert (kv['single_proof_rc'], kv['single_proof_status'], kv['sync']) == ('0', '0', '0')


@pytest.fixture(scope='module')
def zkv():
    import stylus_zkvm_verifiers_amd as z
    assert z.device_count() >= 1, 'no gfx950 device visible'
    return z


@pytest.mark.gpu
def test_sharded_host_batches_equal_single_device_on_the_corpus(zkv, real_proofs, verify_corpus):
    """Every corpus case (ragged seals, wrong selectors, malformed points, ragged public values) through verifiers sharded three ways
    on device 0 with ZKV_SHARD_MIN = 1, so that the ranges really split: statuses and received selectors equal the golden values."""
    os.environ['ZKV_SHARD_MIN'] = '1'
    try:
        r = real_proofs['risc0']
        mk = lambda: (lambda v: (v.initialize(H(verify_corpus['risc0_ctx']['control_root']), H(verify_corpus['risc0_ctx']['bn254_control_id'])), v)[1])(zkv.RiscZeroVerifier(0))
        r0 = zkv.shard([mk(), mk(), mk()])
        sp = zkv.shard([zkv.Sp1Verifier(0), zkv.Sp1Verifier(0), zkv.Sp1Verifier(0)])
        c0 = [c for c in verify_corpus['cases'] if c['vm'] == 'risc0']
        c1 = [c for c in verify_corpus['cases'] if c['vm'] == 'sp1']
        st, rv = r0.verify_batch([H(c['seal']) for c in c0], [H(c['image_id']) for c in c0], [H(c['journal_digest']) for c in c0])
        for c, s, x in zip(c0, st, rv):
            assert int(s) == c['status'] and bytes(x).hex() == (c['received'] or '00000000'), c['name']
        st, rv = sp.verify_batch([H(c['vkey']) for c in c1], [H(c['public_values']) for c in c1], [H(c['proof']) for c in c1])
        for c, s, x in zip(c1, st, rv):
            assert int(s) == c['status'] and bytes(x).hex() == (c['received'] or '00000000'), c['name']
        assert r0.verify(H(r['seal']), H(r['image_id']), H(r['journal_digest'])) is True          # n = 1: shard 0
        r0.close(); sp.close()
    finally:
        del os.environ['ZKV_SHARD_MIN']


@pytest.mark.gpu
@pytest.mark.parametrize('staging', ['0', '1'])
def test_sharded_device_batches_equal_single_device(zkv, real_proofs, staging):
    """HBM-resident batches (SP1, RISC Zero, mixed; 1/8 mutated) through two logical shards on device 0 -- directly on the caller's rows
    and through the staging path a shard on another GPU takes (hipMemcpyPeerAsync in two pieces, statuses copied back) -- on a
    caller stream: after synchronising THAT stream the statuses equal the single-device verifier's and the construction."""
    import torch
    from stylus_zkvm_verifiers_amd import parallel, synth
    env = {'ZKV_SHARD_MIN': '256', 'ZKV_SHARD_FORCE_STAGING': staging, 'ZKV_SHARD_FIRST_PIECE': '300'}
    os.environ.update(env)
    try:
        dev = torch.device('cuda', 0)
        r, s = real_proofs['risc0'], real_proofs['sp1']
        k0, k1 = 2100, 1901
        s0, m0, _, f0 = synth.make_batch('risc0', H(r['seal']), k0, 0x5A4B56B1, pool=4, mutate_every=8)
        s1, m1, _, f1 = synth.make_batch('sp1', H(s['proof']), k1, 0x5A4B56B2, pool=4, mutate_every=8)
        ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (k0, 1))
        jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (k0, 1)); jds[f0, 0] ^= 1
        vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (k1, 1))
        pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (k1, 1)); pv[f1, -1] ^= 1
        up = lambda *xs: [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in xs]
        side = torch.cuda.Stream()
        mk0 = lambda: (lambda v: (v.initialize(H(r['control_root']), H(r['bn254_control_id'])), v)[1])(zkv.RiscZeroVerifier(0))
        # RISC Zero
        d = up(s0, ids, jds)
        st = torch.zeros(k0, dtype=torch.uint8, device=dev); rvt = torch.full((k0, 4), 255, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        v = zkv.shard([mk0(), mk0()])
        v.verify_batch_dev(k0, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), st.data_ptr(), rvt.data_ptr(), side.cuda_stream)
        side.synchronize()
        with torch.cuda.stream(side):
            got, grv = st.cpu().numpy(), rvt.cpu().numpy()
        one = mk0()
        st1 = torch.zeros(k0, dtype=torch.uint8, device=dev); rv1 = torch.full((k0, 4), 255, dtype=torch.uint8, device=dev)
        one.verify_batch_dev(k0, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), st1.data_ptr(), rv1.data_ptr(), side.cuda_stream)
        side.synchronize()
        assert ((got == 0) == ~m0).all() and (got == st1.cpu().numpy()).all() and (grv == rv1.cpu().numpy()).all()
        v.close(); one.close()
        # SP1 (NULL stream: the shards' own streams, then zkv_ctx_synchronize)
        d = up(vk, pv, s1)
        st = torch.zeros(k1, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        v = zkv.shard([zkv.Sp1Verifier(0), zkv.Sp1Verifier(0)])
        v.verify_batch_dev(k1, d[0].data_ptr(), d[1].data_ptr(), 96, d[2].data_ptr(), st.data_ptr(), 0, 0)
        v.synchronize()
        with torch.cuda.stream(side):
            got = st.cpu().numpy()
        assert ((got == 0) == ~m1).all() and 0 < (got != 0).sum() < k1
        v.close()
        # mixed
        vm, seals, a, b, perm = parallel.interleave([(0, s0, ids, jds), (1, s1, vk, pv)], 0x5A4B5603)
        mut = np.concatenate([m0, m1])[perm]
        n = k0 + k1
        d = up(vm, seals, a, b)
        st = torch.zeros(n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        v = zkv.shard([zkv.MixedVerifier(H(r['control_root']), H(r['bn254_control_id']), 0) for _ in range(2)])
        v.verify_batch_dev(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 96, 96, st.data_ptr(), 0, side.cuda_stream)
        side.synchronize()
        with torch.cuda.stream(side):
            got = st.cpu().numpy()
        assert ((got == 0) == ~mut).all()
        hst, _ = v.verify_batch(vm.tolist(), [x.tobytes() for x in seals], [x.tobytes() for x in a], [b[i, :32 if vm[i] == 0 else 96].tobytes() for i in range(n)])
        assert (hst == got).all()
        v.close()
    finally:
        for k in env:
            del os.environ[k]


@pytest.mark.gpu
def test_back_to_back_staged_batches_do_not_overwrite_each_other(zkv, real_proofs):
    """Two DIFFERENT HBM-resident batches enqueued back to back on a sharded verifier with no caller stream (NULL: the shards' own
    streams), both through the staging rows a shard on another GPU uses (ZKV_SHARD_FORCE_STAGING): the second call's copies into those
    rows must wait for the first call's kernels.  After one zkv_ctx_synchronize both status arrays are the construction's; repeated
    a few times, and with many small calls in flight."""
    import torch
    from stylus_zkvm_verifiers_amd import synth
    env = {'ZKV_SHARD_MIN': '256', 'ZKV_SHARD_FORCE_STAGING': '1'}
    os.environ.update(env)
    try:
        dev = torch.device('cuda', 0)
        s = real_proofs['sp1']
        k = 6000
        up = lambda *xs: [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in xs]
        batches = []
        for seed, every in ((0x5A4B56E1, 3), (0x5A4B56E2, 5), (0x5A4B56E3, 2)):
            seals, mut, _, flip = synth.make_batch('sp1', H(s['proof']), k, seed, pool=4, mutate_every=every)
            vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (k, 1))
            pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (k, 1)); pv[flip, -1] ^= 1
            batches.append((up(vk, pv, seals), mut, torch.full((k,), 255, dtype=torch.uint8, device=dev)))
        torch.cuda.synchronize()
        v = zkv.shard([zkv.Sp1Verifier(0), zkv.Sp1Verifier(0), zkv.Sp1Verifier(0)])
        for rep in range(3):
            for d, mut, st in batches:
                st.fill_(255)
            torch.cuda.synchronize()
            for d, mut, st in batches:                              # no synchronisation between the calls
                v.verify_batch_dev(k, d[0].data_ptr(), d[1].data_ptr(), 96, d[2].data_ptr(), st.data_ptr(), 0, 0)
            v.synchronize()
            for j, (d, mut, st) in enumerate(batches):
                got = st.cpu().numpy()
                assert ((got == 0) == ~mut).all(), (rep, j, int(((got == 0) != ~mut).sum()))
        assert zkv.shard_peer_access(v) == [2, 2, 2]             # every shard on the source GPU: no peer access was needed
        v.close()
    finally:
        for kk in env:
            del os.environ[kk]


@pytest.mark.gpu
def test_caller_supplied_keys_travel_as_context_blobs_and_shard(zkv):
    """The caller-keyed verifiers over several devices (missing #4): a generic Groth16 trapdoor key
    (`Groth16Verifier::verify_proof_with_key`, common/groth16.rs:23-49) and the PLONK key of the golden corpus (a) as context blobs
    through `parallel.sharded_step` + `make_verifier` (one rank: the blob path without a second GPU) and (b) as sharded contexts with
    three logical shards on device 0.  Results equal the directly constructed single-device verifier / the golden statuses."""
    import json
    import random
    import torch
    import spec_model as m
    from stylus_zkvm_verifiers_amd import parallel
    os.environ['ZKV_SHARD_MIN'] = '4'
    try:
        rng = random.Random(91)
        vk, td = m.trapdoor_vk(rng, 4)
        proofs, sigs, exp = [], [], []
        for i in range(40):
            sig = [rng.randrange(m.R) for _ in range(3)]
            prf = m.trapdoor_prove(rng, td, sig, 'sp1')
            if i % 4 == 3:
                sig[i % 3] ^= 2
            proofs.append(m.proof_to_words(*prf)); sigs.append([m.be32(s) for s in sig]); exp.append(i % 4 != 3)
        vkb = m.vk_to_words(vk)
        # (a) context blob -> verifier on "every rank"
        blob = parallel.pack_context(parallel.CTX_GROTH16, vk=vkb, n_ic=4, vm_type=zkv.errors.VM_SP1)
        dev = torch.device('cuda', 0)
        rows = [torch.from_numpy(np.frombuffer(b''.join(proofs), dtype=np.uint8).reshape(40, 256).copy()),
                torch.from_numpy(np.frombuffer(b''.join(b''.join(s) for s in sigs), dtype=np.uint8).reshape(40, 96).copy())]
        state = {}

        def verify_fn(b, p_t, s_t):
            v = state.setdefault('v', parallel.make_verifier(b, 0))
            pr = p_t.cpu().numpy(); sg = s_t.cpu().numpy()
            ok = v.verify_batch([pr[i].tobytes() for i in range(len(pr))], [[sg[i, 32 * k:32 * k + 32].tobytes() for k in range(3)] for i in range(len(sg))])
            return torch.from_numpy((~ok).astype(np.uint8)).to(p_t.device)          # 0 = verified, as a status byte
        out, _ = parallel.sharded_step(blob, rows, 40, verify_fn, torch.device('cpu'), torch.device('cpu'))
        assert [int(x) == 0 for x in out] == exp
        state['v'].close()
        # (b) three logical shards behind zkv_groth16_verify_batch
        sv = zkv.shard([zkv.Groth16Verifier(vkb, 4, zkv.errors.VM_SP1, 0) for _ in range(3)])
        assert zkv.shard_count(sv) == 3 and list(sv.verify_batch(proofs, sigs)) == exp
        sv.close()
        with pytest.raises(ValueError):                                 # shards of one verifier: a different key is refused
            vk2, _ = m.trapdoor_vk(rng, 4)
            zkv.shard([zkv.Groth16Verifier(vkb, 4, zkv.errors.VM_SP1, 0), zkv.Groth16Verifier(m.vk_to_words(vk2), 4, zkv.errors.VM_SP1, 0)])
        # PLONK (parity unpinned): the golden corpus through a blob-built verifier and through three shards
        cases = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'plonk_cases.json')))
        pk, vh = H(cases['vk']), H(cases['verifier_hash'])
        cs = cases['cases']
        pv = parallel.make_verifier(parallel.pack_context(parallel.CTX_PLONK, vk=pk, verifier_hash=vh), 0)
        sh = zkv.shard([zkv.Sp1PlonkVerifier(pk, vh, 0) for _ in range(3)])
        for v in (pv, sh):
            st, rv = v.verify_batch([H(c['vkey']) for c in cs], [H(c['public_values']) for c in cs], [H(c['proof']) for c in cs])
            assert [int(x) for x in st] == [c['status'] for c in cs]
            assert [bytes(x).hex() for x in rv] == [(c['received'] or '00000000') for c in cs]
        pv.close(); sh.close()
    finally:
        del os.environ['ZKV_SHARD_MIN']
