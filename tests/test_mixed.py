"""Mixed batches (BASELINE.json configs[3]; per-proof `VMType`, /root/reference/contracts/src/common/types.rs:24-26):
the seeded permutation and the distribution step of config 4 on CPU (gloo, world_size 2), the per-proof-VM entry point and the
config-4 job on the GPU.  Accept side pinned by the two real proofs; everything else agreement with the oracle (parity unpinned)."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

H = bytes.fromhex
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


# ---------------------------------------------------------------- CPU: permutation, interleave, distribution over gloo
def test_seeded_permutation_is_a_fixed_permutation():
    from stylus_zkvm_verifiers_amd import parallel
    p = parallel.seeded_permutation(1000, 0x5A4B5603)
    assert sorted(p.tolist()) == list(range(1000))
    assert (p == parallel.seeded_permutation(1000, 0x5A4B5603)).all()
    assert (p != parallel.seeded_permutation(1000, 0x5A4B5604)).any()
    # pinned values: the permutation must not drift with numpy versions or platforms (SplitMix64 keys, stable argsort)
    assert parallel.seeded_permutation(8, 1).tolist() == [int(x) for x in np.argsort(parallel.splitmix64(np.arange(8, dtype=np.uint64) + np.uint64(1)), kind='stable')]
    assert int(parallel.splitmix64(np.array([0], dtype=np.uint64))[0]) == 0xE220A8397B1DCDAF       # SplitMix64 reference output for seed 0
    assert len(parallel.seeded_permutation(0, 3)) == 0


def test_interleave_keeps_rows_together_and_mixes_the_vms():
    from stylus_zkvm_verifiers_amd import parallel
    k0, k1 = 300, 212
    r0 = (0, np.full((k0, 260), 1, np.uint8), np.arange(k0 * 32, dtype=np.uint32).astype(np.uint8).reshape(k0, 32), np.full((k0, 32), 7, np.uint8))
    s1 = (1, np.full((k1, 260), 2, np.uint8), np.full((k1, 32), 9, np.uint8), np.full((k1, 96), 5, np.uint8))
    vm, seals, a, b, perm = parallel.interleave([r0, s1], 0x5A4B5603)
    assert b.shape == (k0 + k1, 96) and int((vm == 0).sum()) == k0
    assert ((seals[:, 0] == 1) == (vm == 0)).all() and ((b[:, 0] == 7) == (vm == 0)).all() and (b[vm == 0][:, 32:] == 0).all()
    src = perm[vm == 0]
    assert (a[vm == 0] == r0[2][src]).all()                       # rows travel as a whole
    half = vm[:(k0 + k1) // 2]
    assert 0.3 < (half == 0).mean() < 0.8                          # both VMs in both halves: every shard is mixed


def _dist_worker(rank, world, port, n, q):
    import torch
    import torch.distributed as dist
    from stylus_zkvm_verifiers_amd import parallel
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cpu')
    rng = np.random.default_rng(5)
    root = None
    if rank == 0:
        vm = torch.from_numpy(rng.integers(0, 2, n).astype(np.uint8))
        root = [vm.reshape(n, 1)] + [torch.from_numpy(rng.integers(0, 256, (n, w)).astype(np.uint8)) for w in (260, 32, 96)]
    seen = {}

    def verify_fn(params, vm_t, seals_t, a_t, b_t):               # stand-in "verifier": a checksum of the proof's own row
        seen['params'] = params
        seen['rows'] = int(vm_t.numel())
        s = seals_t.to(torch.int64).sum(1) + 3 * a_t.to(torch.int64).sum(1) + 5 * b_t.to(torch.int64).sum(1) + 7 * vm_t.to(torch.int64)
        return (s % 251).to(torch.uint8)

    out, t = parallel.mixed_step(bytes(range(64)) if rank == 0 else None, root, n, verify_fn, dev, dev)
    ok = seen['params'] == bytes(range(64)) and seen['rows'] == parallel.shard_bounds(n, world, rank)[1] - parallel.shard_bounds(n, world, rank)[0]
    if rank == 0:
        want = verify_fn(None, root[0].reshape(-1), root[1], root[2], root[3])
        ok = ok and out is not None and bool((out == want).all()) and set(t) == {'distribute', 'verify', 'collect'}
    else:
        ok = ok and out is None
    # gather_rows: the set-up collective of bench.py --workload mixed
    lo, hi = parallel.shard_bounds(n, world, rank)
    mine = (torch.arange(lo, hi).reshape(-1, 1) % 200).to(torch.uint8).repeat(1, 3)
    allr = parallel.gather_rows(mine, n, dev)
    if rank == 0:
        ok = ok and bool((allr == (torch.arange(n).reshape(-1, 1) % 200).to(torch.uint8).repeat(1, 3)).all())
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n', [64, 9])
def test_config4_step_over_gloo_world2(n):
    """broadcast of the parameters + scatter of the four row arrays + per-shard verify + status gather in the original order."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_mixed_entry_points_reject_bad_arguments_without_a_device(real_proofs):
    import ctypes as C
    from stylus_zkvm_verifiers_amd import _lib
    L = _lib.lib()
    r = real_proofs['risc0']
    assert L.zkv_mixed_ctx_create(None, H(r['bn254_control_id']), 0) is None
    h = L.zkv_mixed_ctx_create(H(r['control_root']), H(r['bn254_control_id']), 0)
    assert h and L.zkv_ctx_vm(h) == 5
    sel = C.create_string_buffer(4)
    assert L.zkv_risc0_get_selector(L.zkv_mixed_ctx_risc0(h), sel) == 0 and sel.raw.hex() == r['selector']
    st = np.zeros(2, dtype=np.uint8)
    off = np.array([0, 260, 520], dtype=np.uint64)
    vm = np.array([0, 1], dtype=np.uint8)
    seals = np.zeros(520, dtype=np.uint8); a = np.zeros(64, dtype=np.uint8); b = np.zeros(128, dtype=np.uint8)
    call = lambda boff: L.zkv_mixed_verify_batch(h, 2, vm.ctypes.data, seals.ctypes.data, off.ctypes.data, a.ctypes.data, b.ctypes.data,
                                                 np.asarray(boff, dtype=np.uint64).ctypes.data, st.ctypes.data, None)
    assert call([0, 31, 127]) == _lib.ERR_INVALID_ARG               # a RISC Zero journal digest is exactly 32 bytes
    assert call([0, 64, 32]) == _lib.ERR_INVALID_ARG                # offsets run backwards
    assert L.zkv_mixed_verify_batch(h, 0, None, None, None, None, None, None, None, None) == 0
    assert L.zkv_mixed_verify_batch(h, 1, None, None, None, None, None, None, None, None) == _lib.ERR_INVALID_ARG
    sp = L.zkv_sp1_ctx_create(0)
    assert L.zkv_mixed_verify_batch(sp, 0, None, None, None, None, None, None, None, None) == _lib.ERR_WRONG_CTX
    assert L.zkv_mixed_verify_batch_dev(h, 4, 1, 1, 1, 1, 16, 0, 1, None, None) == _lib.ERR_INVALID_ARG      # b_stride < 32
    assert L.zkv_mixed_verify_batch_dev(h, 4, 1, 1, 1, 1, 96, 97, 1, None, None) == _lib.ERR_INVALID_ARG     # pv_len > b_stride
    import torch
    if not torch.cuda.is_available():
        assert call([0, 32, 128]) == _lib.ERR_NO_DEVICE             # no CPU fallback
    L.zkv_ctx_destroy(sp); L.zkv_ctx_destroy(h)


# ---------------------------------------------------------------- GPU
@pytest.fixture(scope='module')
def zkv():
    import stylus_zkvm_verifiers_amd as z
    assert z.device_count() >= 1, 'no gfx950 device visible'
    return z


@pytest.fixture(scope='module')
def mixed(zkv, real_proofs):
    r = real_proofs['risc0']
    v = zkv.MixedVerifier(H(r['control_root']), H(r['bn254_control_id']))
    yield v
    v.close()


@pytest.mark.gpu
def test_mixed_corpus_in_permuted_order_matches_golden_and_oracle(zkv, mixed, verify_corpus):
    """Every case of the verify corpus (both VMs: ragged seal lengths 0..292, wrong selectors, malformed points, ragged public values)
    in ONE batch, interleaved by the seeded permutation, through the ragged host entry point: each status / received selector equals
    the golden value and what the oracle's verifier of that proof's VM returns.  Tags that are no VMType get status 7."""
    import oracle_lib as ol
    from stylus_zkvm_verifiers_amd import parallel
    orc = ol.Risc0Oracle()
    orc.initialize(H(verify_corpus['risc0_ctx']['control_root']), H(verify_corpus['risc0_ctx']['bn254_control_id']))
    cases = list(verify_corpus['cases'])
    perm = parallel.seeded_permutation(len(cases), 0x5A4B5603)
    cases = [cases[i] for i in perm]
    vm = [0 if c['vm'] == 'risc0' else 1 for c in cases]
    seals = [H(c['seal'] if c['vm'] == 'risc0' else c['proof']) for c in cases]
    a = [H(c['image_id'] if c['vm'] == 'risc0' else c['vkey']) for c in cases]
    b = [H(c['journal_digest'] if c['vm'] == 'risc0' else c['public_values']) for c in cases]
    assert 0 < sum(vm) < len(vm)
    # two strangers in the middle of the batch
    vm[5:5] = [2]; seals[5:5] = [seals[0]]; a[5:5] = [a[0]]; b[5:5] = [b'']
    vm.append(255); seals.append(b''); a.append(bytes(32)); b.append(b'xyz')
    cases[5:5] = [None]; cases.append(None)
    st, rv = mixed.verify_batch(vm, seals, a, b)
    for c, s, r, sl, ia, ib in zip(cases, st, rv, seals, a, b):
        if c is None:
            assert int(s) == 7 and bytes(r) == bytes(4)
            continue
        if c['vm'] == 'risc0':
            ost, orecv = orc.verify(sl, ia, ib)
        else:
            ost, orecv = ol.sp1_verify_proof(ia, ib, sl)
        assert int(s) == c['status'] == ost, c['name']
        assert bytes(r).hex() == (c['received'] or '00000000'), c['name']
    assert (np.asarray(st) == 0).sum() >= 2                        # both real proofs are in the corpus: pinned accepts


@pytest.mark.gpu
def test_mixed_device_path_equals_separate_verifiers_and_oracle(zkv, mixed, real_proofs):
    """HBM-resident mixed batch (seeded synthetic proofs of both VMs, every mutation class, 1/8 mutated, interleaved): statuses equal
    the two single-VM verifiers on the demultiplexed halves, the CPU oracle on a sample, and accept <=> not mutated everywhere.
    The batch size is no multiple of the partition block (256) or of a wavefront."""
    import torch
    import oracle_lib as ol
    from stylus_zkvm_verifiers_amd import parallel, synth
    dev = torch.device('cuda', 0)
    r, s = real_proofs['risc0'], real_proofs['sp1']
    k0, k1 = 1500, 1101
    s0, m0, _, f0 = synth.make_batch('risc0', H(r['seal']), k0, 0x5A4B5671, pool=4, mutate_every=8)
    s1, m1, _, f1 = synth.make_batch('sp1', H(s['proof']), k1, 0x5A4B5672, pool=4, mutate_every=8)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (k0, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (k0, 1)); jds[f0, 0] ^= 1
    vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (k1, 1))
    pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (k1, 1)); pv[f1, -1] ^= 1
    vm, seals, a, b, perm = parallel.interleave([(0, s0, ids, jds), (1, s1, vk, pv)], 0x5A4B5603)
    mut = np.concatenate([m0, m1])[perm]
    n = k0 + k1
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (vm, seals, a, b)]
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev); d_rv = torch.full((n, 4), 255, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    mixed.verify_batch_dev(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 96, 96, d_st.data_ptr(), d_rv.data_ptr(), stream)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy(); rv = d_rv.cpu().numpy()
    assert ((st == 0) == ~mut).all() and set(st[mut]) == {1, 5}
    # the two single-VM verifiers on the demultiplexed halves
    r0v = zkv.RiscZeroVerifier(); r0v.initialize(H(r['control_root']), H(r['bn254_control_id']))
    spv = zkv.Sp1Verifier()
    i0, i1 = np.where(vm == 0)[0], np.where(vm == 1)[0]
    st0, rv0 = r0v.verify_batch([seals[i].tobytes() for i in i0], [a[i].tobytes() for i in i0], [b[i, :32].tobytes() for i in i0])
    st1, rv1 = spv.verify_batch([a[i].tobytes() for i in i1], [b[i].tobytes() for i in i1], [seals[i].tobytes() for i in i1])
    assert (st[i0] == st0).all() and (st[i1] == st1).all() and (rv[i0] == rv0).all() and (rv[i1] == rv1).all()
    # the ragged host entry point gives the same bytes
    hst, hrv = mixed.verify_batch(vm.tolist(), [x.tobytes() for x in seals], [x.tobytes() for x in a],
                                  [b[i, :32 if vm[i] == 0 else 96].tobytes() for i in range(n)])
    assert (hst == st).all() and (hrv == rv).all()
    # CPU oracle on the first 192 proofs of the interleaved order
    orc = ol.Risc0Oracle(); orc.initialize(H(r['control_root']), H(r['bn254_control_id']))
    for i in range(192):
        want = orc.verify(seals[i].tobytes(), a[i].tobytes(), b[i, :32].tobytes())[0] if vm[i] == 0 else \
            ol.sp1_verify_proof(a[i].tobytes(), b[i].tobytes(), seals[i].tobytes())[0]
        assert int(st[i]) == want, i
    # every kernel mapping behind the tag gives the same bytes (the knob reaches both verifiers of the mixed context)
    for lanes in (2, 16, 64, 128):
        mixed.set_lanes_per_proof(lanes)
        d_st.fill_(255)
        mixed.verify_batch_dev(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 96, 96, d_st.data_ptr(), d_rv.data_ptr(), stream)
        torch.cuda.synchronize()
        assert (d_st.cpu().numpy() == st).all(), lanes
    mixed.set_lanes_per_proof(0)
    # one-VM batches and the empty batch go through the same entry point
    only = torch.zeros(n, dtype=torch.uint8, device=dev)
    keep = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (s0, ids, jds)]
    mixed.verify_batch_dev(k0, only.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), 32, 0, d_st.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    assert ((d_st.cpu().numpy()[:k0] == 0) == ~m0).all()
    st_e, _ = mixed.verify_batch([], [], [], [])
    assert len(st_e) == 0
    r0v.close(); spv.close()


@pytest.mark.gpu
def test_all_sp1_batch_then_synchronize_without_reserve(zkv, real_proofs):
    """A FRESH mixed verifier (no reserve), one batch that holds SP1 proofs only, then MixedVerifier.synchronize() -- not
    torch.cuda.synchronize(): the statuses must be final when it returns.  (The RISC Zero child of such a verifier is never set up;
    a synchronize() that waits on that child alone returns at once and the caller reads stale bytes -- on a zero-filled buffer a stale
    byte is 0 = accept.)  Also: the stage times of that call hold no contribution of the unused child."""
    import torch
    from stylus_zkvm_verifiers_amd import synth
    dev = torch.device('cuda', 0)
    r, s = real_proofs['risc0'], real_proofs['sp1']
    n = 20000                                             # lane-pair kernels: long enough that an unsynchronised read would see stale bytes
    s1, m1, _, f1 = synth.make_batch('sp1', H(s['proof']), n, 0x5A4B56A1, pool=4, mutate_every=5)
    vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (n, 1))
    pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (n, 1)); pv[f1, -1] ^= 1
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (np.ones(n, dtype=np.uint8), s1, vk, pv)]
    d_st = torch.zeros(n, dtype=torch.uint8, device=dev)            # zero-filled on purpose: stale = accept
    torch.cuda.synchronize()
    v = zkv.MixedVerifier(H(r['control_root']), H(r['bn254_control_id']))
    side, other = torch.cuda.Stream(), torch.cuda.Stream()      # two non-blocking streams: work on `other` is NOT ordered after `side`
    v.verify_batch_dev(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 96, 96, d_st.data_ptr(), 0, side.cuda_stream)
    v.synchronize()
    with torch.cuda.stream(other):                               # this copy only waits for `other`: it sees what is in HBM right now
        host = d_st.cpu().numpy()
    assert ((host == 0) == ~m1).all() and 0 < (host != 0).sum() < n
    ms = v.last_stage_ms()
    assert len(ms) == 5 and ms[3] > 0 and ms[4] > 0 and all(x >= 0 for x in ms)
    # then an all-RISC-Zero batch on the same verifier: the stage times now come from the other child alone
    k0 = 3000
    s0, m0, _, f0 = synth.make_batch('risc0', H(r['seal']), k0, 0x5A4B56A2, pool=4, mutate_every=7)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (k0, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (k0, 1)); jds[f0, 0] ^= 1
    e = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (np.zeros(k0, dtype=np.uint8), s0, ids, jds)]
    e_st = torch.zeros(k0, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    v.verify_batch_dev(k0, e[0].data_ptr(), e[1].data_ptr(), e[2].data_ptr(), e[3].data_ptr(), 32, 0, e_st.data_ptr(), 0, side.cuda_stream)
    v.synchronize()
    with torch.cuda.stream(other):
        host0 = e_st.cpu().numpy()
    assert ((host0 == 0) == ~m0).all()
    ms0 = v.last_stage_ms()
    assert ms0[3] < ms[3]                                  # 3,000 proofs (16-lane kernels) against 20,000: not the sum of both children
    v.close()


def _job_worker(rank, world, port, n0, n1, q):
    """One rank of the config-4 job on a one-GPU box: collectives over gloo on host tensors, compute on cuda:0."""
    import json
    import torch
    import torch.distributed as dist
    import stylus_zkvm_verifiers_amd as z
    from stylus_zkvm_verifiers_amd import parallel, synth
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'real_proofs.json')))
    r, s = g['risc0'], g['sp1']
    dev, cdev = torch.device('cuda', 0), torch.device('cpu')
    root, mut = None, None
    n = n0 + n1
    if rank == 0:
        s0, m0, _, f0 = synth.make_batch('risc0', H(r['seal']), n0, 0x5A4B5681, pool=4, mutate_every=8)
        s1, m1, _, f1 = synth.make_batch('sp1', H(s['proof']), n1, 0x5A4B5682, pool=4, mutate_every=8)
        ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n0, 1))
        jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n0, 1)); jds[f0, 0] ^= 1
        vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (n1, 1))
        pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (n1, 1)); pv[f1, -1] ^= 1
        vm, seals, a, b, perm = parallel.interleave([(0, s0, ids, jds), (1, s1, vk, pv)], 0x5A4B5603)
        mut = np.concatenate([m0, m1])[perm]
        root = [torch.from_numpy(np.ascontiguousarray(x)) for x in (vm.reshape(-1, 1), seals, a, b)]
    state = {}

    def verify_fn(p, vm_t, seals_t, a_t, b_t):
        if 'v' not in state:
            state['v'] = z.MixedVerifier(p[:32], p[32:], 0)
        m = int(vm_t.numel())
        st = torch.full((m,), 255, dtype=torch.uint8, device=dev)
        state['v'].verify_batch_dev(m, vm_t.data_ptr(), seals_t.data_ptr(), a_t.data_ptr(), b_t.data_ptr(), int(b_t.shape[1]), 96, st.data_ptr(), 0,
                                    torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return st

    params = H(r['control_root']) + H(r['bn254_control_id'])
    out, t = parallel.mixed_step(params if rank == 0 else None, root, n, verify_fn, dev, cdev, sync=torch.cuda.synchronize)
    res = None
    if rank == 0:
        res = (out.cpu().numpy().tobytes(), mut.tobytes())
    q.put((rank, res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.gpu
def test_config4_job_world1_and_world2_agree_with_construction(zkv):
    """The config-4 job (bench.py --workload mixed: parameters broadcast, rows scattered, shards verified through the per-proof-VM
    entry point, statuses gathered in the original order) run with ONE rank and with TWO ranks (both on this box's GPU, collectives
    over gloo): identical status bytes, accept <=> not mutated."""
    ctx = mp.get_context('spawn')
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_job_worker, args=(rk, world, port, 1200, 1000, q)) for rk in range(world)]
        for p in procs:
            p.start()
        got = dict(q.get(timeout=600) for _ in procs)
        for p in procs:
            p.join(timeout=120)
        res[world] = got[0]
    st1 = np.frombuffer(res[1][0], dtype=np.uint8); mut = np.frombuffer(res[1][1], dtype=bool)
    st2 = np.frombuffer(res[2][0], dtype=np.uint8)
    assert (st1 == st2).all()
    assert ((st1 == 0) == ~mut).all() and (st1 == 0).sum() > 1500


def _config4_through_properties(zkv, mixed, real_proofs, k):
    """k RISC Zero + k SP1 proofs interleaved by the seeded permutation through the job's own step function on one GPU, checked by
    size-independent properties: the batch is a seeded shuffle of copies of a 2^12 + 2^12 base batch, so status[i] must equal
    base_status[source[i]] (permutation equivariance across the demultiplexer and across chunk boundaries), accept <=> not mutated,
    and a second run returns the same bytes (idempotence)."""
    import torch
    from stylus_zkvm_verifiers_amd import parallel, synth
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream().cuda_stream
    r, s = real_proofs['risc0'], real_proofs['sp1']
    nb = 1 << 12
    s0, m0, _, f0 = synth.make_batch('risc0', H(r['seal']), nb, 0x5A4B5691, pool=8, mutate_every=16)
    s1, m1, _, f1 = synth.make_batch('sp1', H(s['proof']), nb, 0x5A4B5692, pool=8, mutate_every=16)
    ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (nb, 1))
    jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (nb, 1)); jds[f0, 0] ^= 1
    vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (nb, 1))
    pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (nb, 1)); pv[f1, -1] ^= 1
    src0 = np.random.default_rng(0x5A4B5693).permutation(k) % nb
    src1 = np.random.default_rng(0x5A4B5694).permutation(k) % nb
    vm, seals, a, b, perm = parallel.interleave([(0, s0[src0], ids[src0], jds[src0]), (1, s1[src1], vk[src1], pv[src1])], 0x5A4B5603)
    mut = np.concatenate([m0[src0], m1[src1]])[perm]
    n = 2 * k
    root = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (vm.reshape(-1, 1), seals, a, b)]
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)

    def verify_fn(p, vm_t, seals_t, a_t, b_t):
        mixed.verify_batch_dev(int(vm_t.numel()), vm_t.data_ptr(), seals_t.data_ptr(), a_t.data_ptr(), b_t.data_ptr(), 96, 96, d_st.data_ptr(), 0, stream)
        return d_st

    params = H(r['control_root']) + H(r['bn254_control_id'])
    out, t = parallel.mixed_step(params, root, n, verify_fn, dev, dev, sync=torch.cuda.synchronize)
    st = out.cpu().numpy().copy()
    assert ((st == 0) == ~mut).all()
    # base statuses from the single-VM device entry points on the 2^12 base batches
    r0v = zkv.RiscZeroVerifier(); r0v.initialize(H(r['control_root']), H(r['bn254_control_id']))
    spv = zkv.Sp1Verifier()
    bs0 = torch.full((nb,), 255, dtype=torch.uint8, device=dev); bs1 = torch.full((nb,), 255, dtype=torch.uint8, device=dev)
    t0 = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (s0, ids, jds)]
    t1 = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (vk, pv, s1)]
    r0v.verify_batch_dev(nb, t0[0].data_ptr(), t0[1].data_ptr(), t0[2].data_ptr(), bs0.data_ptr(), 0, stream)
    spv.verify_batch_dev(nb, t1[0].data_ptr(), t1[1].data_ptr(), 96, t1[2].data_ptr(), bs1.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    base = np.concatenate([bs0.cpu().numpy()[src0], bs1.cpu().numpy()[src1]])[perm]
    assert (st == base).all()
    d_st.fill_(255)
    out2, _ = parallel.mixed_step(params, root, n, verify_fn, dev, dev, sync=torch.cuda.synchronize)
    assert (out2.cpu().numpy() == st).all()
    r0v.close(); spv.close()


@pytest.mark.gpu
def test_config4_per_gpu_share_2p19_through_properties(zkv, mixed, real_proofs):
    """The per-GPU share of BASELINE config 4 (2^18 RISC Zero + 2^18 SP1) on one GPU."""
    _config4_through_properties(zkv, mixed, real_proofs, 1 << 18)


@pytest.mark.gpu
def test_config4_full_2p22_batch_on_one_gpu(zkv, mixed, real_proofs):
    """BASELINE configs[3] at its FULL size -- 2^21 RISC Zero + 2^21 SP1 proofs, interleaved by the seeded permutation -- through
    parallel.mixed_step on one MI355X: the demultiplexer splits it into two 2^21-proof sub-batches, each of which takes two 2^20-proof
    chunks of the stage pipeline (four chunks in all); what eight GPUs would each see an eighth of."""
    from stylus_zkvm_verifiers_amd import _lib
    assert _lib.lib().zkv_chunk_capacity() <= 1 << 20             # so that the sub-batches really cross chunk boundaries
    _config4_through_properties(zkv, mixed, real_proofs, 1 << 21)


def _nccl_world1_worker(port, q):
    """RCCL on this box at all: process group over backend "nccl" with one rank, the collectives bench.py's N > 1 path uses that are
    defined for a single rank (broadcast, all_reduce, barrier), then one config-4 step under the initialised group."""
    import json
    import torch
    import torch.distributed as dist
    import stylus_zkvm_verifiers_amd as z
    from stylus_zkvm_verifiers_amd import parallel, synth
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    rank, local_rank, world = parallel.init_distributed('nccl')           # what bench.py does (backend None picks nccl on a GPU box)
    assert (rank, world) == (0, 1)
    if not dist.is_initialized():                                        # init_distributed skips a one-rank world: initialise it here
        torch.cuda.set_device(0)
        dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    dev = torch.device('cuda', 0)
    t = torch.arange(8, dtype=torch.int64, device=dev)
    dist.broadcast(t, src=0); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
    blob = parallel.broadcast_context(parallel.pack_context(parallel.CTX_SP1), dev)
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'real_proofs.json')))
    r, s = g['risc0'], g['sp1']
    n1 = 600
    s1, m1, _, f1 = synth.make_batch('sp1', H(s['proof']), n1, 0x5A4B56D1, pool=4, mutate_every=6)
    vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (n1, 1))
    pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (n1, 1)); pv[f1, -1] ^= 1
    root = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (np.ones((n1, 1), dtype=np.uint8), s1, vk, pv)]
    mv = z.MixedVerifier(H(r['control_root']), H(r['bn254_control_id']), 0)
    st = torch.full((n1,), 255, dtype=torch.uint8, device=dev)

    def verify_fn(p, vm_t, seals_t, a_t, b_t):
        mv.verify_batch_dev(int(vm_t.numel()), vm_t.data_ptr(), seals_t.data_ptr(), a_t.data_ptr(), b_t.data_ptr(), 96, 96, st.data_ptr(), 0,
                            torch.cuda.current_stream().cuda_stream)
        return st
    out, _ = parallel.mixed_step(H(r['control_root']) + H(r['bn254_control_id']), root, n1, verify_fn, dev, dev, sync=torch.cuda.synchronize)
    ok = bool(((out.cpu().numpy() == 0) == ~m1).all()) and parallel.unpack_context(blob)['kind'] == parallel.CTX_SP1 and t.tolist() == list(range(8))
    q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_backend_initialises_and_a_step_runs_under_it(zkv):
    """The N > 1 path of bench.py has never met more than one real rank (one-GPU boxes); this at least proves that backend "nccl"
    (= RCCL) initialises on the box with the device_id binding bench.py uses, that its collectives run, and that a config-4 step runs
    inside an initialised process group."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(_free_port(), q))
    p.start()
    ok = q.get(timeout=600)
    p.join(timeout=120)
    assert ok is True and p.exitcode == 0


@pytest.mark.gpu
def test_bench_two_ranks_rehearsed_on_one_gpu_prints_one_json_line():
    """`python bench.py --gpus 2 --rehearse-single-gpu` end to end, as the driver would start the N > 1 bench (a fresh process that
    launches its ranks itself; both ranks on cuda:0, collectives over gloo): exactly ONE JSON line on stdout, n_gpus = 2, config 4
    (mixed), accept <=> construction on every rank.  (No 8-GPU node is available to this build: this is the whole multi-rank path short
    of RCCL with more than one real device.)"""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--rehearse-single-gpu', '--steps', '1', '--warmup', '1', '--proofs', '8192',
                        '--no-cpu-baseline', '--no-mulmod'], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 1 and j['config']['workload'] == 'mixed' and j['scaling'] == 'weak'
    assert j['parity']['accept_reject_matches_construction'] is True
    assert j['value'] > 0 and j['config']['global_batch'] == 2 * 8192
