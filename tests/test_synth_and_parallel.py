"""Synthetic batch generator (checked by the independent CPU oracle) and the multi-GPU sharding helpers on the gloo
backend with world_size 2 (the RCCL path runs the same code on HBM tensors)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol
from stylus_zkvm_verifiers_amd import parallel, synth

H = bytes.fromhex


def test_rerandomised_batches_are_valid_and_mutations_are_not(real_proofs):
    r = real_proofs['risc0']
    n = 96
    seals, mut, mclass, flip = synth.make_batch('risc0', H(r['seal']), n, 0x5A4B5601, pool=4, mutate_every=6)
    assert len({s.tobytes() for s in seals}) == n                     # all distinct
    v = ol.Risc0Oracle(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
    jd = H(r['journal_digest'])
    jds = [bytes([jd[0] ^ 1]) + jd[1:] if f else jd for f in flip]
    st, _ = v.verify_batch([s.tobytes() for s in seals], [H(r['image_id'])] * n, jds, threads=4)
    assert (st[~mut] == 0).all() and (st[mut] != 0).all()
    assert set(mclass[mut]) <= set(range(len(synth.MUTATION_CLASSES))) and (mclass[~mut] == -1).all()
    s = real_proofs['sp1']
    seals, mut, mclass, flip = synth.make_batch('sp1', H(s['proof']), 48, 0x5A4B5602, pool=4, mutate_every=6)
    pv = H(s['public_values'])
    pvs = [pv[:-1] + bytes([pv[-1] ^ 1]) if f else pv for f in flip]
    st, _ = ol.sp1_verify_batch([H(s['vkey'])] * 48, pvs, [x.tobytes() for x in seals], threads=4)
    assert (st[~mut] == 0).all() and (st[mut] != 0).all()


def test_generator_is_seeded(real_proofs):
    r = real_proofs['risc0']
    a = synth.make_batch('risc0', H(r['seal']), 8, 42, pool=2)[0]
    b = synth.make_batch('risc0', H(r['seal']), 8, 42, pool=2)[0]
    c = synth.make_batch('risc0', H(r['seal']), 8, 43, pool=2)[0]
    assert (a == b).all() and not (a == c).all()


def test_shard_bounds_partition():
    for n in (0, 1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, out_q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cpu')
    params = parallel.broadcast_bytes(bytes(range(64)) if rank == 0 else b'', 64, dev)
    full = torch.arange(n * 260, dtype=torch.int64).remainder(251).to(torch.uint8).reshape(n, 260) if rank == 0 else None
    mine = parallel.scatter_rows(full, n, 260, dev)
    lo, hi = parallel.shard_bounds(n, world, rank)
    expect = torch.arange(n * 260, dtype=torch.int64).remainder(251).to(torch.uint8).reshape(n, 260)[lo:hi]
    ok = params == bytes(range(64)) and mine.shape[0] == hi - lo and bool((mine == expect).all())
    status = (torch.arange(lo, hi) % 6).to(torch.uint8)            # stand-in for per-proof status bytes
    allst = parallel.gather_status(status, n, dev)
    if rank == 0:
        ok = ok and bool((allst == (torch.arange(n) % 6).to(torch.uint8)).all())
    out_q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n', [10, 7])
def test_broadcast_scatter_gather_world2(n):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_context_blob_round_trip():
    """pack_context / unpack_context for every verifier kind; wrong shapes are refused."""
    cr, cid = bytes(range(32)), bytes(range(32, 64))
    for kind in (parallel.CTX_RISC0, parallel.CTX_MIXED):
        c = parallel.unpack_context(parallel.pack_context(kind, cr, cid))
        assert (c['kind'], c['control_root'], c['bn254_control_id']) == (kind, cr, cid)
    assert parallel.unpack_context(parallel.pack_context(parallel.CTX_SP1))['kind'] == parallel.CTX_SP1
    vk = bytes((7 * i) & 255 for i in range(448 + 64 * 3))
    c = parallel.unpack_context(parallel.pack_context(parallel.CTX_GROTH16, vk=vk, n_ic=3, vm_type=0))
    assert (c['kind'], c['vk'], c['n_ic'], c['vm_type']) == (parallel.CTX_GROTH16, vk, 3, 0)
    pk = bytes(1056)
    c = parallel.unpack_context(parallel.pack_context(parallel.CTX_PLONK, vk=pk, verifier_hash=bytes(range(32))))
    assert (c['kind'], c['vk'], c['verifier_hash']) == (parallel.CTX_PLONK, pk, bytes(range(32)))
    with pytest.raises(ValueError):
        parallel.pack_context(parallel.CTX_GROTH16, vk=vk, n_ic=2)
    with pytest.raises(ValueError):
        parallel.pack_context(parallel.CTX_RISC0, cr, cid[:31])
    with pytest.raises(ValueError):
        parallel.unpack_context(parallel.pack_context(parallel.CTX_SP1) + b'x')


def _sharded_worker(rank, world, port, n, first_piece, out_q):
    """A generic-Groth16-shaped job (caller-supplied key in the context blob, 256-byte proofs + k x 32-byte signals) through
    sharded_step; the stand-in verifier folds the KEY BYTES it received into every status, so a rank that did not get the
    broadcast key cannot produce the expected bytes."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cpu')
    vk = bytes((11 * i + 3) & 255 for i in range(448 + 64 * 3))
    blob = parallel.pack_context(parallel.CTX_GROTH16, vk=vk, n_ic=3, vm_type=0) if rank == 0 else None
    root = None
    if rank == 0:
        g = torch.Generator().manual_seed(9)
        root = [torch.randint(0, 256, (n, 256), dtype=torch.uint8, generator=g), torch.randint(0, 256, (n, 64), dtype=torch.uint8, generator=g)]
    calls = []

    def fold(b, proofs, signals):
        c = parallel.unpack_context(b)
        key = sum(c['vk']) + 1000 * c['n_ic'] + 7 * c['vm_type'] + c['kind']
        return ((proofs.to(torch.int64).sum(1) + 3 * signals.to(torch.int64).sum(1) + key) % 251).to(torch.uint8)

    def verify_fn(b, proofs, signals):
        calls.append(int(proofs.shape[0]))
        return fold(b, proofs, signals)

    out, t = parallel.sharded_step(blob, root, n, verify_fn, dev, dev, first_piece=first_piece)
    lo, hi = parallel.shard_bounds(n, world, rank)
    want_calls = [b - a for a, b in parallel._pieces_of(lo, hi, first_piece) if b > a]
    ok = calls == want_calls and set(t) == {'distribute', 'verify', 'collect'}
    if rank == 0:
        ok = ok and out is not None and bool((out == fold(blob, root[0], root[1])).all())
    else:
        ok = ok and out is None
    out_q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n,first_piece', [(2, 50, 1 << 16), (2, 50, 4), (3, 41, 3), (2, 1, 4), (2, 50, None)])
def test_sharded_step_broadcasts_caller_keys_and_pipelines_the_scatter(world, n, first_piece):
    """sharded_step over gloo: the caller-supplied key (generic Groth16 blob) reaches every rank, shards arrive in one piece or in
    two (first_piece smaller than half a shard: the second piece's transfer is posted before the first is verified), statuses come
    back in the original order; also three ranks and a batch smaller than the world."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, n, first_piece, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]
