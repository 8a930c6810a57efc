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
