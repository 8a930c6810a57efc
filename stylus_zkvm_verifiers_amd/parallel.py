"""Multi-GPU sharding of a proof batch: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

Proofs are independent units: the batch shards by contiguous ranges after a seeded permutation, the
verify path itself contains NO collective.  The only communication is distribution and collection around it:
  * broadcast of the 64-byte verifier parameters (control_root, bn254_control_id) from rank 0 -- every rank derives
    the same selector and device tables from them (the VK itself is a compiled-in constant, as in the reference);
  * scatter of seal / input rows from rank 0 (point-to-point sends, one direct xGMI link per peer);
  * gather of one status byte per proof.
All three work on CPU tensors with the gloo backend (tests) and on HBM tensors with RCCL.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def init_distributed(backend=None):
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            kw['device_id'] = torch.device('cuda', local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_bounds(n, world, rank):
    """Contiguous range of rank `rank` when n units are split over `world` ranks (remainder to the low ranks)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_bytes(data, nbytes, device, src=0):
    """Broadcast a small byte string (verifier parameters) from `src`; returns bytes on every rank."""
    t = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    if dist.get_rank() == src:
        t.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
    dist.broadcast(t, src=src)
    return bytes(t.cpu().numpy().tobytes())


def scatter_rows(full, n_total, row_bytes, device, src=0):
    """Rank `src` holds `full` (uint8 [n_total, row_bytes] on `device`); every rank receives its contiguous shard.

    Implemented as point-to-point sends from the root (RCCL has no native scatter; on the xGMI full mesh each
    peer is reached over its own direct link, so the scatter is per-link bound and needs no ring)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_bounds(n_total, world, rank)
    mine = torch.empty((hi - lo, row_bytes), dtype=torch.uint8, device=device)
    if rank == src:
        reqs = []
        for peer in range(world):
            plo, phi = shard_bounds(n_total, world, peer)
            if peer == src:
                mine.copy_(full[plo:phi])
            elif phi > plo:
                reqs.append(dist.isend(full[plo:phi].contiguous(), dst=peer))
        for r in reqs:
            r.wait()
    elif hi > lo:
        dist.recv(mine, src=src)
    return mine


def gather_status(local_status, n_total, device, dst=0):
    """Collect one status byte per proof on rank `dst` (returns the full uint8 [n_total] there, None elsewhere)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == dst:
        out = torch.empty(n_total, dtype=torch.uint8, device=device)
        for peer in range(world):
            plo, phi = shard_bounds(n_total, world, peer)
            if peer == dst:
                out[plo:phi].copy_(local_status)
            elif phi > plo:
                buf = torch.empty(phi - plo, dtype=torch.uint8, device=device)
                dist.recv(buf, src=peer)
                out[plo:phi].copy_(buf)
        return out
    if local_status.numel():
        dist.send(local_status.contiguous(), dst=dst)
    return None
