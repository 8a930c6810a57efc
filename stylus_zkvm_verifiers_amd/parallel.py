"""Multi-GPU sharding of a proof batch: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

Proofs are independent units: the batch shards by contiguous ranges after a seeded permutation
(`seeded_permutation`, `interleave`: RISC Zero and SP1 proofs end up evenly mixed in every shard, and so do the rejects
that leave the pipeline early), the verify path itself contains NO collective.  The only communication is distribution and collection around it:
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


def scatter_rows_multi(fulls, n_total, widths, device, src=0):
    """scatter_rows for several row arrays at once: all sends (root) / receives (peers) are posted as ONE batch
    (`dist.batch_isend_irecv` = one ncclGroupStart/End on RCCL), so the root drives its seven xGMI links concurrently instead of
    peer after peer.  fulls: list of uint8 [n_total, w] tensors on the root (ignored elsewhere).  Returns this rank's shards."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_bounds(n_total, world, rank)
    mine = [torch.empty((hi - lo, w), dtype=torch.uint8, device=device) for w in widths]
    ops = []
    if rank == src:
        for peer in range(world):
            plo, phi = shard_bounds(n_total, world, peer)
            for k, w in enumerate(widths):
                if peer == src:
                    mine[k].copy_(fulls[k][plo:phi])
                elif phi > plo:
                    ops.append(dist.P2POp(dist.isend, fulls[k][plo:phi].contiguous(), peer))
    elif hi > lo:
        for k in range(len(widths)):
            ops.append(dist.P2POp(dist.irecv, mine[k], src))
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
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


def gather_rows(local, n_total, device, dst=0):
    """Collect contiguous row shards (uint8 [k, w], split as shard_bounds) on rank `dst`; returns [n_total, w] there, None elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    w = local.shape[1]
    if rank == dst:
        out = torch.empty((n_total, w), dtype=torch.uint8, device=device)
        for peer in range(world):
            plo, phi = shard_bounds(n_total, world, peer)
            if peer == dst:
                out[plo:phi].copy_(local)
            elif phi > plo:
                buf = torch.empty((phi - plo, w), dtype=torch.uint8, device=device)
                dist.recv(buf, src=peer)
                out[plo:phi].copy_(buf)
        return out
    if local.numel():
        dist.send(local.contiguous(), dst=dst)
    return None


# ---------------------------------------------------------------- BASELINE config 4: mixed RISC Zero + SP1 batch on the root
def splitmix64(x):
    """Vectorised SplitMix64 finaliser on uint64 arrays (counter-based: element i depends on x[i] only)."""
    import numpy as np
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def seeded_permutation(n, seed):
    """Permutation of range(n) that depends on (n, seed) only: stable argsort of the SplitMix64 keys of seed + i."""
    import numpy as np
    with np.errstate(over='ignore'):
        keys = splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF))
    return np.argsort(keys, kind='stable')


def interleave(parts, seed):
    """parts: list of (vm_tag, seals uint8[k,260], in_a uint8[k,32], in_b uint8[k,w]) -- one entry per VM, any widths w >= 32.
    Returns (vm uint8[n], seals, in_a, in_b uint8[n, max w] zero-padded, perm) with row i = row perm[i] of the VM-major
    concatenation: the mixed batch of SURVEY.md 8(d) config 4 ("interleaved by seeded permutation")."""
    import numpy as np
    width = max(p[3].shape[1] for p in parts)
    vm = np.concatenate([np.full(len(p[1]), p[0], dtype=np.uint8) for p in parts])
    seals = np.concatenate([p[1] for p in parts])
    in_a = np.concatenate([p[2] for p in parts])
    in_b = np.zeros((len(vm), width), dtype=np.uint8)
    at = 0
    for p in parts:
        in_b[at:at + len(p[1]), :p[3].shape[1]] = p[3]
        at += len(p[1])
    perm = seeded_permutation(len(vm), seed)
    return vm[perm], seals[perm], in_a[perm], in_b[perm], perm


def mixed_step(params, root, n_total, verify_fn, dev, cdev, sync=lambda: None):
    """One pass of config 4: rank 0 holds the mixed batch `root` = (vm [n], seals [n,260], in_a [n,32], in_b [n,w]) as uint8
    tensors on `cdev`; the 64 bytes of verifier parameters are broadcast, the four row arrays are scattered (contiguous
    shards, one direct link per peer), every rank verifies its shard with `verify_fn(params, vm, seals, in_a, in_b)` (tensors on
    `dev`; returns the shard's status tensor) and the status bytes are gathered in the original order.
    Returns (status on rank 0 / None elsewhere, {'distribute','verify','collect'} seconds on this rank)."""
    import time
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    t0 = time.perf_counter()
    widths = (1, 260, 32, None)
    if world > 1:
        params = broadcast_bytes(params, 64, cdev)
        w = torch.zeros(1, dtype=torch.int64, device=cdev)
        if rank == 0:
            w[0] = root[3].shape[1]
        dist.broadcast(w, src=0)
        wids = [int(w.item()) if x is None else x for x in widths]
        fulls = [root[k].reshape(n_total, wids[k]) for k in range(4)] if rank == 0 else None
        shard = scatter_rows_multi(fulls, n_total, wids, cdev)
    else:
        shard = [root[0].reshape(n_total, 1), root[1], root[2], root[3]]
    local = [t if t.device == dev else t.to(dev) for t in shard]
    sync()
    t1 = time.perf_counter()
    st = verify_fn(params, local[0].reshape(-1), local[1], local[2], local[3])
    sync()
    t2 = time.perf_counter()
    if world > 1:
        out = gather_status(st if st.device == cdev else st.to(cdev), n_total, cdev)
    else:
        out = st
    sync()
    t3 = time.perf_counter()
    return out, {'distribute': t1 - t0, 'verify': t2 - t1, 'collect': t3 - t2}
