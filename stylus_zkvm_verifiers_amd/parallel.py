"""Multi-GPU sharding of a proof batch: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

Proofs are independent units: the batch shards by contiguous ranges after a seeded permutation
(`seeded_permutation`, `interleave`: RISC Zero and SP1 proofs end up evenly mixed in every shard, and so do the rejects
that leave the pipeline early), the verify path itself contains NO collective.  The only communication is distribution and collection around it:
  * broadcast of the CONTEXT BLOB from rank 0 (`pack_context` / `broadcast_context` / `make_verifier`): the 64 bytes of RISC Zero
    verifier parameters (control_root, bn254_control_id) -- the RISC Zero / SP1 keys themselves are compiled-in constants, as in the
    reference -- or, for the caller-keyed verifiers, the key itself: the `VerificationKey` words of a generic Groth16 context
    (common/groth16.rs:23-31, common/types.rs:17-23) or the PLONK verifying key + verifier hash.  Every rank builds the same
    verifier (selector, device tables) from the blob;
  * scatter of seal / input rows from rank 0 (point-to-point sends, one direct xGMI link per peer), in PIECES: the sends of piece
    k + 1 are posted before piece k is verified, so only the first, small piece's transfer is exposed;
  * gather of one status byte per proof (all receives posted as one group).
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
    """Collect one status byte per proof on rank `dst` (returns the full uint8 [n_total] there, None elsewhere).  The root posts
    all its receives as ONE group, straight into the slices of the output (contiguous shards): seven links drain concurrently."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == dst:
        out = torch.empty(n_total, dtype=torch.uint8, device=device)
        ops = []
        for peer in range(world):
            plo, phi = shard_bounds(n_total, world, peer)
            if peer == dst:
                out[plo:phi].copy_(local_status)
            elif phi > plo:
                ops.append(dist.P2POp(dist.irecv, out[plo:phi], peer))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        return out
    if local_status.numel():
        for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, local_status.contiguous(), dst)]):
            r.wait()
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


# ---------------------------------------------------------------- context blobs: what every rank needs to build the same verifier
CTX_RISC0, CTX_SP1, CTX_GROTH16, CTX_PLONK, CTX_MIXED = 0, 1, 3, 6, 5          # = the ZKV_VM_* tags of include/zkv.h


def pack_context(kind, control_root=b'', bn254_control_id=b'', vk=b'', n_ic=0, vm_type=1, verifier_hash=b''):
    """Serialise the parameters of one verifier: kind (1 byte) | n_ic | vm_type | 3 length-prefixed byte strings.
    RISC Zero / mixed: control_root + bn254_control_id (`initialize`, risc0/verifier.rs:58-76); SP1: nothing; generic Groth16: the
    `VerificationKey` words (448 + 64 n_ic bytes), n_ic and the VMType; PLONK: the verifying key and the verifier hash."""
    a = bytes(control_root) + bytes(bn254_control_id)
    if kind in (CTX_RISC0, CTX_MIXED) and len(a) != 64:
        raise ValueError('control_root and bn254_control_id are 32 bytes each')
    if kind == CTX_GROTH16 and len(vk) != 448 + 64 * n_ic:
        raise ValueError('a generic key is 448 + 64 * n_ic bytes')
    if kind == CTX_PLONK and len(verifier_hash) != 32:
        raise ValueError('verifier_hash is 32 bytes')
    parts = [a, bytes(vk), bytes(verifier_hash)]
    return bytes([kind, n_ic & 255, vm_type & 255]) + b''.join(len(x).to_bytes(4, 'big') + x for x in parts)


def unpack_context(blob):
    kind, n_ic, vm_type = blob[0], blob[1], blob[2]
    at, parts = 3, []
    for _ in range(3):
        ln = int.from_bytes(blob[at:at + 4], 'big'); at += 4
        parts.append(bytes(blob[at:at + ln])); at += ln
    if at != len(blob):
        raise ValueError('malformed context blob')
    return {'kind': kind, 'n_ic': n_ic, 'vm_type': vm_type, 'control_root': parts[0][:32], 'bn254_control_id': parts[0][32:],
            'vk': parts[1], 'verifier_hash': parts[2]}


def broadcast_context(blob, device, src=0):
    """Broadcast a context blob of any length from `src` (its length first); returns the bytes on every rank."""
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if dist.get_rank() == src:
        n[0] = len(blob)
    dist.broadcast(n, src=src)
    return broadcast_bytes(blob, int(n.item()), device, src=src)


def make_verifier(blob, device_index=0, aggregate_sub_batch=0):
    """The verifier a context blob describes, bound to HIP device `device_index` (the same object on every rank).  aggregate_sub_batch =
    16 / 32 / 64 / 128 / 256 switches the opt-in aggregate check on (include/zkv.h): every rank draws its OWN secret from its operating system -- the
    coefficients are never part of the broadcast blob."""
    from . import Groth16Verifier, MixedVerifier, RiscZeroVerifier, Sp1PlonkVerifier, Sp1Verifier
    c = unpack_context(blob)
    if c['kind'] == CTX_RISC0:
        v = RiscZeroVerifier(device_index); v.initialize(c['control_root'], c['bn254_control_id'])
    elif c['kind'] == CTX_SP1:
        v = Sp1Verifier(device_index)
    elif c['kind'] == CTX_MIXED:
        v = MixedVerifier(c['control_root'], c['bn254_control_id'], device_index)
    elif c['kind'] == CTX_GROTH16:
        v = Groth16Verifier(c['vk'], c['n_ic'], c['vm_type'], device_index)
    elif c['kind'] == CTX_PLONK:
        v = Sp1PlonkVerifier(c['vk'], c['verifier_hash'], device_index)
    else:
        raise ValueError('unknown context kind %d' % c['kind'])
    if aggregate_sub_batch:
        v.set_aggregate_check(True, seed=None, sub_batch=aggregate_sub_batch)
    return v


# ---------------------------------------------------------------- one sharded pass: broadcast context, scatter rows in pieces, verify, gather
def _pieces_of(lo, hi, first_piece):
    """Split a shard [lo, hi) into (at most) two contiguous pieces: a first piece whose transfer is exposed, then the rest, whose
    transfer runs behind the first piece's kernels.  first_piece = None picks HALF the shard for shards of at least 2^17 rows and one
    piece below: the stage kernels lose efficiency on small launches (2^15-proof sub-batches of a mixed piece run at one wavefront per
    SIMD), which costs more than the transfer a smaller first piece would hide (multi-GPU)."""
    m = hi - lo
    if first_piece is None:
        first_piece = m // 2 if m >= (1 << 17) else 0
        if first_piece:
            return [(lo, lo + first_piece), (lo + first_piece, hi)]
    if first_piece <= 0 or m <= 2 * first_piece:
        return [(lo, hi)]
    return [(lo, lo + first_piece), (lo + first_piece, hi)]


def _post_piece(fulls, n_total, widths, device, j, first_piece, src):
    """Posts (as one group) the sends / receives of piece j of every rank's shard; returns (this rank's piece tensors, requests)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    ops, mine = [], None
    if rank == src:
        for peer in range(world):
            pcs = _pieces_of(*shard_bounds(n_total, world, peer), first_piece)
            if j >= len(pcs):
                continue
            a, b = pcs[j]
            for k in range(len(widths)):
                if peer == src:
                    mine = (mine or []) + [fulls[k][a:b]]
                elif b > a:
                    ops.append(dist.P2POp(dist.isend, fulls[k][a:b].contiguous(), peer))
    else:
        pcs = _pieces_of(*shard_bounds(n_total, world, rank), first_piece)
        if j < len(pcs) and pcs[j][1] > pcs[j][0]:
            mine = [torch.empty((pcs[j][1] - pcs[j][0], w), dtype=torch.uint8, device=device) for w in widths]
            ops = [dist.P2POp(dist.irecv, t, src) for t in mine]
    return mine, (dist.batch_isend_irecv(ops) if ops else [])


def sharded_step(ctx_blob, root, n_total, verify_fn, dev, cdev, sync=lambda: None, first_piece=None):
    """One data-parallel pass over a batch held by rank 0.  `ctx_blob` (rank 0; None elsewhere) is broadcast; `root` = list of uint8
    row tensors [n_total, w_k] on `cdev` (rank 0; None elsewhere) is scattered in contiguous shards -- each shard in two pieces, the
    second piece's transfer posted before the first piece is verified --; every rank calls `verify_fn(ctx_blob, *row_tensors_on_dev)`
    per piece (it returns that piece's uint8 status tensor) and the status bytes are gathered on rank 0 in the original order.
    Returns (status on rank 0 / None elsewhere, {'distribute' (exposed), 'verify', 'collect'} seconds on this rank)."""
    import time
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    t0 = time.perf_counter()
    if world > 1:
        blob = broadcast_context(ctx_blob, cdev)
        w = torch.zeros(16, dtype=torch.int64, device=cdev)
        if rank == 0:
            w[0] = len(root)
            for k, x in enumerate(root):
                w[1 + k] = x.shape[1]
        dist.broadcast(w, src=0)
        widths = [int(v) for v in w[1:1 + int(w[0].item())].tolist()]
        fulls = [root[k].reshape(n_total, widths[k]) for k in range(len(widths))] if rank == 0 else None
        lo, hi = shard_bounds(n_total, world, rank)
        n_pieces = max(len(_pieces_of(*shard_bounds(n_total, world, p), first_piece)) for p in range(world))
        status = torch.empty(hi - lo, dtype=torch.uint8, device=dev)
        cur, reqs = _post_piece(fulls, n_total, widths, cdev, 0, first_piece, 0)
        t1 = None
        at = 0
        for j in range(n_pieces):
            for r in reqs:
                r.wait()
            nxt, nreqs = _post_piece(fulls, n_total, widths, cdev, j + 1, first_piece, 0) if j + 1 < n_pieces else (None, [])
            if t1 is None:
                sync()
                t1 = time.perf_counter()
            if cur is not None and cur[0].shape[0]:
                local = [t if t.device == dev else t.to(dev) for t in cur]
                st = verify_fn(blob, *local)
                status[at:at + st.numel()].copy_(st)
                at += st.numel()
            cur, reqs = nxt, nreqs
        sync()
        t2 = time.perf_counter()
        out = gather_status(status if status.device == cdev else status.to(cdev), n_total, cdev)
    else:
        local = [t if t.device == dev else t.to(dev) for t in root]
        sync()
        t1 = time.perf_counter()
        out = verify_fn(ctx_blob, *local)
        sync()
        t2 = time.perf_counter()
    sync()
    t3 = time.perf_counter()
    return out, {'distribute': t1 - t0, 'verify': t2 - t1, 'collect': t3 - t2}


def mixed_step(params, root, n_total, verify_fn, dev, cdev, sync=lambda: None, first_piece=None):
    """One pass of config 4: rank 0 holds the mixed batch `root` = (vm [n], seals [n,260], in_a [n,32], in_b [n,w]) as uint8
    tensors on `cdev`; the 64 bytes of verifier parameters travel as a CTX_MIXED context blob, the four row arrays are scattered
    (contiguous shards, one direct link per peer, second piece behind the first piece's kernels), every rank verifies its shard with
    `verify_fn(params, vm, seals, in_a, in_b)` (tensors on `dev`; returns the status tensor of what it was given) and the status bytes
    are gathered in the original order.  Returns (status on rank 0 / None elsewhere, {'distribute','verify','collect'} seconds)."""
    blob = pack_context(CTX_MIXED, params[:32], params[32:]) if params is not None else None
    rows = [root[0].reshape(n_total, 1), root[1], root[2], root[3]] if root is not None else None

    def fn(b, vm, seals, a, bb):
        c = unpack_context(b)
        return verify_fn(c['control_root'] + c['bn254_control_id'], vm.reshape(-1), seals, a, bb)
    return sharded_step(blob, rows, n_total, fn, dev, cdev, sync=sync, first_piece=first_piece)
