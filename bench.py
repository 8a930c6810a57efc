#!/usr/bin/env python3
"""Headline benchmark: Groth16 proofs verified / s (BN254) on N x MI355X, beside the CPU oracle on the host cores.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the verify path (prep -> vk_x MSM -> G2 subgroup check -> Miller loop -> final
exponentiation) over one synthetic batch whose seals and public inputs are already resident in HBM.
Workloads (BASELINE.json configs): risc0_2p16 (default, configs[1]), sp1_2p20 (configs[2]), mixed (configs[3],
2^19 proofs per GPU, half RISC Zero half SP1).  Weak scaling: the per-GPU batch is fixed, every rank verifies its
own seeded shard, no collective in the data path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_PROOF = {'risc0': 260 + 32 + 32 + 1, 'sp1': 260 + 32 + 96 + 1}
STAGES = ['prep', 'msm', 'g2chk', 'miller', 'finalexp']
SEEDS = {'risc0_2p16': 0x5A4B5601, 'sp1_2p20': 0x5A4B5602, 'mixed': 0x5A4B5603}


def golden():
    with open(os.path.join(ROOT, 'tests', 'golden', 'real_proofs.json')) as f:
        return json.load(f)


class Shard:
    """One VM's device-resident part of a rank's batch."""

    def __init__(self, vm, n, seed, dev, g, mutate_every):
        from stylus_zkvm_verifiers_amd import RiscZeroVerifier, Sp1Verifier, synth
        H = bytes.fromhex
        self.vm, self.n = vm, n
        base = H(g['risc0']['seal'] if vm == 'risc0' else g['sp1']['proof'])
        seals, mutated, mclass, flip = synth.make_batch(vm, base, n, seed, mutate_every=mutate_every)
        self.h_seals, self.mutated = seals, mutated
        self.expected_ok = int(n - mutated.sum())
        if vm == 'risc0':
            r = g['risc0']
            self.ctx = RiscZeroVerifier(dev.index)
            self.ctx.initialize(H(r['control_root']), H(r['bn254_control_id']))
            ids = np.tile(np.frombuffer(H(r['image_id']), dtype=np.uint8), (n, 1))
            jds = np.tile(np.frombuffer(H(r['journal_digest']), dtype=np.uint8), (n, 1))
            jds[flip, 0] ^= 1
            self.h_a, self.h_b = ids, jds
        else:
            s = g['sp1']
            self.ctx = Sp1Verifier(dev.index)
            vk = np.tile(np.frombuffer(H(s['vkey']), dtype=np.uint8), (n, 1))
            pv = np.tile(np.frombuffer(H(s['public_values']), dtype=np.uint8), (n, 1))
            pv[flip, -1] ^= 1
            self.h_a, self.h_b = vk, pv
        self.d_seals = torch.from_numpy(seals).to(dev)
        self.d_a = torch.from_numpy(np.ascontiguousarray(self.h_a)).to(dev)
        self.d_b = torch.from_numpy(np.ascontiguousarray(self.h_b)).to(dev)
        self.d_status = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        self.stage_ms = np.zeros(5)
        self.stage_samples = 0
        # context set-up (stream, VK tables, workspace) happens here, outside any timed region even with --warmup 0
        self.ctx.reserve(n)
        self.ctx.synchronize()

    def enqueue(self, stream):
        if self.vm == 'risc0':
            self.ctx.verify_batch_dev(self.n, self.d_seals.data_ptr(), self.d_a.data_ptr(), self.d_b.data_ptr(),
                                      self.d_status.data_ptr(), 0, stream)
        else:
            self.ctx.verify_batch_dev(self.n, self.d_a.data_ptr(), self.d_b.data_ptr(), self.h_b.shape[1], self.d_seals.data_ptr(),
                                      self.d_status.data_ptr(), 0, stream)

    def sample_stage_ms(self):
        self.stage_ms += np.array(self.ctx.last_stage_ms())
        self.stage_samples += 1


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes on this same default workload; tools/summarize_profiles.py).  None when no profile covers the kernel."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json'))):
        try:
            v = json.load(open(f)).get('bytes_per_launch', {}).get(kernel)
        except (OSError, ValueError):
            v = None
        if v is not None:
            best = v
    return best


def host_boundary_rate(shard, repeats=2):
    """Same batch handed over as HOST buffers through the reference-shaped entry point (zkv_risc0_verify_batch /
    zkv_sp1_verify_batch): H2D staging, all stages and the status D2H inside the call.  Reported next to `value`, never as it."""
    import ctypes as C
    from stylus_zkvm_verifiers_amd import _lib
    L = _lib.lib()
    n = shard.n
    seals = np.ascontiguousarray(shard.h_seals)
    off = (np.arange(n + 1, dtype=np.uint64) * 260)
    a = np.ascontiguousarray(shard.h_a); b = np.ascontiguousarray(shard.h_b)
    st = np.zeros(n, dtype=np.uint8)
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        if shard.vm == 'risc0':
            rc = L.zkv_risc0_verify_batch(shard.ctx._h, n, seals.ctypes.data, off.ctypes.data, a.ctypes.data, b.ctypes.data, st.ctypes.data, None)
        else:
            pvoff = (np.arange(n + 1, dtype=np.uint64) * b.shape[1])
            rc = L.zkv_sp1_verify_batch(shard.ctx._h, n, a.ctypes.data, b.ctypes.data, pvoff.ctypes.data, seals.ctypes.data, off.ctypes.data,
                                        st.ctypes.data, None)
        dt = time.perf_counter() - t0
        _lib.check(rc, 'host batch')
        best = dt if best is None else min(best, dt)
    ok = bool(((st == 0) == ~shard.mutated).all())
    return {'value': n / best, 'unit': 'proofs/s', 'ms': best * 1e3, 'statuses_match_construction': ok,
            'note': 'host buffers in, statuses out: PCIe staging and copies inside the call (pageable memory, synchronous per chunk)'}


def wire_leg(shard, dev, stream, steps=3):
    """The same batch arriving as eth_call calldata already resident in HBM (`verify(uint8[],bytes32,bytes32)` /
    `verifyProof(bytes32,uint8[],uint8[])`: one 32-byte word per seal byte): decode kernel (HBM-bound, its own roofline line)
    followed by the usual stages.  Reported next to `value`, never as it."""
    from stylus_zkvm_verifiers_amd import synth, wire
    if shard.vm == 'risc0':
        cd = synth.calldata_risc0_verify(shard.h_seals, shard.h_a, shard.h_b)
    else:
        cd = synth.calldata_sp1_verify_proof(shard.h_a, shard.h_b, shard.h_seals)
    n = shard.n
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(cd.shape[1])
    d_cd = torch.from_numpy(cd).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    run = lambda: wire.eth_call_batch_dev(shard.ctx, n, d_cd.data_ptr(), d_off.data_ptr(), cd.size, d_st.data_ptr(), 0, stream)
    run(); torch.cuda.synchronize()
    wire_ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
        wire_ms.append(wire.last_wire_ms(shard.ctx))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    same = bool((d_st.cpu().numpy() == shard.d_status.cpu().numpy()).all())
    # the same requests handed over as HOST calldata (zkv_*_eth_call_batch): 8.4-11.6 KB per proof cross PCIe inside the call
    host_rate = None
    if n <= 1 << 16:
        import ctypes as C
        from stylus_zkvm_verifiers_amd import _lib
        L = _lib.lib()
        rv8 = np.zeros(n, dtype=np.uint8); st8 = np.zeros(n, dtype=np.uint8)
        retb = np.zeros((n, wire.RETURNDATA_STRIDE), dtype=np.uint8); rl = np.zeros(n, dtype=np.uint32)
        fn = L.zkv_risc0_eth_call_batch if shard.vm == 'risc0' else L.zkv_sp1_eth_call_batch
        t0 = time.perf_counter()
        _lib.check(fn(shard.ctx._h, n, cd.ctypes.data, off.ctypes.data, rv8.ctypes.data, retb.ctypes.data, rl.ctypes.data, st8.ctypes.data), 'eth_call_batch')
        host_rate = n / (time.perf_counter() - t0)
        same = same and bool((st8 == shard.d_status.cpu().numpy()).all())
    chunk = int(os.environ.get('ZKV_CHUNK', 1 << 20))
    last = n - (-(-n // chunk) - 1) * chunk                      # the events bracket the last chunk's decode launch
    ms = float(np.mean(wire_ms))
    out_bytes = 260 + (64 if shard.vm == 'risc0' else 32 + shard.h_b.shape[1]) + 4 + 1
    gbs = last * (cd.shape[1] + out_bytes) / (ms * 1e-3) / 1e9
    default = shard.vm == 'risc0' and n == 1 << 16
    return {'kernel': 'k_wire_' + shard.vm, 'traffic': pmc_traffic('k_wire_' + shard.vm) if default else None, 'calldata_bytes_per_proof': int(cd.shape[1]), 'decoded_bytes_per_proof': out_bytes,
            'kernel_ms': ms, 'proofs_per_launch': int(last), 'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': gbs / HBM_PEAK_GBS, 'end_to_end_proofs_per_s': n / dt, 'host_calldata_proofs_per_s': host_rate,
            'statuses_equal_seal_path': same}


def host_cores():
    """CPU threads this process may actually use: affinity mask, capped by the cgroup CPU quota when one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(shard, budget_s=15.0):
    """The CPU oracle (oracle/zkv_oracle.c, a port of the reference-shaped path) on the host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle_lib as ol
    H = bytes.fromhex
    g = golden()
    cores = host_cores()
    def run(k):
        seals = [shard.h_seals[i].tobytes() for i in range(k)]
        a = [shard.h_a[i].tobytes() for i in range(k)]
        b = [shard.h_b[i].tobytes() for i in range(k)]
        t0 = time.time()
        if shard.vm == 'risc0':
            v = ol.Risc0Oracle(); v.initialize(H(g['risc0']['control_root']), H(g['risc0']['bn254_control_id']))
            st, _ = v.verify_batch(seals, a, b, threads=cores)
        else:
            st, _ = ol.sp1_verify_batch(a, b, seals, threads=cores)
        return time.time() - t0, st
    probe = min(shard.n, 8 * cores)
    dt, _ = run(probe)
    k = int(min(shard.n, max(probe, budget_s * probe / max(dt, 1e-6))))
    dt, st = run(k)
    return {'value': k / dt, 'unit': 'proofs/s', 'cores': cores, 'kind': 'port',
            'sample': 'first %d proofs of the same %s batch, OpenMP over %d threads, %.1f s' % (k, shard.vm, cores, dt)}, st, k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='risc0_2p16', choices=list(SEEDS))
    ap.add_argument('--proofs', dest='n', type=int, default=0, help='override proofs per GPU')
    ap.add_argument('--mutate-every', type=int, default=64)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-wire', action='store_true', help='skip the eth_call calldata leg (N=1 only)')
    ap.add_argument('--rehearse-single-gpu', action='store_true',
                    help='multi-process rehearsal on a one-GPU box: every rank uses cuda:0 and the collectives run over gloo on host tensors')
    args = ap.parse_args()

    from stylus_zkvm_verifiers_amd import parallel
    rank, local_rank, world = parallel.init_distributed('gloo' if args.rehearse_single_gpu else None)
    assert world == args.gpus, 'launch one process per GPU (WORLD_SIZE=%d, --gpus %d)' % (world, args.gpus)
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)')
    if args.rehearse_single_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    cdev = torch.device('cpu') if args.rehearse_single_gpu else dev          # where the collectives' tensors live
    g = golden()
    seed = SEEDS[args.workload] + 0x1000 * rank
    if args.workload == 'risc0_2p16':
        n = args.n or (1 << 16)
        shards = [Shard('risc0', n, seed, dev, g, args.mutate_every)]
    elif args.workload == 'sp1_2p20':
        n = args.n or (1 << 20)
        shards = [Shard('sp1', n, seed, dev, g, args.mutate_every)]
    else:
        n = args.n or (1 << 19)
        shards = [Shard('risc0', n // 2, seed, dev, g, args.mutate_every), Shard('sp1', n - n // 2, seed + 1, dev, g, args.mutate_every)]
    n_rank = sum(s.n for s in shards)
    stream = torch.cuda.current_stream().cuda_stream

    def step(sample):
        for s in shards:
            s.enqueue(stream)
            if sample:
                s.sample_stage_ms()

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of the timed work: every valid proof accepted, every mutated proof rejected
    ok_counts, parity = [], True
    for s in shards:
        st = s.d_status.cpu().numpy()
        ok_counts.append(int((st == 0).sum()))
        parity &= bool(((st == 0) == ~s.mutated).all())
    flag = torch.tensor([1 if parity else 0], dtype=torch.int32, device=cdev)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    parity_all = bool(flag.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total = n_rank * world
        value = total * args.steps / elapsed
        # dominant kernel of the dominant shard, measured with HIP events on the launch stream inside the timed region
        dom = max(shards, key=lambda s: s.stage_ms.sum())
        avg = dom.stage_ms / max(dom.stage_samples, 1)
        k = int(np.argmax(avg))
        launches = -(-dom.n // (1 << 20)) if not os.environ.get('ZKV_CHUNK') else -(-dom.n // int(os.environ['ZKV_CHUNK']))
        per_launch = min(dom.n, int(os.environ.get('ZKV_CHUNK', 1 << 20)))
        last_chunk = dom.n - (launches - 1) * per_launch          # events bracket the last chunk of a step
        bpp = BYTES_PER_PROOF[dom.vm]
        achieved = last_chunk * bpp / (avg[k] * 1e-3) / 1e9
        pair = os.environ.get('ZKV_LANES_PER_PROOF', '2') != '1'
        kname = 'k_' + STAGES[k] + ('_risc0' if k == 0 and dom.vm == 'risc0' else '_sp1' if k == 0 else '') + ('2' if pair and k >= 2 else '')
        traffic = pmc_traffic(kname) if (args.workload == 'risc0_2p16' and not args.n) else None
        out = {
            'metric': 'Groth16 proofs verified/sec (BN254)', 'value': value, 'unit': 'proofs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u32', 'data': 'synthetic',
            'config': {'workload': args.workload, 'proofs_per_gpu': n_rank, 'global_batch': total,
                       'mutated_fraction': (1.0 / args.mutate_every) if args.mutate_every else 0.0,
                       'parallelism': 'dp%d (independent shards, no data-path collective)' % world},
            'roofline': {'bound': 'hbm', 'kernel': kname, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel_ms': float(avg[k]), 'proofs_per_launch': int(last_chunk), 'algorithmic_bytes_per_proof': bpp},
            'stage_ms': {STAGES[i]: float(avg[i]) for i in range(5)},
            'parity': {'accept_reject_matches_construction': parity_all, 'ok_counts_rank0': ok_counts,
                       'expected_ok_rank0': [s.expected_ok for s in shards]},
        }
        if world == 1:
            out['host_boundary'] = host_boundary_rate(shards[0])
        if world == 1 and not args.no_wire and shards[0].n * 12000 < (8 << 30):       # calldata blob must fit comfortably
            out['wire'] = wire_leg(shards[0], dev, stream)
        if world == 1 and not args.no_cpu_baseline:
            base, cst, kk = cpu_baseline(shards[0])
            out['cpu_baseline'] = base
            gst = shards[0].d_status.cpu().numpy()[:kk]
            out['parity']['gpu_equals_cpu_oracle_on_sample'] = bool((gst == cst).all())
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
