"""Throughput of the ecPairing seam (zkv_bn254_pairing_batch, host buffers): n calls of the reference's own 4-pair calldata
(common/groth16.rs:109-128) built from re-randomised real proofs, results against the expectation.  python tools/bench_seam.py [--log2 16]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import bench
import spec_model as m
H = bytes.fromhex


def ecmul(pc, z, sizes):
    rng = np.random.default_rng(0xEC)
    pts = [m.g1_mul((1, 2), k) for k in (1, 5, 0x1234567)]
    for lg in sizes:
        n = 1 << lg
        blob = np.zeros((n, 96), dtype=np.uint8)
        for i in range(n):
            p = pts[i % 3]
            blob[i, :64] = np.frombuffer(m.be32(p[0]) + m.be32(p[1]), dtype=np.uint8)
        blob[:, 64:] = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        out = np.zeros((n, 64), dtype=np.uint8); ok = np.zeros(n, dtype=np.uint8)
        call = lambda: z._lib.check(pc._L.zkv_bn254_ecmul_batch(pc._h, n, blob.ctypes.data, out.ctypes.data, ok.ctypes.data), 'zkv_bn254_ecmul_batch')
        call()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); call(); best = min(best, time.perf_counter() - t0)
        chk = all(out[i].tobytes() == (lambda q: m.be32(q[0]) + m.be32(q[1]))(m.g1_mul(pts[i % 3], int.from_bytes(blob[i, 64:].tobytes(), 'big') % m.R)) for i in range(0, n, max(1, n // 50)))
        print(json.dumps({'ecmul_calls': n, 'ms': round(best * 1e3, 3), 'calls_per_s': round(n / best), 'all_ok': bool(ok.all()), 'sample_equals_spec_model': bool(chk)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--log2', default='12,16')
    ap.add_argument('--pairs', default='4')
    ap.add_argument('--ecmul', action='store_true', help='time zkv_bn254_ecmul_batch (random 256-bit scalars on a few points) instead')
    args = ap.parse_args()
    g = bench.golden()
    r = g['risc0']
    vk = m.RISC0_VK
    g2 = lambda q: b''.join(m.be32(v) for v in (q[0][0], q[0][1], q[1][0], q[1][1]))
    tail = (m.be32(vk['alpha1'][0]) + m.be32(vk['alpha1'][1]) + g2(vk['beta2']), H(r['vk_x'][0]) + H(r['vk_x'][1]) + g2(vk['gamma2']), g2(vk['delta2']))
    import stylus_zkvm_verifiers_amd as z
    pc = z.Bn254Precompiles()
    if args.ecmul:
        return ecmul(pc, z, [int(x) for x in args.log2.split(',')])
    for lg in [int(x) for x in args.log2.split(',')]:
        n = 1 << lg
        host = bench.synthesize('risc0', n, 0x5A4B5601, g, 64)          # every 64th proof mutated; flip-input ones stay valid here (vk_x is the real one)
        seals = host['seals']
        calls = np.zeros((n, 768), dtype=np.uint8)
        w = seals[:, 4:].reshape(n, 8, 32)
        P = int(m.P)
        for i in range(n):
            ax, ay = int.from_bytes(w[i, 0].tobytes(), 'big'), int.from_bytes(w[i, 1].tobytes(), 'big')
            if ax < P and ay < P: ax, ay = m.negate_g1_words(ax, ay)
            row = (m.be32(ax % (1 << 256)) + m.be32(ay % (1 << 256)) + w[i, 2:6].tobytes() + tail[0] + tail[1] + w[i, 6].tobytes() + w[i, 7].tobytes() + tail[2])
            calls[i] = np.frombuffer(row, dtype=np.uint8)
        for k in [int(x) for x in args.pairs.split(',')]:
            blob = np.ascontiguousarray(calls[:, :192 * k])          # the C ABI call itself (host buffers in, results out), no Python marshalling
            out_b = np.zeros(n, dtype=np.uint8); ok_b = np.zeros(n, dtype=np.uint8)
            L = pc._L
            call = lambda: z._lib.check(L.zkv_bn254_pairing_batch(pc._h, n, k, blob.ctypes.data, out_b.ctypes.data, ok_b.ctypes.data), 'zkv_bn254_pairing_batch')
            call()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); call(); best = min(best, time.perf_counter() - t0)
            out = {'calls': n, 'pairs_per_call': k, 'ms': round(best * 1e3, 3), 'calls_per_s': round(n / best), 'true_results': int(((out_b == 1) & (ok_b == 1)).sum()),
                   'invalid_inputs': int((ok_b == 0).sum())}
            import torch
            dev = torch.device('cuda', 0)
            d_in = torch.from_numpy(blob).to(dev); d_res = torch.zeros(n, dtype=torch.uint8, device=dev); d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            dcall = lambda: z._lib.check(L.zkv_bn254_pairing_batch_dev(pc._h, n, k, d_in.data_ptr(), d_res.data_ptr(), d_ok.data_ptr(), st), 'zkv_bn254_pairing_batch_dev')
            dcall(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): dcall()
            torch.cuda.synchronize()
            dms = (time.perf_counter() - t0) * 1e3 / 5
            out['device_resident_ms'] = round(dms, 3); out['device_resident_calls_per_s'] = round(n / dms * 1e3)
            out['device_resident_equals_host'] = bool((d_res.cpu().numpy() == out_b).all() and (d_ok.cpu().numpy() == ok_b).all())
            print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
