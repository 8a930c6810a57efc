#!/usr/bin/env python3
"""Latency of the ecPairing seam for few calls: the reference's own 768-byte calldata (4 pairs) through zkv_bn254_pairing_batch,
two-wavefront kernels (default for small batches) against the lane-pair kernels (ZKV_DUAL_BELOW=0)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import stylus_zkvm_verifiers_amd as z
import spec_model as m
H = bytes.fromhex
r = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'real_proofs.json')))['risc0']
seal = H(r['seal']); w = [seal[4 + 32 * i:36 + 32 * i] for i in range(8)]
ax, ay = m.negate_g1_words(int.from_bytes(w[0], 'big'), int.from_bytes(w[1], 'big'))
vk = m.RISC0_VK
g2 = lambda q: b''.join(m.be32(v) for v in (q[0][0], q[0][1], q[1][0], q[1][1]))
data = (m.be32(ax) + m.be32(ay) + b''.join(w[2:6]) + m.be32(vk['alpha1'][0]) + m.be32(vk['alpha1'][1]) + g2(vk['beta2'])
        + H(r['vk_x'][0]) + H(r['vk_x'][1]) + g2(vk['gamma2']) + w[6] + w[7] + g2(vk['delta2']))
pc = z.Bn254Precompiles()
out = {}
for dual in ('768', '0'):
    os.environ['ZKV_DUAL_BELOW'] = dual
    for n in (1, 64, 512):
        assert pc.pairing([data] * n, 4) == [True] * n
        best = min((lambda t0: (pc.pairing([data] * n, 4), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(5))
        out['dual_below=%s n=%d' % (dual, n)] = round(best * 1e3, 3)
print(json.dumps(out))
