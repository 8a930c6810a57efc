#!/usr/bin/env python3
"""Small-batch probe: the golden corpus and one single verification through the C ABI, with the per-stage HIP-event times.
Run it with ZKV_WIDE_BELOW=0 and without to compare the lane-pair kernels with the 16-lanes-per-proof kernels."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import stylus_zkvm_verifiers_amd as z
import oracle_lib as ol
H = bytes.fromhex
g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'real_proofs.json'))); corpus = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'verify_corpus.json')))
r = g['risc0']
v = z.RiscZeroVerifier(); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
rc = [c for c in corpus['cases'] if c['vm'] == 'risc0']
st, _ = v.verify_batch([H(c['seal']) for c in rc], [H(c['image_id']) for c in rc], [H(c['journal_digest']) for c in rc])
bad = [(c['name'], int(s), c['status']) for c, s in zip(rc, st) if int(s) != c['status']]
print('risc0 corpus mismatches:', bad[:10], 'of', len(rc), 'stage ms', v.last_stage_ms())
sc = [c for c in corpus['cases'] if c['vm'] == 'sp1']
sp = z.Sp1Verifier()
st, _ = sp.verify_batch([H(c['vkey']) for c in sc], [H(c['public_values']) for c in sc], [H(c['proof']) for c in sc])
bad = [(c['name'], int(s), c['status']) for c, s in zip(sc, st) if int(s) != c['status']]
print('sp1 corpus mismatches:', bad[:10], 'of', len(sc), 'stage ms', sp.last_stage_ms())
t0 = time.time(); ok = v.verify(H(r['seal']), H(r['image_id']), H(r['journal_digest'])); print('single verify', ok, (time.time() - t0) * 1e3, 'ms', v.last_stage_ms())
# host-call latency of the reference-shaped single-proof entry points (host buffers in, status out, everything inside the call)
for name, fn in (('zkv_risc0_verify', lambda: v.verify(H(r['seal']), H(r['image_id']), H(r['journal_digest']))),
                 ('zkv_sp1_verify_proof', lambda: sp.verify_proof(H(g['sp1']['vkey']), H(g['sp1']['public_values']), H(g['sp1']['proof'])))):
    fn()
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print('%s host call: median %.3f ms, min %.3f ms, p90 %.3f ms over 50 calls' % (name, ts[25], ts[0], ts[45]))
