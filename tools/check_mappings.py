"""Quick device check of the kernel mappings (lane pairs / 16 lanes per proof / automatic) at a few batch sizes and of the PLONK path:
prints the statuses (0 = verified) and stage times.  `python tools/check_mappings.py` on a GPU box."""
import json, sys
sys.path.insert(0, '.')
import stylus_zkvm_verifiers_amd as z
H = bytes.fromhex
g = json.load(open('tests/golden/real_proofs.json'))
r = g['risc0']
v = z.RiscZeroVerifier(0); v.initialize(H(r['control_root']), H(r['bn254_control_id']))
for lanes in (2, 16, 0):
    v.set_lanes_per_proof(lanes)
    for n in (1, 3, 64, 100):
        st, _ = v.verify_batch([H(r['seal'])] * n, [H(r['image_id'])] * n, [H(r['journal_digest'])] * n)
        print('lanes', lanes, 'n', n, list(st[:8]), v.last_stage_ms())
pk = json.load(open('tests/golden/plonk_cases.json'))
p = z.Sp1PlonkVerifier(H(pk['vk']), H(pk['verifier_hash']), 0)
c = pk['cases'][0]
for lanes in (2, 16):
    p.set_lanes_per_proof(lanes)
    st, _ = p.verify_batch([H(c['vkey'])] * 4, [H(c['public_values'])] * 4, [H(c['proof'])] * 4)
    print('plonk lanes', lanes, list(st))
