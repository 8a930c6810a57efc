#!/usr/bin/env python3
"""Generates tests/golden/plonk_cases.json and tests/golden/plonk_pool.json from the spec model of the PLONK path
(oracle/plonk_model.py: trapdoor key for a toy circuit, gnark-style verifier).  PARITY UNPINNED BY CONSTRUCTION: the reference
holds no PLONK code, key or proof (README.md:25); the expected statuses below are the spec model's.

    python tests/golden/make_plonk_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..', 'oracle'))
import plonk_model as pm      # noqa: E402
import spec_model as m        # noqa: E402

VERIFIER_HASH = bytes.fromhex('d4e8ecd2357dd882209800acd6abb443d231cf287d77ba62b732ce937c8b56e7')    # stand-in: any 32 bytes; the selector is its first four


def main():
    rng = random.Random(0x5A4B5605)
    circ = pm.ToyCircuit(rng)
    vk = circ.vk
    cases = []

    def add(name, vkey, pv, proof):
        st, recv = pm.sp1_plonk_verify_proof(vk, VERIFIER_HASH, vkey, pv, proof)
        cases.append(dict(name=name, vkey=vkey.hex(), public_values=pv.hex(), proof=proof.hex(), status=st, received=recv.hex() if recv else None))

    def fresh(pv_len=None):
        vkey = m.be32(rng.randrange(m.R))
        pv = bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 32, 96, 200]) if pv_len is None else pv_len))
        words = circ.prove(int.from_bytes(vkey, 'big'), m.sp1_hash_public_values(pv))
        return vkey, pv, VERIFIER_HASH[:4] + words

    for k in range(6):
        vkey, pv, proof = fresh()
        add('valid %d (public values %d bytes)' % (k, len(pv)), vkey, pv, proof)
    vkey, pv, proof = fresh(96)
    add('valid, 96-byte public values', vkey, pv, proof)
    # ---- ordered checks of verify_proof
    for n in (0, 3, 4, 5, 867, 869, 900):
        add('length %d' % n, vkey, pv, (proof + bytes(64))[:n])
    add('wrong selector', vkey, pv, bytes([proof[0] ^ 1]) + proof[1:])
    add('wrong selector and wrong length', vkey, pv, b'\x01\x02\x03\x04' + proof[4:100])
    # ---- public inputs
    add('public values changed', vkey, pv[:-1] + bytes([pv[-1] ^ 1]), proof)
    add('program vkey changed', m.be32((int.from_bytes(vkey, 'big') + 1) % m.R), pv, proof)
    add('program vkey = R (not a field element)', m.be32(m.R), pv, proof)
    add('program vkey = 2^256 - 1', b'\xff' * 32, pv, proof)
    # ---- every word of the proof tampered
    names = ['L.x', 'L.y', 'R.x', 'R.y', 'O.x', 'O.y', 'H0.x', 'H0.y', 'H1.x', 'H1.y', 'H2.x', 'H2.y', 'l(zeta)', 'r(zeta)', 'o(zeta)', 's1(zeta)', 's2(zeta)',
             'Z.x', 'Z.y', 'z(omega zeta)', 'Hzeta.x', 'Hzeta.y', 'Hzetaomega.x', 'Hzetaomega.y', 'qcp(zeta)', 'BSB22.x', 'BSB22.y']
    for i, nm in enumerate(names):
        bad = bytearray(proof); bad[4 + 32 * i + 31] ^= 1
        add('word %d (%s) low bit flipped' % (i, nm), vkey, pv, bytes(bad))
    word = lambda i: int.from_bytes(proof[4 + 32 * i:36 + 32 * i], 'big')
    setw = lambda i, v: proof[:4 + 32 * i] + m.be32(v) + proof[36 + 32 * i:]
    for i in (12, 16, 19, 24):
        add('scalar word %d += R (same residue, not canonical)' % i, vkey, pv, setw(i, word(i) + m.R) if word(i) + m.R < (1 << 256) else setw(i, m.R))
    for i in (0, 7, 17, 22, 26):
        add('coordinate word %d += P (same residue, not canonical)' % i, vkey, pv, setw(i, word(i) + m.P) if word(i) + m.P < (1 << 256) else setw(i, m.P))
    # ---- points replaced by other valid points / infinity
    g = m.G1_GEN
    setp = lambda i, pt: proof[:4 + 32 * i] + pm.g1_bytes(pt) + proof[68 + 32 * i:]
    for i, nm in ((0, 'L'), (6, 'H0'), (17, 'Z'), (20, 'Hzeta'), (22, 'Hzetaomega'), (25, 'BSB22')):
        add('%s = infinity' % nm, vkey, pv, setp(i, None))
        add('%s = generator' % nm, vkey, pv, setp(i, g))
        add('%s negated' % nm, vkey, pv, setp(i, (word(i), (-word(i + 1)) % m.P)))
    add('L and R swapped', vkey, pv, proof[:4] + proof[68:132] + proof[4:68] + proof[132:])
    # ---- a witness that does not satisfy the circuit, and a proof for other inputs
    bad = circ.prove(int.from_bytes(vkey, 'big'), m.sp1_hash_public_values(pv), tamper='o5')
    add('witness violates a gate', vkey, pv, VERIFIER_HASH[:4] + bad)
    v2, p2, pr2 = fresh(96)
    add('valid proof for other public inputs', vkey, pv, pr2)
    add('all-zero proof body', vkey, pv, VERIFIER_HASH[:4] + bytes(864))
    out = dict(note='PARITY UNPINNED BY CONSTRUCTION: no PLONK code, key or proof exists in the reference; statuses are the spec model\'s (oracle/plonk_model.py)',
               vk=pm.vk_bytes(vk).hex(), verifier_hash=VERIFIER_HASH.hex(), cases=cases)
    with open(os.path.join(HERE, 'plonk_cases.json'), 'w') as f:
        json.dump(out, f, indent=0)
    print('plonk_cases.json:', len(cases), 'cases;', sum(1 for c in cases if c['status'] == 0), 'accept')
    # ---- pool of valid proofs with distinct public inputs (fixed 96-byte public values) for batches: bench.py tiles it
    pool = []
    for _ in range(64):
        vkey, pv, proof = fresh(96)
        assert pm.sp1_plonk_verify_proof(vk, VERIFIER_HASH, vkey, pv, proof)[0] == 0
        pool.append(dict(vkey=vkey.hex(), public_values=pv.hex(), proof=proof.hex()))
    with open(os.path.join(HERE, 'plonk_pool.json'), 'w') as f:
        json.dump(dict(note='64 valid toy-circuit PLONK proofs under the trapdoor key of plonk_cases.json, distinct public inputs (data for batches; see make_plonk_golden.py)',
                       vk=pm.vk_bytes(vk).hex(), verifier_hash=VERIFIER_HASH.hex(), proofs=pool), f, indent=0)
    print('plonk_pool.json:', len(pool), 'proofs')


if __name__ == '__main__':
    main()
