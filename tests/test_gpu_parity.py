"""Parity of the HIP path (through the C ABI) against the golden vectors and the CPU oracle.  Needs an MI355X.

Pinned by the reference: the two real proofs (examples/*/examples/interact.rs) must ACCEPT.
Everything else is agreement with the oracle / spec model => "parity unpinned" (SURVEY.md 8c)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
H = bytes.fromhex


@pytest.fixture(scope='module')
def zkv():
    import stylus_zkvm_verifiers_amd as z
    assert z.device_count() >= 1, 'no gfx950 device visible'
    return z


@pytest.fixture(scope='module')
def r0(zkv, real_proofs):
    v = zkv.RiscZeroVerifier()
    v.initialize(H(real_proofs['risc0']['control_root']), H(real_proofs['risc0']['bn254_control_id']))
    yield v
    v.close()


@pytest.fixture(scope='module')
def sp1(zkv):
    v = zkv.Sp1Verifier()
    yield v
    v.close()


def test_real_risc0_proof_accepts(r0, real_proofs):
    r = real_proofs['risc0']
    assert r0.get_selector().hex() == r['selector']
    assert r0.verify(H(r['seal']), H(r['image_id']), H(r['journal_digest'])) is True
    assert r0.verify_integrity(H(r['seal']), H(r['claim_digest'])) is True


def test_real_sp1_proof_accepts(sp1, real_proofs):
    s = real_proofs['sp1']
    assert sp1.verify_proof(H(s['vkey']), H(s['public_values']), H(s['proof'])) is None


def test_verify_corpus_matches_golden_and_oracle(zkv, r0, sp1, verify_corpus):
    import oracle_lib as ol
    orc = ol.Risc0Oracle()
    orc.initialize(H(verify_corpus['risc0_ctx']['control_root']), H(verify_corpus['risc0_ctx']['bn254_control_id']))
    rc = [c for c in verify_corpus['cases'] if c['vm'] == 'risc0']
    sc = [c for c in verify_corpus['cases'] if c['vm'] == 'sp1']
    st, rv = r0.verify_batch([H(c['seal']) for c in rc], [H(c['image_id']) for c in rc], [H(c['journal_digest']) for c in rc])
    for c, s, r in zip(rc, st, rv):
        ost, orecv = orc.verify(H(c['seal']), H(c['image_id']), H(c['journal_digest']))
        assert int(s) == c['status'] == ost, c['name']
        assert bytes(r).hex() == (c['received'] or '00000000'), c['name']
    st, rv = sp1.verify_batch([H(c['vkey']) for c in sc], [H(c['public_values']) for c in sc], [H(c['proof']) for c in sc])
    for c, s, r in zip(sc, st, rv):
        ost, orecv = ol.sp1_verify_proof(H(c['vkey']), H(c['public_values']), H(c['proof']))
        assert int(s) == c['status'] == ost, c['name']
        assert bytes(r).hex() == (c['received'] or '00000000'), c['name']


def test_single_call_errors_carry_revert_bytes(zkv, r0, sp1, verify_corpus, revert_vectors):
    for c in verify_corpus['cases']:
        if c['status'] == 0:
            continue
        with pytest.raises(zkv.VerifierError) as ei:
            if c['vm'] == 'risc0':
                r0.verify(H(c['seal']), H(c['image_id']), H(c['journal_digest']))
            else:
                sp1.verify_proof(H(c['vkey']), H(c['public_values']), H(c['proof']))
        assert ei.value.status == c['status'], c['name']
        if c['status'] == 5:
            assert ei.value.received.hex() == c['received']
            assert ei.value.revert[4:8].hex() == c['received']


def test_context_cases(zkv, verify_corpus):
    for c in verify_corpus['ctx_cases']:
        v = zkv.RiscZeroVerifier()
        if c['name'] == 'second initialize':
            cr = H(verify_corpus['risc0_ctx']['control_root']); cid = H(verify_corpus['risc0_ctx']['bn254_control_id'])
            v.initialize(cr, cid)
            with pytest.raises(zkv.VerifierError) as ei:
                v.initialize(cr, cid)
            assert ei.value.status == c['status']
            continue
        if c['control_root'] is not None:
            v.initialize(H(c['control_root']), H(c['bn254_control_id']))
            assert v.get_selector().hex() == c['selector']
        st, _ = v.verify_batch([H(c['seal'])], [H(c['image_id'])], [H(c['journal_digest'])])
        assert int(st[0]) == c['status'], c['name']
        v.close()


def test_empty_batch(r0, sp1):
    st, rv = r0.verify_batch([], [], [])
    assert len(st) == 0
    st, rv = sp1.verify_batch([], [], [])
    assert len(st) == 0
