"""TEST INFRASTRUCTURE -- slow spec model of the SP1 PLONK path (BASELINE.json configs[4]).

PARITY UNPINNED BY CONSTRUCTION: the reference holds no PLONK code, key or proof -- only the words "PLONK in progress"
(/root/reference/README.md:25, /root/reference/contracts/src/lib.rs:11).  The entry-point shape is the reference's
`ISp1Verifier::verify_proof(program_vkey, public_values, proof_bytes)` (/root/reference/contracts/src/sp1/verifier.rs:16-29, 58-111:
length check, selector check, strict decode, two public inputs = (program_vkey, hash_public_values), VerificationFailed otherwise).
The verification algorithm restated here is the published BN254 PLONK verifier of gnark (github.com/Consensys/gnark v0.10-0.11,
backend/plonk/bn254/verify.go, and gnark-crypto ecc/bn254/kzg + fiat-shamir + hash_to_field), which is what SP1 v5.0.0 wraps
for its on-chain PLONK verifier; neither package is in the container, so this file restates the algorithm from its published
description:
  * Fiat-Shamir transcript over SHA-256 with the challenge names "gamma", "beta", "alpha", "zeta"; a challenge hashes its name,
    the previous challenge's raw 32 bytes and its bindings (verifying-key commitments and public inputs for gamma, then the wire
    commitments; the BSB22 commitments and Z for alpha; the quotient commitments for zeta);
  * the BSB22 commitment enters the public-input polynomial through hash_to_field (RFC 9380 expand_message_xmd, SHA-256,
    DST "BSB22-Plonk", 48 bytes reduced mod r) at row nb_public + commitment_constraint_index;
  * linearised polynomial with the quotient folded in, batched KZG opening at zeta (folding challenge from a second "gamma"
    transcript), single opening of Z at omega*zeta, the two openings batched with a hash-derived lambda as the Solidity
    template does (the Go verifier samples lambda at random), one 2-pair pairing check.
  * proof wire format = the 27 words of gnark's MarshalSolidity for one BSB22 commitment, after the 4-byte selector (868 bytes).
Because no real SP1 PLONK verifying key is available, keys here are TRAPDOOR keys for a toy circuit (known tau): the prover below
produces valid proofs for arbitrary public inputs, which is what the parity tests need.
"""
import hashlib

import spec_model as m

R = m.R
P = m.P
OK, VERIFICATION_FAILED, INVALID_PROOF_DATA, SELECTOR_MISMATCH = 0, 1, 4, 5
PROOF_WORDS = 27
PROOF_BYTES = 4 + 32 * PROOF_WORDS


def finv(a): return pow(a % R, -1, R)
def be32(x): return int(x).to_bytes(32, 'big')
def g1_bytes(pt): return bytes(64) if pt is None else be32(pt[0]) + be32(pt[1])
def g1_wire(pt): return (0, 0) if pt is None else pt          # a point as two words: (0, 0) = infinity


# ---------------------------------------------------------------- hashing
def expand_message_xmd(msg, dst, n_bytes):
    """RFC 9380 5.3.1 with SHA-256."""
    ell = -(-n_bytes // 32)
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(bytes(64) + msg + n_bytes.to_bytes(2, 'big') + b'\0' + dst_prime).digest()
    b = [hashlib.sha256(b0 + b'\x01' + dst_prime).digest()]
    for i in range(2, ell + 1):
        b.append(hashlib.sha256(bytes(x ^ y for x, y in zip(b0, b[-1])) + bytes([i]) + dst_prime).digest())
    return b''.join(b)[:n_bytes]


def hash_to_field_bsb22(msg):
    return int.from_bytes(expand_message_xmd(msg, b'BSB22-Plonk', 48), 'big') % R


class Transcript:
    """gnark-crypto fiat-shamir: challenge = H(name || previous challenge bytes (if any) || bindings in order)."""

    def __init__(self, *names):
        self.names = list(names); self.bind = {k: [] for k in names}; self.value = {}

    def add(self, name, data):
        self.bind[name].append(bytes(data))

    def challenge(self, name):
        i = self.names.index(name)
        h = hashlib.sha256(name.encode())
        if i:
            h.update(self.value[self.names[i - 1]])
        for b in self.bind[name]:
            h.update(b)
        self.value[name] = h.digest()
        return self.value[name]


# ---------------------------------------------------------------- verifying key
def vk_bytes(vk):
    """Serialisation handed to zkv_sp1_plonk_ctx_create (include/zkv.h): 32-byte big-endian words
    size | size_inv | generator | coset_shift | nb_public | n_qcp (0 or 1) | commitment_constraint_index |
    S1 S2 S3 Ql Qr Qm Qo Qk [Qcp] (G1: x, y) | G2 generator | [tau]G2 (EIP-197 order x_im x_re y_im y_re)."""
    out = b''.join(be32(v) for v in (vk['size'], vk['size_inv'], vk['generator'], vk['coset_shift'], vk['nb_public'], len(vk['qcp']),
                                     vk['cci'][0] if vk['cci'] else 0))
    for k in ('s1', 's2', 's3', 'ql', 'qr', 'qm', 'qo', 'qk'):
        out += g1_bytes(vk[k])
    for q in vk['qcp']:
        out += g1_bytes(q)
    for q in (vk['g2'], vk['g2_tau']):
        (x0, x1), (y0, y1) = m.g2_words(q)
        out += be32(x0) + be32(x1) + be32(y0) + be32(y1)
    return out


# ---------------------------------------------------------------- verifier (gnark backend/plonk/bn254/verify.go)
def parse_proof(words):
    w = [int.from_bytes(words[32 * i:32 * i + 32], 'big') for i in range(PROOF_WORDS)]
    pt = lambda i: (w[i], w[i + 1])
    return dict(lro=[pt(0), pt(2), pt(4)], h=[pt(6), pt(8), pt(10)], l=w[12], r=w[13], o=w[14], s1=w[15], s2=w[16],
                z=pt(17), zu=w[19], h_zeta=pt(20), h_zeta_omega=pt(22), qcp=[w[24]], bsb=[pt(25)])


def _g1(pt):
    """A proof / key point as the precompiles would take it: coordinates < P, on the curve or (0,0) = infinity."""
    x, y = pt
    if x >= P or y >= P:
        raise ValueError('coordinate >= P')
    if x == 0 and y == 0:
        return None
    if not m.g1_on_curve((x, y)):
        raise ValueError('not on curve')
    return (x, y)


def plonk_verify(vk, proof_words, public_inputs):
    """True / False.  public_inputs: list of ints (each must be < R).  proof_words: 27 x 32 bytes."""
    if len(proof_words) != 32 * PROOF_WORDS or len(public_inputs) != vk['nb_public']:
        return False
    if any(x >= R for x in public_inputs):
        return False
    pr = parse_proof(proof_words)
    if any(pr[k] >= R for k in ('l', 'r', 'o', 's1', 's2', 'zu')) or any(v >= R for v in pr['qcp']):
        return False
    n_c = len(vk['qcp'])
    try:
        lro = [_g1(p) for p in pr['lro']]; hq = [_g1(p) for p in pr['h']]; z = _g1(pr['z'])
        hz, hzw = _g1(pr['h_zeta']), _g1(pr['h_zeta_omega'])
        bsb = [_g1(p) for p in pr['bsb'][:n_c]]
        key = {k: _g1(vk[k]) for k in ('s1', 's2', 's3', 'ql', 'qr', 'qm', 'qo', 'qk')}
        qcp_pts = [_g1(q) for q in vk['qcp']]
    except ValueError:
        return False
    if not (m.g2_on_curve(vk['g2']) and m.g2_in_subgroup(vk['g2']) and m.g2_on_curve(vk['g2_tau']) and m.g2_in_subgroup(vk['g2_tau'])):
        return False
    # ---- challenges
    fs = Transcript('gamma', 'beta', 'alpha', 'zeta')
    for k in ('s1', 's2', 's3', 'ql', 'qr', 'qm', 'qo', 'qk'):
        fs.add('gamma', g1_bytes(key[k]))
    for q in qcp_pts:
        fs.add('gamma', g1_bytes(q))
    for x in public_inputs:
        fs.add('gamma', be32(x))
    for p in lro:
        fs.add('gamma', g1_bytes(p))
    gamma = int.from_bytes(fs.challenge('gamma'), 'big') % R
    beta = int.from_bytes(fs.challenge('beta'), 'big') % R
    for p in bsb:
        fs.add('alpha', g1_bytes(p))
    fs.add('alpha', g1_bytes(z))
    alpha = int.from_bytes(fs.challenge('alpha'), 'big') % R
    for p in hq:
        fs.add('zeta', g1_bytes(p))
    zeta = int.from_bytes(fs.challenge('zeta'), 'big') % R
    # ---- public-input polynomial at zeta
    n, w, n_inv = vk['size'], vk['generator'], vk['size_inv']
    zeta_n = pow(zeta, n, R)
    zh = (zeta_n - 1) % R
    if (zeta - 1) % R == 0:
        return False                                   # 1/(zeta - 1): gnark's Inverse(0) = 0 makes the relation fail; treated as failure
    lagrange0 = zh * finv(zeta - 1) % R * n_inv % R
    pi = 0
    acc = 1
    for x in public_inputs:
        den = (zeta - acc) % R
        if den == 0:
            return False
        pi = (pi + zh * finv(den) % R * n_inv % R * acc % R * x) % R
        acc = acc * w % R
    for i in range(n_c):
        hashed = hash_to_field_bsb22(g1_bytes(bsb[i]))
        wi = pow(w, vk['nb_public'] + vk['cci'][i], R)
        den = (zeta - wi) % R
        if den == 0:
            return False
        pi = (pi + zh * wi % R * finv(den) % R * n_inv % R * hashed) % R
    l, r, o, s1, s2, zu = pr['l'], pr['r'], pr['o'], pr['s1'], pr['s2'], pr['zu']
    a2l0 = lagrange0 * alpha % R * alpha % R
    t1 = (l + beta * s1 + gamma) % R
    t2 = (r + beta * s2 + gamma) % R
    lin_eval = -(pi - a2l0 + alpha * t1 % R * t2 % R * ((o + gamma) % R) % R * zu) % R           # opening of the linearised polynomial
    # ---- linearised polynomial digest
    u = vk['coset_shift']
    _s1 = alpha * t1 % R * t2 % R * beta % R * zu % R
    _s2 = -alpha * ((l + beta * zeta + gamma) % R) % R * ((r + beta * u % R * zeta + gamma) % R) % R * ((o + beta * u % R * u % R * zeta + gamma) % R) % R
    coeff_z = (a2l0 + _s2) % R
    zeta_n2 = pow(zeta, n + 2, R)
    terms = [(pr['qcp'][i], bsb[i]) for i in range(n_c)] + [
        (l, key['ql']), (r, key['qr']), (l * r % R, key['qm']), (o, key['qo']), (1, key['qk']), (_s1, key['s3']), (coeff_z, z),
        (-zh % R, hq[0]), (-zeta_n2 * zh % R, hq[1]), (-zeta_n2 * zeta_n2 % R * zh % R, hq[2])]
    lin = None
    for k, pt in terms:
        lin = m.g1_add(lin, m.g1_mul(pt, k))
    # ---- fold the openings at zeta (kzg.FoldProof)
    digests = [lin, lro[0], lro[1], lro[2], key['s1'], key['s2']] + qcp_pts
    values = [lin_eval, l, r, o, s1, s2] + pr['qcp'][:n_c]
    fk = Transcript('gamma')
    fk.add('gamma', be32(zeta))
    for d in digests:
        fk.add('gamma', g1_bytes(d))
    for v in values:
        fk.add('gamma', be32(v))
    fk.add('gamma', be32(zu))
    g_kzg = int.from_bytes(fk.challenge('gamma'), 'big') % R
    folded_digest, folded_eval, gi = None, 0, 1
    for d, v in zip(digests, values):
        folded_digest = m.g1_add(folded_digest, m.g1_mul(d, gi))
        folded_eval = (folded_eval + gi * v) % R
        gi = gi * g_kzg % R
    # ---- batch the two openings (kzg.BatchVerifyMultiPoints; lambda derived as the Solidity template does)
    lam = int.from_bytes(hashlib.sha256(g1_bytes(folded_digest) + g1_bytes(hz) + g1_bytes(z) + g1_bytes(hzw) + be32(zeta) + be32(g_kzg)).digest(), 'big') % R
    zeta_w = zeta * w % R
    quot = m.g1_add(hz, m.g1_mul(hzw, lam))
    dig = m.g1_add(folded_digest, m.g1_mul(z, lam))
    evals = (folded_eval + lam * zu) % R
    dig = m.g1_add(dig, m.g1_neg(m.g1_mul(m.G1_GEN, evals)))
    dig = m.g1_add(dig, m.g1_add(m.g1_mul(hz, zeta), m.g1_mul(hzw, lam * zeta_w % R)))
    pairs = []
    if dig is not None:
        pairs.append((dig, vk['g2']))
    if quot is not None:
        pairs.append((m.g1_neg(quot), vk['g2_tau']))
    return m.pairing_product_is_one(pairs)


def sp1_plonk_verify_proof(vk, verifier_hash, program_vkey, public_values, proof_bytes):
    """Status of `verify_proof` with the PLONK proof system behind it: the check order of sp1/verifier.rs:58-111."""
    if len(proof_bytes) < 4:
        return INVALID_PROOF_DATA, None
    if proof_bytes[:4] != verifier_hash[:4]:
        return SELECTOR_MISMATCH, proof_bytes[:4]
    if len(proof_bytes) != PROOF_BYTES:
        return INVALID_PROOF_DATA, None
    signals = [int.from_bytes(program_vkey, 'big'), m.sp1_hash_public_values(public_values)]
    return (OK if plonk_verify(vk, proof_bytes[4:], signals) else VERIFICATION_FAILED), None


# ---------------------------------------------------------------- toy circuit, trapdoor key, prover (test tooling)
def _interp(vals, w, n):
    """Coefficients of the polynomial with p(w^i) = vals[i] (naive inverse DFT)."""
    n_inv, w_inv = finv(n), finv(w)
    return [sum(vals[i] * pow(w_inv, i * j, R) for i in range(n)) % R * n_inv % R for j in range(n)]


def _pmul(a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % R
    return out


def _padd(a, b, kb=1):
    out = [0] * max(len(a), len(b))
    for i, x in enumerate(a):
        out[i] = x
    for i, y in enumerate(b):
        out[i] = (out[i] + kb * y) % R
    return out


def _peval(a, x):
    acc = 0
    for c in reversed(a):
        acc = (acc * x + c) % R
    return acc


def _pscale_arg(a, k):          # p(kX)
    return [c * pow(k, i, R) % R for i, c in enumerate(a)]


class ToyCircuit:
    """n = 8 rows, two public inputs (a, b), one BSB22 commitment.
       row 0: l = a (public)          row 1: l = b (public)          row 2: l = c, the hash of the BSB22 commitment (public-input row nb_public + 0)
       row 3: l * r = o   with l = a, r = b, o = ab          row 4: l = committed value (qcp = 1, ql = -1): pi2[4] = l = ab
       row 5: l * r = o   with l = c, r = ab, o = c ab       rows 6, 7: empty."""
    N = 8

    def __init__(self, rng):
        n = self.N
        self.w = pow(5, (R - 1) // n, R)
        assert pow(self.w, n // 2, R) != 1 and pow(self.w, n, R) == 1
        self.u = 5
        self.tau = rng.randrange(2, R)
        z = [0] * n
        self.ql, self.qr, self.qm, self.qo, self.qk, self.qcp = (list(z) for _ in range(6))
        self.ql[0] = self.ql[1] = self.ql[2] = self.ql[4] = R - 1
        self.qm[3] = self.qm[5] = 1
        self.qo[3] = self.qo[5] = R - 1
        self.qcp[4] = 1
        # copy cycles over the 3n wire slots (l: 0..n-1, r: n..2n-1, o: 2n..3n-1)
        L, Rr, O = (lambda i: i), (lambda i: n + i), (lambda i: 2 * n + i)
        cycles = [[L(0), L(3)], [L(1), Rr(3)], [O(3), L(4), Rr(5)], [L(2), L(5)]]
        self.sigma = list(range(3 * n))
        for c in cycles:
            for i, s in enumerate(c):
                self.sigma[s] = c[(i + 1) % len(c)]
        ids = [pow(self.w, i, R) * k % R for k in (1, self.u, self.u * self.u % R) for i in range(n)]
        self.ids = ids
        self.s = [[ids[self.sigma[j * n + i]] for i in range(n)] for j in range(3)]
        self.polys = {k: _interp(getattr(self, k), self.w, n) for k in ('ql', 'qr', 'qm', 'qo', 'qk', 'qcp')}
        for j in range(3):
            self.polys['s%d' % (j + 1)] = _interp(self.s[j], self.w, n)
        com = lambda p: m.g1_mul(m.G1_GEN, _peval(p, self.tau))
        self.vk = dict(size=n, size_inv=finv(n), generator=self.w, coset_shift=self.u, nb_public=2, cci=[0],
                       qcp=[g1_wire(com(self.polys['qcp']))], g2=m.G2_GEN, g2_tau=m.g2_mul(m.G2_GEN, self.tau),
                       **{k: g1_wire(com(self.polys[k])) for k in ('s1', 's2', 's3', 'ql', 'qr', 'qm', 'qo', 'qk')})

    def prove(self, a, b, tamper=None):
        """27 proof words (bytes) for public inputs (a, b).  `tamper`: name of a witness slot to corrupt (proof must then fail)."""
        n, w, tau, vk = self.N, self.w, self.tau, self.vk
        com = lambda p: m.g1_mul(m.G1_GEN, _peval(p, tau))
        ab = a * b % R
        pi2_vals = [0] * n; pi2_vals[4] = ab
        pi2 = _interp(pi2_vals, w, n)
        c_pi2 = com(pi2)
        c = hash_to_field_bsb22(g1_bytes(c_pi2))
        lv = [a, b, c, a, ab, c, 0, 0]; rv = [0, 0, 0, b, 0, ab, 0, 0]; ov = [0, 0, 0, ab, 0, c * ab % R, 0, 0]
        if tamper == 'o5':
            ov[5] = (ov[5] + 1) % R
        lp, rp, op = (_interp(v, w, n) for v in (lv, rv, ov))
        c_l, c_r, c_o = com(lp), com(rp), com(op)
        fs = Transcript('gamma', 'beta', 'alpha', 'zeta')
        for k in ('s1', 's2', 's3', 'ql', 'qr', 'qm', 'qo', 'qk'):
            fs.add('gamma', g1_bytes(vk[k]))
        fs.add('gamma', g1_bytes(vk['qcp'][0]))
        fs.add('gamma', be32(a)); fs.add('gamma', be32(b))
        for p in (c_l, c_r, c_o):
            fs.add('gamma', g1_bytes(p))
        gamma = int.from_bytes(fs.challenge('gamma'), 'big') % R
        beta = int.from_bytes(fs.challenge('beta'), 'big') % R
        # grand product
        wires = lv + rv + ov
        zv = [1] * n
        for i in range(n - 1):
            num = den = 1
            for j in range(3):
                num = num * ((wires[j * n + i] + beta * self.ids[j * n + i] + gamma) % R) % R
                den = den * ((wires[j * n + i] + beta * self.s[j][i] + gamma) % R) % R
            zv[i + 1] = zv[i] * num % R * finv(den) % R
        zp = _interp(zv, w, n)
        c_z = com(zp)
        fs.add('alpha', g1_bytes(c_pi2)); fs.add('alpha', g1_bytes(c_z))
        alpha = int.from_bytes(fs.challenge('alpha'), 'big') % R
        # quotient
        pi_vals = [a, b, c] + [0] * (n - 3)
        pip = _interp(pi_vals, w, n)
        q = self.polys
        gate = _padd(_padd(_padd(_padd(_padd(_pmul(q['ql'], lp), _pmul(q['qr'], rp)), _pmul(q['qm'], _pmul(lp, rp))), _pmul(q['qo'], op)),
                           _padd(q['qk'], pip)), _pmul(q['qcp'], pi2))
        bg = lambda poly, spoly: _padd(_padd(poly, [gamma]), spoly, beta)
        perm1 = _pmul(_pmul(_pmul(bg(lp, q['s1']), bg(rp, q['s2'])), bg(op, q['s3'])), _pscale_arg(zp, w))
        perm2 = _pmul(_pmul(_pmul(bg(lp, [0, 1]), bg(rp, [0, self.u])), bg(op, [0, self.u * self.u % R])), zp)
        l1 = _interp([1] + [0] * (n - 1), w, n)
        num = _padd(_padd(gate, _padd(perm1, perm2, R - 1), alpha), _pmul(l1, _padd(zp, [R - 1])), alpha * alpha % R)
        # divide by X^n - 1
        num = num + [0] * max(0, 3 * (n + 2) + n - len(num))
        h = [0] * (len(num) - n)
        rem = list(num)
        for i in range(len(num) - 1, n - 1, -1):
            h[i - n] = rem[i]
            rem[i - n] = (rem[i - n] + rem[i]) % R
            rem[i] = 0
        if tamper is None:
            assert not any(rem), 'the toy witness does not satisfy the circuit'
        hs = [h[k * (n + 2):(k + 1) * (n + 2)] for k in range(3)]
        c_h = [com(x) for x in hs]
        for p in c_h:
            fs.add('zeta', g1_bytes(p))
        zeta = int.from_bytes(fs.challenge('zeta'), 'big') % R
        ev = lambda poly, x: _peval(poly, x)
        l, r, o, s1, s2, zu, qcpz = ev(lp, zeta), ev(rp, zeta), ev(op, zeta), ev(q['s1'], zeta), ev(q['s2'], zeta), ev(zp, zeta * w % R), ev(q['qcp'], zeta)
        # the verifier's linearised polynomial, as a polynomial (same scalars as plonk_verify)
        zeta_n = pow(zeta, n, R); zhz = (zeta_n - 1) % R
        lag0 = zhz * finv(zeta - 1) % R * finv(n) % R
        a2l0 = lag0 * alpha % R * alpha % R
        t1 = (l + beta * s1 + gamma) % R; t2 = (r + beta * s2 + gamma) % R
        _s1 = alpha * t1 % R * t2 % R * beta % R * zu % R
        _s2 = -alpha * ((l + beta * zeta + gamma) % R) % R * ((r + beta * self.u % R * zeta + gamma) % R) % R * ((o + beta * self.u % R * self.u % R * zeta + gamma) % R) % R
        zn2 = pow(zeta, n + 2, R)
        lin = [0]
        for k, poly in ((qcpz, pi2), (l, q['ql']), (r, q['qr']), (l * r % R, q['qm']), (o, q['qo']), (1, q['qk']), (_s1, q['s3']), ((a2l0 + _s2) % R, zp),
                        (-zhz % R, hs[0]), (-zn2 * zhz % R, hs[1]), (-zn2 * zn2 % R * zhz % R, hs[2])):
            lin = _padd(lin, poly, k % R)
        lin_eval = ev(lin, zeta)
        c_lin = com(lin)
        digests = [c_lin, c_l, c_r, c_o, vk['s1'], vk['s2'], vk['qcp'][0]]
        polys = [lin, lp, rp, op, q['s1'], q['s2'], q['qcp']]
        values = [lin_eval, l, r, o, s1, s2, qcpz]
        fk = Transcript('gamma')
        fk.add('gamma', be32(zeta))
        for d in digests:
            fk.add('gamma', g1_bytes(d))
        for v in values:
            fk.add('gamma', be32(v))
        fk.add('gamma', be32(zu))
        g_kzg = int.from_bytes(fk.challenge('gamma'), 'big') % R
        f_tau = f_zeta = 0; gi = 1
        for poly, v in zip(polys, values):
            f_tau = (f_tau + gi * ev(poly, tau)) % R; f_zeta = (f_zeta + gi * v) % R; gi = gi * g_kzg % R
        h_zeta = m.g1_mul(m.G1_GEN, (f_tau - f_zeta) * finv(tau - zeta) % R)
        h_zeta_w = m.g1_mul(m.G1_GEN, (ev(zp, tau) - zu) * finv(tau - zeta * w) % R)
        words = [c_l, c_r, c_o, c_h[0], c_h[1], c_h[2]]
        out = b''.join(g1_bytes(p) for p in words) + b''.join(be32(v) for v in (l, r, o, s1, s2)) + g1_bytes(c_z) + be32(zu) + \
            g1_bytes(h_zeta) + g1_bytes(h_zeta_w) + be32(qcpz) + g1_bytes(c_pi2)
        assert len(out) == 32 * PROOF_WORDS
        return out
