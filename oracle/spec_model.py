"""Slow, obviously-correct big-int model of the verify path.  TEST INFRASTRUCTURE ONLY.

This file is the *specification* leg of the three-way parity check
(spec model == C oracle == HIP kernels).  It is never imported by the product
package; only tests/, tools that regenerate tests/golden/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may touch anything under oracle/.

What it restates (reference paths relative to /root/reference/contracts/src):

* common/groth16.rs:23-128   -- Groth16Verifier: signal range check, vk_x via
  ecMul/ecAdd "precompile calls", negate_g1 (wrapping subtract), 4-pair
  ecPairing call, `unwrap_or(false)` on any precompile error.
* risc0/verifier.rs:58-196, risc0/types.rs:44-94, risc0/crypto.rs:95-195,
  risc0/config.rs -- initialize / verify / verify_integrity, tagged digests,
  split_digest, selector derivation.
* sp1/verifier.rs:58-111, sp1/types.rs:22-38, sp1/config.rs -- verify_proof.
* The BN254 arithmetic itself is NOT in the reference (it STATICCALLs the EVM
  precompiles 0x06/0x07/0x08, groth16.rs:12-14).  It is restated here from the
  published EIP-196 / EIP-197 text: affine double-and-add, Fp12 as plain
  polynomials mod w^12 - 18 w^6 + 82, affine optimal-ate Miller loop, final
  exponentiation by the literal integer (p^12-1)/r.  Parity of those error paths
  is therefore "unpinned" beyond the two real proofs.
"""
import hashlib

# ---------------------------------------------------------------- parameters
U = 4965661367192848881
P = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1          # Q in groth16.rs:10
R = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1          # R in groth16.rs:9
assert P == 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
assert R == 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
ATE_LOOP = 6 * U + 2
B1 = 3

# status codes of the C ABI (include/zkv.h) -- order of evaluation is the reference's
OK, VERIFICATION_FAILED, INVALID_INITIALIZATION, ALREADY_INITIALIZED, INVALID_PROOF_DATA, SELECTOR_MISMATCH = range(6)
BAD_CALLDATA = 6     # wire layer only: calldata the contract's router cannot decode (reverts with empty data)


# ---------------------------------------------------------------- Fp2 (for curve arithmetic on the twist)
def f2add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2neg(a): return (-a[0] % P, -a[1] % P)
def f2mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2smul(a, k): return (a[0] * k % P, a[1] * k % P)
def f2inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * d % P, -a[1] * d % P)
def f2conj(a): return (a[0], -a[1] % P)

XI = (9, 1)
B2 = f2mul((3, 0), f2inv(XI))                  # twist: y^2 = x^3 + 3/(9+i)


# ---------------------------------------------------------------- Fp12 as polynomials in w, w^12 = 18 w^6 - 82
def f12mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for k in range(22, 11, -1):
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return [x % P for x in t[:12]]

F12_ONE = [1] + [0] * 11

def f12pow(a, e):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12mul(r, a)
        a = f12mul(a, a)
        e >>= 1
    return r

def f2_to_f12(a, k):
    """(a0 + a1 i) * w^k with i = w^6 - 9."""
    out = [0] * 12
    out[k] = (a[0] - 9 * a[1]) % P
    out[k + 6] = a[1] % P
    return out


# ---------------------------------------------------------------- groups (affine, None = infinity)
def g1_on_curve(pt):
    x, y = pt
    return (y * y - x * x * x - B1) % P == 0

def g1_add(p1, p2):
    if p1 is None: return p2
    if p2 is None: return p1
    x1, y1 = p1; x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)

def g1_mul(pt, k):
    acc = None
    while k:
        if k & 1:
            acc = g1_add(acc, pt)
        pt = g1_add(pt, pt)
        k >>= 1
    return acc

def g1_neg(pt): return None if pt is None else (pt[0], -pt[1] % P)

def g2_on_curve(pt):
    x, y = pt
    return f2sub(f2mul(y, y), f2add(f2mul(f2mul(x, x), x), B2)) == (0, 0)

def g2_add(p1, p2):
    if p1 is None: return p2
    if p2 is None: return p1
    x1, y1 = p1; x2, y2 = p2
    if x1 == x2:
        if f2add(y1, y2) == (0, 0):
            return None
        lam = f2mul(f2smul(f2mul(x1, x1), 3), f2inv(f2smul(y1, 2)))
    else:
        lam = f2mul(f2sub(y2, y1), f2inv(f2sub(x2, x1)))
    x3 = f2sub(f2sub(f2mul(lam, lam), x1), x2)
    return (x3, f2sub(f2mul(lam, f2sub(x1, x3)), y1))

def g2_mul(pt, k):
    acc = None
    while k:
        if k & 1:
            acc = g2_add(acc, pt)
        pt = g2_add(pt, pt)
        k >>= 1
    return acc

def g2_neg(pt): return None if pt is None else (pt[0], f2neg(pt[1]))

def g2_in_subgroup(pt):
    return g2_mul(pt, R) is None

def g2_frobenius(pt):
    """pi_p on the twist: (conj(x) * xi^((p-1)/3), conj(y) * xi^((p-1)/2))."""
    x, y = pt
    return (f2mul(f2conj(x), FROB_X), f2mul(f2conj(y), FROB_Y))

def f2pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2mul(r, a)
        a = f2mul(a, a)
        e >>= 1
    return r

FROB_X = f2pow(XI, (P - 1) // 3)
FROB_Y = f2pow(XI, (P - 1) // 2)


# ---------------------------------------------------------------- pairing (EIP-197 semantics)
def _line(T, Q2, Pt):
    """Line through untwisted T, Q2 (affine twist points) evaluated at G1 point Pt; returns (f12, T+Q2)."""
    xT, yT = T
    xQ, yQ = Q2
    xP, yP = Pt
    if xT == xQ and f2add(yT, yQ) == (0, 0):
        # vertical line: x_P - x_T w^2
        out = [0] * 12
        out[0] = xP
        v = f2_to_f12(f2neg(xT), 2)
        return [(a + b) % P for a, b in zip(out, v)], None
    if T == Q2:
        lam = f2mul(f2smul(f2mul(xT, xT), 3), f2inv(f2smul(yT, 2)))
    else:
        lam = f2mul(f2sub(yQ, yT), f2inv(f2sub(xQ, xT)))
    # l(P) = yP - lam*xP * w + (lam*xT - yT) * w^3
    out = [0] * 12
    out[0] = yP % P
    a = f2_to_f12(f2neg(f2smul(lam, xP)), 1)
    b = f2_to_f12(f2sub(f2mul(lam, xT), yT), 3)
    out = [(o + x + y) % P for o, x, y in zip(out, a, b)]
    x3 = f2sub(f2sub(f2mul(lam, lam), xT), xQ)
    y3 = f2sub(f2mul(lam, f2sub(xT, x3)), yT)
    return out, (x3, y3)

def miller_loop(Q2, Pt):
    if Q2 is None or Pt is None:
        return F12_ONE
    f = F12_ONE
    T = Q2
    for bit in bin(ATE_LOOP)[3:]:
        l, T2 = _line(T, T, Pt)
        f = f12mul(f12mul(f, f), l)
        T = T2
        if bit == '1':
            l, T2 = _line(T, Q2, Pt)
            f = f12mul(f, l)
            T = T2
    Q1 = g2_frobenius(Q2)
    nQ2 = g2_neg(g2_frobenius(Q1))
    l, T2 = _line(T, Q1, Pt)
    f = f12mul(f, l); T = T2
    l, T2 = _line(T, nQ2, Pt)
    f = f12mul(f, l)
    return f

FINAL_EXP = (P**12 - 1) // R

def final_exponentiate(f):
    return f12pow(f, FINAL_EXP)

def pairing_product_is_one(pairs):
    f = F12_ONE
    for g1, g2 in pairs:
        f = f12mul(f, miller_loop(g2, g1))
    return final_exponentiate(f) == F12_ONE


# ---------------------------------------------------------------- EVM precompile byte ABIs (EIP-196/197)
class PrecompileError(Exception):
    pass

def _rd_g1(buf):
    x = int.from_bytes(buf[0:32], 'big'); y = int.from_bytes(buf[32:64], 'big')
    if x >= P or y >= P:
        raise PrecompileError('coordinate >= Q')
    if x == 0 and y == 0:
        return None
    if not g1_on_curve((x, y)):
        raise PrecompileError('G1 not on curve')
    return (x, y)

def _rd_g2(buf):
    # EIP-197 wire order: (x_im, x_re, y_im, y_re)  -- see SURVEY a8
    xi = int.from_bytes(buf[0:32], 'big'); xr = int.from_bytes(buf[32:64], 'big')
    yi = int.from_bytes(buf[64:96], 'big'); yr = int.from_bytes(buf[96:128], 'big')
    if max(xi, xr, yi, yr) >= P:
        raise PrecompileError('coordinate >= Q')
    if xi == 0 and xr == 0 and yi == 0 and yr == 0:
        return None
    pt = ((xr, xi), (yr, yi))
    if not g2_on_curve(pt):
        raise PrecompileError('G2 not on twist')
    if not g2_in_subgroup(pt):
        raise PrecompileError('G2 not in subgroup')
    return pt

def _wr_g1(pt):
    if pt is None:
        return bytes(64)
    return pt[0].to_bytes(32, 'big') + pt[1].to_bytes(32, 'big')

def ecadd(data):
    data = data[:128].ljust(128, b'\0')
    return _wr_g1(g1_add(_rd_g1(data[0:64]), _rd_g1(data[64:128])))

def ecmul(data):
    data = data[:96].ljust(96, b'\0')
    pt = _rd_g1(data[0:64])
    k = int.from_bytes(data[64:96], 'big')
    return _wr_g1(g1_mul(pt, k))

def ecpairing(data):
    if len(data) % 192:
        raise PrecompileError('length')
    pairs = []
    for off in range(0, len(data), 192):
        g1 = _rd_g1(data[off:off + 64])
        g2 = _rd_g2(data[off + 64:off + 192])
        pairs.append((g1, g2))
    ok = pairing_product_is_one(pairs)
    return (1 if ok else 0).to_bytes(32, 'big')


# ---------------------------------------------------------------- common/groth16.rs
def be32(x): return int(x).to_bytes(32, 'big')

def negate_g1_words(x, y):
    """groth16.rs:75-84: (0,0) stays; otherwise y <- Q.wrapping_sub(y) (mod 2^256, no reduction)."""
    if x == 0 and y == 0:
        return x, y
    return x, (P - y) % (1 << 256)

def groth16_verify(vm_type, vk, a, b, c, signals):
    """groth16.rs:23-49.  vk = dict(alpha1=(x,y), beta2=((x0,x1),(y0,y1)), gamma2, delta2, ic=[(x,y)...]) raw words."""
    if len(signals) + 1 != len(vk['ic']) or any(s >= R for s in signals):
        return False
    try:
        vkx = vk['ic'][0]
        for s, ic in zip(signals, vk['ic'][1:]):
            m = ecmul(be32(ic[0]) + be32(ic[1]) + be32(s))
            r = ecadd(be32(vkx[0]) + be32(vkx[1]) + m)
            vkx = (int.from_bytes(r[:32], 'big'), int.from_bytes(r[32:], 'big'))
    except PrecompileError:
        return False
    a0 = negate_g1_words(*a) if vm_type == 'risc0' else tuple(a)
    g1s = [a0, vk['alpha1'], vkx, tuple(c)]
    g2s = [b, vk['beta2'], vk['gamma2'], vk['delta2']]
    data = b''
    for g1, g2 in zip(g1s, g2s):
        data += be32(g1[0]) + be32(g1[1]) + be32(g2[0][0]) + be32(g2[0][1]) + be32(g2[1][0]) + be32(g2[1][1])
    try:
        return int.from_bytes(ecpairing(data), 'big') != 0
    except PrecompileError:
        return False

def compute_vk_x(vk, signals):
    acc = vk['ic'][0]
    for s, ic in zip(signals, vk['ic'][1:]):
        acc = g1_add(acc, g1_mul(ic, s))
    return acc


# ---------------------------------------------------------------- verification keys (data from risc0/crypto.rs:16-79, sp1/crypto.rs:7-79)
RISC0_VK = dict(
    alpha1=(0x2D4D9AA7E302D9DF41749D5507949D05DBEA33FBB16C643B22F599A2BE6DF2E2,
            0x14BEDD503C37CEB061D8EC60209FE345CE89830A19230301F076CAFF004D1926),
    beta2=((0x0967032FCBF776D1AFC985F88877F182D38480A653F2DECAA9794CBC3BF3060C,
            0x0E187847AD4C798374D0D6732BF501847DD68BC0E071241E0213BC7FC13DB7AB),
           (0x304CFBD1E08A704A99F5E847D93F8C3CAAFDDEC46B7A0D379DA69A4D112346A7,
            0x1739C1B1A457A8C7313123D24D2F9192F896B7C63EEA05A9D57F06547AD0CEC8)),
    gamma2=((0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2,
             0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED),
            (0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B,
             0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA)),
    delta2=((0x03B03CD5EFFA95AC9BEE94F1F5EF907157BDA4812CCF0B4C91F42BB629F83A1C,
             0x1AA085FF28179A12D922DBA0547057CCAAE94B9D69CFAA4E60401FEA7F3E0333),
            (0x110C10134F200B19F6490846D518C9AEA868366EFB7228CA5C91D2940D030762,
             0x1E60F31FCBF757E837E867178318832D0B2D74D59E2FEA1C7142DF187D3FC6D3)),
    ic=[(0x12AC9A25DCD5E1A832A9061A082C15DD1D61AA9C4D553505739D0F5D65DC3BE4,
         0x025AA744581EBE7AD91731911C898569106FF5A2D30F3EEE2B23C60EE980ACD4),
        (0x0707B920BC978C02F292FAE2036E057BE54294114CCC3C8769D883F688A1423F,
         0x2E32A094B7589554F7BC357BF63481ACD2D55555C203383782A4650787FF6642),
        (0x0BCA36E2CBE6394B3E249751853F961511011C7148E336F4FD974644850FC347,
         0x2EDE7C9ACF48CF3A3729FA3D68714E2A8435D4FA6DB8F7F409C153B1FCDF9B8B),
        (0x1B8AF999DBFBB3927C091CC2AAF201E488CBACC3E2C6B6FB5A25F9112E04F2A7,
         0x2B91A26AA92E1B6F5722949F192A81C850D586D81A60157F3E9CF04F679CCCD6),
        (0x2B5F494ED674235B8AC1750BDFD5A7615F002D4A1DCEFEDDD06EDA5A076CCD0D,
         0x2FE520AD2020AAB9CBBA817FCBB9A863B8A76FF88F14F912C5E71665B2AD5E82),
        (0x0F1C3C0D5D9DA0FA03666843CDE4E82E869BA5252FCE3C25D5940320B1C4D493,
         0x214BFCFF74F425F6FE8C0D07B307482D8BC8BB2F3608F68287AA01BD0B69E809)],
)

SP1_VK = dict(
    alpha1=RISC0_VK['alpha1'],
    beta2=(RISC0_VK['beta2'][0],
           (0x001752A100A72FDF1E5A5D6EA841CC20EC838BCCFCF7BD559E79F1C9C759B6A0,
            0x192A8CC13CD9F762871F21E43451C6CA9EEAB2CB2987C4E366A185C25DAC2E7F)),
    gamma2=(RISC0_VK['gamma2'][0],
            (0x275DC4A288D1AFB3CBB1AC09187524C7DB36395DF7BE3B99E673B13A075A65EC,
             0x1D9BEFCD05A5323E6DA4D435F3B617CDB3AF83285C2DF711EF39C01571827F9D)),
    delta2=((0x1CC7CB8DE715675F21F01ECC9B46D236E0865E0CC020024521998269845F74E6,
             0x03FF41F4BA0C37FE2CAF27354D28E4B8F83D3B76777A63B327D736BFFB0122ED),
            (0x01909CD7827E0278E6B60843A4ABC7B111D7F8B2725CD5902A6B20DA7A2938FB,
             0x192BD3274441670227B4F69A44005B8711266E474227C6439CA25CA8E1EC1FC2)),
    ic=[(0x26091E1CAFB0AD8A4EA0A694CD3743EBF524779233DB734C451D28B58AA9758E,
         0x009FF50A6B8B11C3CA6FDB2690A124F8CE25489FEFA65A3E782E7BA70B66690E),
        (0x061C3FD0FD3DA25D2607C227D090CCA750ED36C6EC878755E537C1C48951FB4C,
         0x0FA17AE9C2033379DF7B5C65EFF0E107055E9A273E6119A212DD09EB51707219),
        (0x04EAB241388A79817FE0E0E2EAD0B2EC4FFDEC51A16028DEE020634FD129E71C,
         0x07236256D21C60D02F0BDBF95CFF83E03EA9E16FCA56B18D5544B0889A65C1F5)],
)


def vk_g2_point(words):
    """reference G2Point {x:[im,re], y:[im,re]} -> ((re,im),(re,im))."""
    (xi, xr), (yi, yr) = words
    return ((xr, xi), (yr, yi))


# ---------------------------------------------------------------- risc0 digests (risc0/types.rs, risc0/crypto.rs, risc0/config.rs)
def sha256(b): return hashlib.sha256(b).digest()

SYSTEM_STATE_ZERO_DIGEST = bytes.fromhex('a3acc27117418996340b84e5a90f3ef4c49d22c79e44aad822ec9c313e1eb8e2')

def output_digest(journal_digest, assumptions_digest=bytes(32)):
    return sha256(sha256(b'risc0.Output') + journal_digest + assumptions_digest + (2 << 8).to_bytes(2, 'big'))

def receipt_claim_ok_digest(image_id, journal_digest):
    out = output_digest(journal_digest)
    buf = (sha256(b'risc0.ReceiptClaim') + bytes(32) + image_id + SYSTEM_STATE_ZERO_DIGEST + out
           + (0 << 24).to_bytes(4, 'big') + (0 << 24).to_bytes(4, 'big') + (4 << 8).to_bytes(2, 'big'))
    return sha256(buf)

def split_digest(d):
    rev = d[::-1]
    return rev[16:], rev[:16]          # (low, high)

def tagged_struct(tag_digest, down):
    return sha256(tag_digest + b''.join(down) + ((len(down) << 8) & 0xffff).to_bytes(2, 'big'))

def tagged_list(tag_digest, items):
    cur = bytes(32)
    for e in reversed(items):
        cur = tagged_struct(tag_digest, [e, cur])
    return cur

def risc0_vk_digest(vk=RISC0_VK):
    ic = [sha256(be32(x) + be32(y)) for x, y in vk['ic']]
    al = sha256(be32(vk['alpha1'][0]) + be32(vk['alpha1'][1]))
    g2 = lambda q: sha256(be32(q[0][0]) + be32(q[0][1]) + be32(q[1][0]) + be32(q[1][1]))
    ic_list = tagged_list(sha256(b'risc0_groth16.VerifyingKey.IC'), ic)
    return sha256(sha256(b'risc0_groth16.VerifyingKey') + al + g2(vk['beta2']) + g2(vk['gamma2'])
                  + g2(vk['delta2']) + ic_list + (5 << 8).to_bytes(2, 'big'))

def risc0_selector(control_root, bn254_control_id):
    tag = sha256(b'risc0.Groth16ReceiptVerifierParameters')
    return sha256(tag + control_root + bn254_control_id[::-1] + risc0_vk_digest() + (3 << 8).to_bytes(2, 'big'))[:4]


class Risc0Verifier:
    """risc0/verifier.rs:44-196."""
    def __init__(self):
        self.initialized = False
        self.selector = bytes(4)
        self.control_root_0 = bytes(16)
        self.control_root_1 = bytes(16)
        self.bn254_control_id = bytes(32)

    def initialize(self, control_root, bn254_control_id):
        if self.initialized:
            return ALREADY_INITIALIZED
        lo, hi = split_digest(control_root)
        self.control_root_0, self.control_root_1 = lo, hi
        self.bn254_control_id = bn254_control_id
        self.selector = risc0_selector(control_root, bn254_control_id)
        self.initialized = True
        return OK

    def verify(self, seal, image_id, journal_digest):
        if not self.initialized:
            return INVALID_INITIALIZATION, None
        return self._verify_integrity_internal(seal, receipt_claim_ok_digest(image_id, journal_digest))

    def verify_integrity(self, seal, claim_digest):
        if not self.initialized:
            return INVALID_INITIALIZATION, None
        return self._verify_integrity_internal(seal, claim_digest)

    def signals(self, claim_digest):
        lo, hi = split_digest(claim_digest)
        return [int.from_bytes(self.control_root_0, 'big'), int.from_bytes(self.control_root_1, 'big'),
                int.from_bytes(lo, 'big'), int.from_bytes(hi, 'big'), int.from_bytes(self.bn254_control_id, 'big')]

    def _verify_integrity_internal(self, seal, claim_digest):
        if len(seal) < 4:
            return INVALID_PROOF_DATA, None
        recv = bytes(seal[:4])
        if recv != self.selector:
            return SELECTOR_MISMATCH, recv
        body = seal[4:]
        if len(body) != 256:            # strict abi_decode(validate=true) of 8 static words (unpinned, SURVEY 8a note)
            return INVALID_PROOF_DATA, None
        w = [int.from_bytes(body[32 * i:32 * i + 32], 'big') for i in range(8)]
        ok = groth16_verify('risc0', RISC0_VK, (w[0], w[1]), ((w[2], w[3]), (w[4], w[5])), (w[6], w[7]),
                            self.signals(claim_digest))
        return (OK if ok else VERIFICATION_FAILED), None


# ---------------------------------------------------------------- sp1 (sp1/verifier.rs, sp1/types.rs, sp1/config.rs)
SP1_VERIFIER_HASH = bytes.fromhex('a4594c59bbc142f3b81c3ecb7f50a7c34bc9af7c4c444b5d48b795427e285913')
SP1_VERSION = 'v5.0.0'
SP1_FIELD_MASK = (1 << 253) - 1

def sp1_hash_public_values(pv):
    return (int.from_bytes(sha256(pv), 'big') & SP1_FIELD_MASK) % R

def sp1_verify_proof(program_vkey, public_values, proof_bytes):
    if len(proof_bytes) < 4:
        return INVALID_PROOF_DATA, None
    recv = bytes(proof_bytes[:4])
    if recv != SP1_VERIFIER_HASH[:4]:
        return SELECTOR_MISMATCH, recv
    body = proof_bytes[4:]
    if len(body) != 256:
        return INVALID_PROOF_DATA, None
    w = [int.from_bytes(body[32 * i:32 * i + 32], 'big') for i in range(8)]
    signals = [int.from_bytes(program_vkey, 'big'), sp1_hash_public_values(public_values)]
    ok = groth16_verify('sp1', SP1_VK, (w[0], w[1]), ((w[2], w[3]), (w[4], w[5])), (w[6], w[7]), signals)
    return (OK if ok else VERIFICATION_FAILED), None


# ---------------------------------------------------------------- revert bytes (common/errors.rs, risc0/errors.rs, sp1/errors.rs)
ERROR_SELECTORS = {
    VERIFICATION_FAILED: bytes.fromhex('439cc0cd'),      # VerificationFailed()
    INVALID_INITIALIZATION: bytes.fromhex('f92ee8a9'),   # InvalidInitialization()
    ALREADY_INITIALIZED: bytes.fromhex('0dc149f0'),      # AlreadyInitialized()
    INVALID_PROOF_DATA: bytes.fromhex('e3e94326'),       # InvalidProofData()
}
RISC0_SELECTOR_MISMATCH = bytes.fromhex('b8b38d4c')      # SelectorMismatch(bytes4,bytes4)
SP1_WRONG_VERIFIER_SELECTOR = bytes.fromhex('988066a1')  # WrongVerifierSelector(bytes4,bytes4)

def revert_bytes(vm, status, received=None, expected=None):
    if status == OK:
        return b''
    if status == SELECTOR_MISMATCH:
        sel = RISC0_SELECTOR_MISMATCH if vm == 'risc0' else SP1_WRONG_VERIFIER_SELECTOR
        return sel + received.ljust(32, b'\0') + expected.ljust(32, b'\0')
    return ERROR_SELECTORS[status]


# ---------------------------------------------------------------- synthetic proofs: Groth16 re-randomisation
def rerandomize(a, b, c, delta2, r1, r2):
    """A' = r1^-1 A, B' = r1 B + r1 r2 delta, C' = C + r2 A   (a,c G1 affine; b,delta2 ((re,im),(re,im)))."""
    a2 = g1_mul(a, pow(r1, -1, R))
    b2 = g2_add(g2_mul(b, r1), g2_mul(delta2, r1 * r2 % R))
    c2 = g1_add(c, g1_mul(a, r2))
    return a2, b2, c2

def seal_bytes(selector, a, b, c):
    (bxr, bxi), (byr, byi) = b
    return selector + b''.join(be32(v) for v in (a[0], a[1], bxi, bxr, byi, byr, c[0], c[1]))


# ---------------------------------------------------------------- trapdoor keys: valid proofs for ARBITRARY public inputs (tests only)
G1_GEN = (1, 2)
G2_GEN = vk_g2_point(RISC0_VK['gamma2'])           # the canonical BN254 G2 generator


def g2_words(pt):
    """affine ((re,im),(re,im)) or None -> the reference's G2Point words (x_im, x_re), (y_im, y_re)."""
    if pt is None:
        return ((0, 0), (0, 0))
    (xr, xi), (yr, yi) = pt
    return ((xi, xr), (yi, yr))


def trapdoor_vk(rng, n_ic):
    """A verification key whose discrete logs are known: returns (vk dict in the reference's layout, trapdoor)."""
    td = dict(alpha=rng.randrange(1, R), beta=rng.randrange(1, R), gamma=rng.randrange(1, R), delta=rng.randrange(1, R),
              ic=[rng.randrange(1, R) for _ in range(n_ic)])
    vk = dict(alpha1=g1_mul(G1_GEN, td['alpha']), beta2=g2_words(g2_mul(G2_GEN, td['beta'])),
              gamma2=g2_words(g2_mul(G2_GEN, td['gamma'])), delta2=g2_words(g2_mul(G2_GEN, td['delta'])),
              ic=[g1_mul(G1_GEN, k) for k in td['ic']])
    return vk, td


def trapdoor_prove(rng, td, signals, vm_type):
    """(a, b, c) in the seal's word layout satisfying the pairing equation of groth16.rs:86-107 for `signals`."""
    ell = (td['ic'][0] + sum(s * k for s, k in zip(signals, td['ic'][1:]))) % R
    a_s, b_s = rng.randrange(1, R), rng.randrange(1, R)
    if vm_type == 'risc0':          # e(-A,B) e(alpha,beta) e(L,gamma) e(C,delta) = 1
        c_s = (a_s * b_s - td['alpha'] * td['beta'] - ell * td['gamma']) * pow(td['delta'], -1, R) % R
    else:                           # e(A,B) e(alpha,beta) e(L,gamma) e(C,delta) = 1
        c_s = -(a_s * b_s + td['alpha'] * td['beta'] + ell * td['gamma']) * pow(td['delta'], -1, R) % R
    A = g1_mul(G1_GEN, a_s); B = g2_words(g2_mul(G2_GEN, b_s)); Cc = g1_mul(G1_GEN, c_s)
    Cw = Cc if Cc is not None else (0, 0)
    return A, B, Cw


def vk_to_words(vk):
    """bytes in the order zkv_groth16_ctx_create / zkvo_groth16_verify_vk expect."""
    out = be32(vk['alpha1'][0]) + be32(vk['alpha1'][1])
    for q in (vk['beta2'], vk['gamma2'], vk['delta2']):
        out += be32(q[0][0]) + be32(q[0][1]) + be32(q[1][0]) + be32(q[1][1])
    for x, y in vk['ic']:
        out += be32(x) + be32(y)
    return out


def proof_to_words(a, b, c):
    return b''.join(be32(v) for v in (a[0], a[1], b[0][0], b[0][1], b[1][0], b[1][1], c[0], c[1]))


# ---------------------------------------------------------------- on-chain wire layer
# What a client of the deployed Stylus contracts sends and receives: `eth_call` calldata for the methods the
# example shells export (examples/risc0-verifier/src/lib.rs, examples/risc0-verifier/examples/interact.rs:31-43,
# examples/sp1-verifier/examples/interact.rs:11-19).  Stylus maps `Vec<u8>` to `uint8[]` (one 32-byte word per
# byte) and snake_case method names to camelCase.  UNPINNED: the Stylus router (stylus-sdk 0.9.0) is not in the
# container; modelled as "decode with alloy-sol-types 0.8.20 abi_decode_params(validate = true)", i.e. calldata
# must equal the canonical ABI encoding of what it decodes to, and any decode failure or unknown function
# selector reverts with empty data.

_KECCAK_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
              0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
              0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
              0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
              0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_KECCAK_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1

def _rol64(x, n): return ((x << n) | (x >> (64 - n))) & _M64 if n else x

def _keccak_f(s):
    for rc in _KECCAK_RC:
        c = [s[x][0] ^ s[x][1] ^ s[x][2] ^ s[x][3] ^ s[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol64(c[(x + 1) % 5], 1) for x in range(5)]
        s = [[s[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol64(s[x][y], _KECCAK_ROT[x][y])
        s = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        s[0][0] ^= rc
    return s

def keccak256(data):
    """Keccak-256 (original padding 0x01, as used by Ethereum), rate 136."""
    rate = 136
    msg = bytearray(data) + b'\x01' + bytes((-len(data) - 2) % rate) + b'\x80' if (len(data) + 1) % rate else bytearray(data) + b'\x81'
    s = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            s[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], 'little')
        s = _keccak_f(s)
    return b''.join(s[i % 5][i // 5].to_bytes(8, 'little') for i in range(4))

def fn_selector(signature): return keccak256(signature.encode())[:4]

RISC0_FUNCTIONS = ['initialize(bytes32,bytes32)', 'verify(uint8[],bytes32,bytes32)', 'verifyIntegrity(uint8[],bytes32)',
                   'isInitialized()', 'getSelector()', 'getControlRoot()', 'getBn254ControlId()', 'getVerifierKeyDigest()']
SP1_FUNCTIONS = ['verifyProof(bytes32,uint8[],uint8[])', 'verifierHash()', 'version()']

def _u8_array_words(b): return be32(len(b)) + b''.join(be32(x) for x in b)

def encode_risc0_verify(seal, image_id, journal_digest):
    return fn_selector(RISC0_FUNCTIONS[1]) + be32(0x60) + image_id + journal_digest + _u8_array_words(seal)

def encode_risc0_verify_integrity(seal, claim_digest):
    return fn_selector(RISC0_FUNCTIONS[2]) + be32(0x40) + claim_digest + _u8_array_words(seal)

def encode_risc0_initialize(control_root, bn254_control_id):
    return fn_selector(RISC0_FUNCTIONS[0]) + control_root + bn254_control_id

def encode_sp1_verify_proof(program_vkey, public_values, proof_bytes):
    return (fn_selector(SP1_FUNCTIONS[0]) + program_vkey + be32(0x60) + be32(0x60 + 32 + 32 * len(public_values)) +
            _u8_array_words(public_values) + _u8_array_words(proof_bytes))

def _decode_u8_array(args, head_word):
    """Lenient structural decode of one uint8[]; canonical form is enforced by the caller through re-encoding."""
    off = int.from_bytes(args[32 * head_word:32 * head_word + 32], 'big')
    if off + 32 > len(args):
        return None
    n = int.from_bytes(args[off:off + 32], 'big')
    if off + 32 + 32 * n > len(args):
        return None
    out = bytearray()
    for i in range(n):
        w = int.from_bytes(args[off + 32 + 32 * i:off + 64 + 32 * i], 'big')
        if w > 255:
            return None
        out.append(w)
    return bytes(out)

def _left(b): return bytes(b).ljust(32, b'\0')

def risc0_eth_call(verifier, calldata):
    """One eth_call against the RISC Zero shell: returns (reverted, returndata, status) with status = the verifier's
    status for verify / verifyIntegrity calls, BAD_CALLDATA for undecodable calldata, None for other methods."""
    if len(calldata) < 4:
        return True, b'', BAD_CALLDATA
    sel, args = bytes(calldata[:4]), bytes(calldata[4:])
    sels = [fn_selector(s) for s in RISC0_FUNCTIONS]
    if sel not in sels:
        return True, b'', BAD_CALLDATA
    k = sels.index(sel)
    if k in (1, 2):
        nhead = 3 if k == 1 else 2
        seal = _decode_u8_array(args, 0) if len(args) >= 32 * nhead else None
        if seal is None:
            return True, b'', BAD_CALLDATA
        a, b = args[32:64], args[64:96]
        canon = encode_risc0_verify(seal, a, b) if k == 1 else encode_risc0_verify_integrity(seal, a)
        if canon != bytes(calldata):
            return True, b'', BAD_CALLDATA
        st, recv = verifier.verify(seal, a, b) if k == 1 else verifier.verify_integrity(seal, a)
        if st == OK:
            return False, be32(1), st
        return True, revert_bytes('risc0', st, recv, verifier.selector), st
    if k == 0:
        if len(args) != 64:
            return True, b'', BAD_CALLDATA
        if verifier.initialized:
            return True, revert_bytes('risc0', ALREADY_INITIALIZED), None
        return False, b'', None                      # eth_call simulates; state is not kept
    if len(args) != 0:
        return True, b'', BAD_CALLDATA
    if k == 3: return False, be32(1 if verifier.initialized else 0), None
    if k == 4: return False, _left(verifier.selector), None
    if k == 5: return False, _left(verifier.control_root_0) + _left(verifier.control_root_1), None
    if k == 6: return False, bytes(verifier.bn254_control_id), None
    return False, risc0_vk_digest(), None

def sp1_eth_call(calldata):
    if len(calldata) < 4:
        return True, b'', BAD_CALLDATA
    sel, args = bytes(calldata[:4]), bytes(calldata[4:])
    sels = [fn_selector(s) for s in SP1_FUNCTIONS]
    if sel not in sels:
        return True, b'', BAD_CALLDATA
    k = sels.index(sel)
    if k == 0:
        if len(args) < 96:
            return True, b'', BAD_CALLDATA
        pv, proof = _decode_u8_array(args, 1), _decode_u8_array(args, 2)
        if pv is None or proof is None or encode_sp1_verify_proof(args[:32], pv, proof) != bytes(calldata):
            return True, b'', BAD_CALLDATA
        st, recv = sp1_verify_proof(args[:32], pv, proof)
        if st == OK:
            return False, b'', st
        return True, revert_bytes('sp1', st, recv, SP1_VERIFIER_HASH[:4]), st
    if len(args) != 0:
        return True, b'', BAD_CALLDATA
    if k == 1: return False, SP1_VERIFIER_HASH, None
    v = SP1_VERSION.encode()
    return False, be32(0x20) + be32(len(v)) + v.ljust(32, b'\0'), None
