"""Synthetic proof batches for benchmarks and size-independent parity tests.

Valid proofs cannot be forged for the real verification keys, so batches are Groth16 re-randomisations of a
real proof:  A' = r1^-1 A,  B' = r1 B + r1 r2 delta,  C' = C + r2 A  (public inputs unchanged), which yields
unlimited distinct valid 260-byte seals for the fixed VK.  A small pool is fully re-randomised (seeded
SplitMix64); the batch is then filled by walking r2 (B += delta, C += A per step), which costs two affine
additions per proof.  A configurable fraction is mutated into proofs that must NOT verify.

Pure Python big-int arithmetic, independent of both the HIP kernels and oracle/ (the generator is part of the
product tooling: the reference leaves proof generation as a TODO, examples/risc0-verifier/examples/interact.rs:110).
"""
import numpy as np

U = 4965661367192848881
P = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
R = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1

# delta2 of the two verification keys in the reference's word order (x_im, x_re, y_im, y_re):
# risc0/crypto.rs:43-52 and sp1/crypto.rs:48-59 (the SP1 key stores -delta).
RISC0_DELTA = (0x03B03CD5EFFA95AC9BEE94F1F5EF907157BDA4812CCF0B4C91F42BB629F83A1C, 0x1AA085FF28179A12D922DBA0547057CCAAE94B9D69CFAA4E60401FEA7F3E0333,
               0x110C10134F200B19F6490846D518C9AEA868366EFB7228CA5C91D2940D030762, 0x1E60F31FCBF757E837E867178318832D0B2D74D59E2FEA1C7142DF187D3FC6D3)
SP1_DELTA_NEG = (0x1CC7CB8DE715675F21F01ECC9B46D236E0865E0CC020024521998269845F74E6, 0x03FF41F4BA0C37FE2CAF27354D28E4B8F83D3B76777A63B327D736BFFB0122ED,
                 0x01909CD7827E0278E6B60843A4ABC7B111D7F8B2725CD5902A6B20DA7A2938FB, 0x192BD3274441670227B4F69A44005B8711266E474227C6439CA25CA8E1EC1FC2)

MUTATION_CLASSES = ('flip_c_x', 'flip_input', 'b_out_of_subgroup', 'coord_plus_q', 'wrong_selector')


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def below(self, n):
        v = 0
        for _ in range(5):
            v = (v << 64) | self.next()
        return v % n

    def scalar(self):
        return 1 + self.below(R - 1)


# ---------------------------------------------------------------- tiny affine BN254 arithmetic (None = infinity)
def _f2mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def _f2sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def _f2add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def _f2inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * d % P, -a[1] * d % P)


def g1_add(p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % P == 0:
            return None
        lam = 3 * p[0] * p[0] * pow(2 * p[1], -1, P) % P
    else:
        lam = (q[1] - p[1]) * pow(q[0] - p[0], -1, P) % P
    x = (lam * lam - p[0] - q[0]) % P
    return (x, (lam * (p[0] - x) - p[1]) % P)


def g1_mul(p, k):
    acc = None
    while k:
        if k & 1:
            acc = g1_add(acc, p)
        p = g1_add(p, p)
        k >>= 1
    return acc


def g2_add(p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if _f2add(p[1], q[1]) == (0, 0):
            return None
        x2 = _f2mul(p[0], p[0])
        lam = _f2mul(((3 * x2[0]) % P, (3 * x2[1]) % P), _f2inv(((2 * p[1][0]) % P, (2 * p[1][1]) % P)))
    else:
        lam = _f2mul(_f2sub(q[1], p[1]), _f2inv(_f2sub(q[0], p[0])))
    x = _f2sub(_f2sub(_f2mul(lam, lam), p[0]), q[0])
    return (x, _f2sub(_f2mul(lam, _f2sub(p[0], x)), p[1]))


def g2_mul(p, k):
    acc = None
    while k:
        if k & 1:
            acc = g2_add(acc, p)
        p = g2_add(p, p)
        k >>= 1
    return acc


def _f2sqrt(a):
    a0, a1 = a
    n = (a0 * a0 + a1 * a1) % P
    s = pow(n, (P + 1) // 4, P)
    if s * s % P != n:
        return None
    inv2 = pow(2, -1, P)
    for sg in (s, -s % P):
        t = (a0 + sg) * inv2 % P
        x0 = pow(t, (P + 1) // 4, P)
        if x0 and x0 * x0 % P == t:
            x1 = a1 * pow(2 * x0, -1, P) % P
            if _f2mul((x0, x1), (x0, x1)) == (a0 % P, a1 % P):
                return (x0, x1)
    return None


def random_twist_point(rng):
    """A point on the twist that is (with overwhelming probability) outside the order-r subgroup."""
    b2 = _f2mul((3, 0), _f2inv((9, 1)))
    while True:
        x = (rng.below(P), rng.below(P))
        y = _f2sqrt(_f2add(_f2mul(_f2mul(x, x), x), b2))
        if y is not None:
            return (x, y)


# ---------------------------------------------------------------- seal <-> points
def parse_seal(seal):
    w = [int.from_bytes(seal[4 + 32 * i:36 + 32 * i], 'big') for i in range(8)]
    return (w[0], w[1]), ((w[3], w[2]), (w[5], w[4])), (w[6], w[7])         # A, B ((re,im),(re,im)), C


def seal_words(a, b, c):
    (bxr, bxi), (byr, byi) = b
    return (a[0], a[1], bxi, bxr, byi, byr, c[0], c[1])


def _delta_point(words, negate):
    xi, xr, yi, yr = words
    pt = ((xr, xi), (yr, yi))
    return (pt[0], (-pt[1][0] % P, -pt[1][1] % P)) if negate else pt


def _write(out, row, selector, words):
    out[row, :4] = np.frombuffer(selector, dtype=np.uint8)
    out[row, 4:] = np.frombuffer(b''.join(int(v).to_bytes(32, 'big') for v in words), dtype=np.uint8)


def make_batch(vm, base_seal, n, seed, pool=16, mutate_every=64, classes=MUTATION_CLASSES):
    """Returns (seals uint8[n,260], mutated bool[n], mutation class index int8[n] (-1 = valid), flip_input bool[n]).

    vm: 'risc0' or 'sp1'.  base_seal: a real 260-byte proof for that VM.  `flip_input` marks proofs whose
    *public input* (journal digest / public values) must have one bit flipped by the caller.
    """
    rng = SplitMix64(seed)
    selector = bytes(base_seal[:4])
    a0, b0, c0 = parse_seal(base_seal)
    delta = _delta_point(RISC0_DELTA if vm == 'risc0' else SP1_DELTA_NEG, negate=(vm != 'risc0'))
    pool = max(1, min(pool, n))
    state = []
    for _ in range(pool):
        r1, r2 = rng.scalar(), rng.scalar()
        a = g1_mul(a0, pow(r1, -1, R))
        b = g2_add(g2_mul(b0, r1), g2_mul(delta, r1 * r2 % R))
        c = g1_add(c0, g1_mul(a0, r2))
        # walking r2 by one step for THIS pool entry: B += r1 * delta, C += r1 * A'  (A' = r1^-1 A  =>  r1 A' = A)
        state.append([a, b, c, g2_mul(delta, r1), a0])
    seals = np.zeros((n, 260), dtype=np.uint8)
    mutated = np.zeros(n, dtype=bool)
    mclass = np.full(n, -1, dtype=np.int8)
    flip_input = np.zeros(n, dtype=bool)
    oos = random_twist_point(rng) if 'b_out_of_subgroup' in classes else None
    for i in range(n):
        st = state[i % pool]
        a, b, c = st[0], st[1], st[2]
        words = list(seal_words(a, b, c))
        sel = selector
        if mutate_every and i % mutate_every == mutate_every - 1 and classes:
            k = rng.below(len(classes))
            name = classes[k]
            mutated[i] = True
            mclass[i] = MUTATION_CLASSES.index(name)
            if name == 'flip_c_x':
                words[6] ^= 1 << rng.below(250)
            elif name == 'flip_input':
                flip_input[i] = True
            elif name == 'b_out_of_subgroup':
                words[2:6] = [oos[0][1], oos[0][0], oos[1][1], oos[1][0]]
            elif name == 'coord_plus_q':
                j = rng.below(8)
                words[j] = words[j] + P
            elif name == 'wrong_selector':
                sel = bytes([selector[0] ^ 0x01]) + selector[1:]
        _write(seals, i, sel, words)
        st[1] = g2_add(b, st[3])
        st[2] = g1_add(c, st[4])
    return seals, mutated, mclass, flip_input


def _lane(args):
    vm, base_seal, n, seed, pool, mutate_every, classes = args
    return make_batch(vm, base_seal, n, seed, pool=pool, mutate_every=mutate_every, classes=classes)


def profiler_preloaded():
    """True when this process was started under a profiler whose preloaded library may already have initialised the GPU (rocprofv3
    --pmc does): forking worker processes from such a process would duplicate HIP-runtime threads into them."""
    import os
    pre = os.environ.get('LD_PRELOAD', '')
    return any(t in pre for t in ('rocprof', 'roctracer', 'rocprofiler')) or any(k.startswith(('ROCPROFILER_', 'ROCPROF_', 'ROCP_')) for k in os.environ)


def make_batch_parallel(vm, base_seal, n, seed, pool=16, mutate_every=64, classes=MUTATION_CLASSES, lanes=16, workers=None, cache=True):
    """make_batch for large n: the batch is the concatenation of `lanes` independent generator lanes (lane j = make_batch of its
    slice with seed + j), run in worker processes.  The result depends on (n, seed, lanes) only, never on the worker count.
    Workers are forked BEFORE the caller touches the GPU (pure Python big-int work; they never call into HIP).  Batches of
    2^16 proofs and more are kept in ZKV_SYNTH_CACHE (default /tmp/zkv_synth_cache) so that repeated runs on one box -- bench,
    then the rocprofv3 passes of the same command -- do not regenerate them."""
    import hashlib
    import os
    if n < 4096 or lanes <= 1:
        return make_batch(vm, base_seal, n, seed, pool=pool, mutate_every=mutate_every, classes=classes)
    cache_file = None
    if cache and n >= (1 << 16):
        d = os.environ.get('ZKV_SYNTH_CACHE', '/tmp/zkv_synth_cache')
        key = hashlib.sha256(repr((vm, bytes(base_seal).hex(), n, seed, pool, mutate_every, tuple(classes), lanes, 1)).encode()).hexdigest()[:24]
        cache_file = os.path.join(d, 'batch_%s.npz' % key)
        try:
            z = np.load(cache_file)                      # our own file: plain arrays, no pickle
            return z['seals'], z['mutated'], z['mclass'], z['flip']
        except (OSError, ValueError, KeyError):
            pass
    q, r = divmod(n, lanes)
    sizes = [q + (1 if j < r else 0) for j in range(lanes)]
    if mutate_every:                                     # keep every lane a multiple of the mutation period (same global pattern)
        per = ((n // lanes) // mutate_every) * mutate_every
        if per:
            sizes = [per] * (lanes - 1) + [n - per * (lanes - 1)]
    jobs = [(vm, bytes(base_seal), sz, seed + 0x9E3779B1 * j, pool, mutate_every, tuple(classes)) for j, sz in enumerate(sizes) if sz]
    if workers is None:
        share = max(1, int(os.environ.get('WORLD_SIZE', '1')))        # ranks of one job generate their shards side by side on one host
        workers = int(os.environ.get('ZKV_SYNTH_WORKERS', '0')) or min(len(jobs), max(1, len(os.sched_getaffinity(0)) // share))
        if profiler_preloaded():                         # cache miss under a profiler: generate in-process, never fork (slow but safe)
            workers = 1
    if workers <= 1:
        parts = [_lane(j) for j in jobs]
    else:
        import multiprocessing as mp
        with mp.get_context('fork').Pool(workers) as ex:
            parts = ex.map(_lane, jobs, chunksize=1)
    out = tuple(np.concatenate([p[k] for p in parts]) for k in range(4))
    if cache_file:
        try:
            os.makedirs(os.path.dirname(cache_file), exist_ok=True)
            tmp = cache_file + '.%d.tmp.npz' % os.getpid()
            np.savez(tmp, seals=out[0], mutated=out[1], mclass=out[2], flip=out[3])
            os.replace(tmp, cache_file)
        except OSError:
            pass
    return out


# ---------------------------------------------------------------- eth_call calldata of a batch (wire layer)
def _u8_array_words(dst, col0, data):
    """dst[:, col0:] receives the ABI encoding of uint8[] `data` (n x L): length word + one 32-byte word per byte."""
    n, length = data.shape
    dst[:, col0 + 28:col0 + 32] = np.frombuffer(int(length).to_bytes(4, 'big'), dtype=np.uint8)
    dst[:, col0 + 32 + 31:col0 + 32 + 32 * length:32] = data
    return col0 + 32 + 32 * length


def calldata_risc0_verify(seals, image_ids, journal_digests):
    """Canonical calldata of `verify(uint8[],bytes32,bytes32)` for every row: uint8[n, 8452] for 260-byte seals."""
    from . import wire
    n, length = seals.shape
    out = np.zeros((n, 4 + 96 + 32 * (length + 1)), dtype=np.uint8)
    out[:, :4] = np.frombuffer(wire.function_selector('verify(uint8[],bytes32,bytes32)'), dtype=np.uint8)
    out[:, 4 + 31] = 0x60
    out[:, 36:68] = image_ids
    out[:, 68:100] = journal_digests
    _u8_array_words(out, 100, seals)
    return out


def calldata_sp1_verify_proof(program_vkeys, public_values, proofs):
    """Canonical calldata of `verifyProof(bytes32,uint8[],uint8[])` for every row (fixed public-values length)."""
    from . import wire
    n, lpv = public_values.shape
    out = np.zeros((n, 4 + 96 + 32 * (lpv + 1) + 32 * (proofs.shape[1] + 1)), dtype=np.uint8)
    out[:, :4] = np.frombuffer(wire.function_selector('verifyProof(bytes32,uint8[],uint8[])'), dtype=np.uint8)
    out[:, 4:36] = program_vkeys
    out[:, 36 + 31] = 0x60
    out[:, 68 + 24:68 + 32] = np.frombuffer(int(0x80 + 32 * lpv).to_bytes(8, 'big'), dtype=np.uint8)
    k = _u8_array_words(out, 100, public_values)
    _u8_array_words(out, k, proofs)
    return out
