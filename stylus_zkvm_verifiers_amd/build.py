"""Builds libzkv_mi355x.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m stylus_zkvm_verifiers_amd.build [--force]

One object per kernel translation unit, compiled in parallel, every compile bounded by a timeout; the
per-kernel register / scratch / LDS report of the compiler is kept next to the objects (build/*.log).
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
BUILD = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libzkv_mi355x.so')
UNITS = ['k_setup', 'k_prep', 'k_msm', 'k_pair', 'k_wide', 'k_precompile', 'k_wire', 'k_mixed', 'k_diag', 'k_plonk', 'k_agg', 'zkv_capi']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-DZKV_FP_MUL_NOINLINE',
         '-Rpass-analysis=kernel-resource-usage']
UNIT_FLAGS = {}          # per-unit extra flags (none at present)
COMPILE_TIMEOUT_S = 1500
EXTRA = os.environ.get('ZKV_EXTRA_FLAGS', '').split()


def _hipcc():
    for c in ('/opt/rocm/bin/hipcc', 'hipcc'):
        if os.path.sep not in c or os.path.exists(c):
            return c
    return 'hipcc'


def _deps_mtime():
    files = glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(HERE, '..', 'include', 'zkv.h')]
    return max(os.path.getmtime(f) for f in files)


def _compile(unit, force, dep_m):
    src = os.path.join(CSRC, unit + '.hip')
    obj = os.path.join(BUILD, unit + '.o')
    log = os.path.join(BUILD, unit + '.log')
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), dep_m):
        return unit, 'cached'
    cmd = [_hipcc()] + FLAGS + UNIT_FLAGS.get(unit, []) + EXTRA + ['-c', src, '-o', obj]
    with open(log, 'w') as lf:
        try:
            rc = subprocess.run(cmd, stdout=lf, stderr=subprocess.STDOUT, timeout=COMPILE_TIMEOUT_S, cwd=CSRC).returncode
        except subprocess.TimeoutExpired:
            rc = 124
    if rc != 0:
        tail = open(log).read()[-4000:]
        raise RuntimeError('hipcc failed for %s (rc=%d)\n%s' % (unit, rc, tail))
    return unit, 'built'


def build(force=False, verbose=True):
    os.makedirs(BUILD, exist_ok=True)
    dep_m = _deps_mtime()
    with cf.ThreadPoolExecutor(max_workers=min(7, os.cpu_count() or 4)) as ex:
        results = list(ex.map(lambda u: _compile(u, force, dep_m), UNITS))
    objs = [os.path.join(BUILD, u + '.o') for u in UNITS]
    need_link = force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)
    if need_link:
        subprocess.check_call([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    if verbose:
        print('libzkv_mi355x.so:', ', '.join('%s=%s' % r for r in results), '(linked)' if need_link else '(up to date)')
    return LIB


def resource_report():
    """Per-kernel VGPR / AGPR / scratch / LDS / occupancy as reported by hipcc (for DESIGN.md and profiles/)."""
    out = []
    for u in UNITS:
        log = os.path.join(BUILD, u + '.log')
        if not os.path.exists(log):
            continue
        cur = None
        for line in open(log):
            if 'remark:' not in line:
                continue
            body = line.split('remark:', 1)[1].split('[-Rpass')[0].strip()
            if body.startswith('Function Name:'):
                cur = {'kernel': body.split(':', 1)[1].strip()}
                out.append(cur)
            elif cur is not None and ':' in body:
                k, v = body.split(':', 1)
                cur[k.strip()] = v.strip()
    return out


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    for r in resource_report():
        print(r)
