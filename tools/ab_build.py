#!/usr/bin/env python3
"""A/B builds of the library: recompiles the named translation units with extra -D flags and links them with the other objects of the
current build into stylus_zkvm_verifiers_amd/csrc/build/ab/libzkv_<tag>.so (git-ignored, travels with gpurun).
    python tools/ab_build.py tag1:k_pair:-DX=1,-DY tag2:k_pair:-DZ ...
Run a variant with ZKV_LIB_PATH=<that file> (stylus_zkvm_verifiers_amd/_lib.py)."""
import concurrent.futures as cf
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylus_zkvm_verifiers_amd import build as B


def one(spec):
    tag, units, flags = spec.split(':')
    units = units.split(',')
    flags = [f for f in flags.split(',') if f]
    out = os.path.join(B.BUILD, 'ab')
    os.makedirs(out, exist_ok=True)
    objs = []
    for u in B.UNITS:
        if u in units:
            o = os.path.join(out, '%s_%s.o' % (u, tag))
            subprocess.check_call([B._hipcc()] + B.FLAGS + flags + ['-c', os.path.join(B.CSRC, u + '.hip'), '-o', o], cwd=B.CSRC,
                                  stderr=open(os.path.join(out, '%s_%s.log' % (u, tag)), 'w'))
            objs.append(o)
        else:
            objs.append(os.path.join(B.BUILD, u + '.o'))
    lib = os.path.join(out, 'libzkv_%s.so' % tag)
    subprocess.check_call([B._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    return lib


if __name__ == '__main__':
    B.build(verbose=False)
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        for lib in ex.map(one, sys.argv[1:]):
            print(lib)
