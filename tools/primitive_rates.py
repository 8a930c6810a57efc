#!/usr/bin/env python3
"""Rates of the library's register-resident primitives on this GPU (zkv_diag_mulmod_rate): operations per second over the chip and
the implied cost of one operation relative to fp_mul.  Used for the instruction-stream analysis in DESIGN.md."""
import ctypes as C
import json
import sys
sys.path.insert(0, '.')
from stylus_zkvm_verifiers_amd import _lib
L = _lib.lib()
names = {0: 'fp_mul', 1: 'f2_mul_lane (x2)', 2: 'fp_add / fp_sub, single chains', 3: 'fp_add_x2 / fp_sub_x2, interleaved pairs', 4: 'f2_mul_xi'}
out = {}
for kind in range(5):
    for w in (1, 2, 3):
        r = C.c_double(0); g = C.c_double(0)
        _lib.check(L.zkv_diag_mulmod_rate(0, kind, w, 4000 if kind < 2 else 20000, C.byref(r), C.byref(g)), 'diag')
        out['%s @%dw' % (names[kind], w)] = r.value
print(json.dumps(out, indent=1))
