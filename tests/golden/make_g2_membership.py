#!/usr/bin/env python3
"""Regenerates tests/golden/g2_membership_points.json: on-twist points for the G2 subgroup tests, with the verdict of the literal
EIP-197 test [r]Q = O from the slow spec model (oracle/spec_model.py).   Run:  python tests/golden/make_g2_membership.py

Kinds: points of G2; random twist points (the cofactor 2p - r is huge: outside); points of exact small order (10069, 5864401, their
product: the small prime factors of the cofactor), i.e. the points most likely to drive incomplete addition formulas into an
exceptional case; G2 points plus a small-order component; cofactor-only points [r]S.  "parity unpinned": nothing in the reference
holds G2 points outside the subgroup."""
import json, os, random, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..', '..', 'oracle'))
import spec_model as m  # noqa: E402
from make_golden import random_twist_point, h32  # noqa: E402


def words(pt):
    (xr, xi), (yr, yi) = pt
    return [h32(xi), h32(xr), h32(yi), h32(yr)]          # EIP-197 wire order


def main():
    rng = random.Random(20261004)
    h2 = 2 * m.P - m.R
    n2 = m.R * h2
    assert h2 % 10069 == 0 and h2 % 5864401 == 0
    gen = m.G2_GEN
    out = []

    def add(kind, pt):
        assert pt is not None and m.g2_on_curve(pt)
        out.append(dict(kind=kind, point=words(pt), in_subgroup=bool(m.g2_in_subgroup(pt))))

    def of_order(ell):
        while True:
            s = m.g2_mul(random_twist_point(rng), n2 // ell)
            if s is not None:
                assert m.g2_mul(s, ell) is None
                return s

    for _ in range(6):
        add('G2', m.g2_mul(gen, rng.randrange(1, m.R)))
    for k in (1, 2, m.R - 1):
        add('G2 small multiple of the generator', m.g2_mul(gen, k))
    for _ in range(6):
        add('random twist point', random_twist_point(rng))
    for ell in (10069, 5864401, 10069 * 5864401):
        s = of_order(ell)
        for k in (1, 2, 3, ell - 1, rng.randrange(1, ell)):
            if k % 10069 and k % 5864401:
                add('order %d' % ell, m.g2_mul(s, k))
        add('G2 + order %d' % ell, m.g2_add(m.g2_mul(gen, rng.randrange(1, m.R)), s))
    for _ in range(3):
        add('cofactor-only [r]S', m.g2_mul(random_twist_point(rng), m.R))
    assert sum(1 for c in out if c['in_subgroup']) == 9
    with open(os.path.join(HERE, 'g2_membership_points.json'), 'w') as f:
        json.dump(dict(note='spec-model verdicts, parity unpinned', points=out), f, indent=1)
    print(len(out), 'points')


if __name__ == '__main__':
    main()
