#!/usr/bin/env python3
"""Which integer vectors (a0, a1, a2, a3) make  a0 Q + a1 psi(Q) + a2 psi^2(Q) + a3 psi^3(Q) = O  a SUFFICIENT test for
Q in G2 (the order-r subgroup of the BN254 sextic twist E'(Fp2))?

psi = twist^-1 o Frobenius o twist satisfies chi(X) = X^2 - tX + p on E'.  For h(X) = sum a_i X^i there are integer polynomials
A, B with A h + B chi = Res(h, chi), so h(psi) Q = O implies Res(h, chi) Q = O: ord(Q) divides gcd(Res(h, chi), #E'(Fp2)).
#E'(Fp2) = r (2p - r), r is prime and does not divide 2p - r, so E'(Fp2)[r] = G2 and the test is exact when that gcd is r.

Both vectors the library uses pass:
  (u+1, u, u, -2u)    csrc/zkv_curve.h g2_in_subgroup (the test of the 16-lane path and the set-up kernels)
  (6u+2, 1, -1, 1)    the optimal-ate vector: the lane-pair Miller loop ends with T = (6u+2)Q + psi(Q) - psi^2(Q), and compares it
                      with -psi^3(Q)  (csrc/zkv_verify.h miller_loop_p / miller_point_closes)
"""
from math import gcd

U = 4965661367192848881
P = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
R = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1
T = 6 * U * U + 1
N2 = R * (2 * P - R)


def resultant_with_chi(a):
    """Res(h, chi) for chi = X^2 - T X + P: reduce h modulo chi to c0 + c1 X, then the norm c0^2 + c0 c1 T + c1^2 P."""
    c = list(a)
    while len(c) > 2:
        top = c.pop()
        c[-1] += top * T
        c[-2] -= top * P
    c0, c1 = c
    return c0 * c0 + c0 * c1 * T + c1 * c1 * P


def vector_is_exact(a):
    res = resultant_with_chi(a)
    relation = sum(ai * pow(P, i, R) for i, ai in enumerate(a)) % R == 0      # holds on G2, where psi acts as p
    return relation and res % R == 0 and gcd(res, N2) == R


VECTORS = {'classical (u+1, u, u, -2u)': [U + 1, U, U, -2 * U], 'optimal ate (6u+2, 1, -1, 1)': [6 * U + 2, 1, -1, 1]}

if __name__ == '__main__':
    assert (2 * P - R) % R != 0
    for name, a in VECTORS.items():
        print('%-30s exact: %s' % (name, vector_is_exact(a)))
