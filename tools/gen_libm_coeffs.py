#!/usr/bin/env python3
"""Derive the polynomial coefficients used by include/urf_libm.h.

The reference calls glibc's acosf/asinf/atan2f ("third-party
arithmetic on the path").  glibc's float versions are not correctly rounded and
differ between glibc releases, so they cannot be reproduced on the GPU.  This
project replaces them by ONE shared-source implementation that evaluates in
double and rounds once to float; host oracle and HIP kernels compile the same
header, so both sides are bit-identical by construction.

Two kernels are needed:

  asin:  asin(x) = x + x*u*P(u),           u = x*x   in [0, 0.25]
  atan:  atan(t) = t + t*v*Q(v),           v = t*t   in [0, (sqrt(2)-1)^2]

P and Q are near-minimax polynomials obtained by Chebyshev interpolation at 50
digits (mpmath) and then re-expanded in the monomial basis.  The script prints
C initialisers with 17 significant digits and the measured max relative error
of the double-precision Horner/fma evaluation against mpmath.

Run:  python tools/gen_libm_coeffs.py
"""
import mpmath as mp

mp.mp.dps = 60


def cheb_fit(f, a, b, deg):
    """Monomial coefficients (in the variable u on [a,b]) of the degree-`deg`
    Chebyshev interpolant of f."""
    n = deg + 1
    nodes = [mp.cos(mp.pi * (k + mp.mpf(1) / 2) / n) for k in range(n)]
    xs = [(a + b) / 2 + (b - a) / 2 * t for t in nodes]
    fx = [f(x) for x in xs]
    # Chebyshev coefficients
    c = []
    for j in range(n):
        s = mp.fsum(fx[k] * mp.cos(mp.pi * j * (k + mp.mpf(1) / 2) / n) for k in range(n))
        c.append(2 * s / n)
    c[0] /= 2
    # expand sum_j c_j T_j(t), t = (2u - (a+b))/(b-a), into monomials of u
    # polynomials represented as coefficient lists in u
    def padd(p, q):
        m = max(len(p), len(q))
        return [(p[i] if i < len(p) else 0) + (q[i] if i < len(q) else 0) for i in range(m)]

    def pmul(p, q):
        r = [mp.mpf(0)] * (len(p) + len(q) - 1)
        for i, pi in enumerate(p):
            for j, qj in enumerate(q):
                r[i + j] += pi * qj
        return r

    tpoly = [-(a + b) / (b - a), mp.mpf(2) / (b - a)]
    T0 = [mp.mpf(1)]
    T1 = tpoly
    acc = [c[0]]
    if n > 1:
        acc = padd(acc, [c[1] * v for v in T1])
    for j in range(2, n):
        T2 = padd(pmul([2 * v for v in tpoly], T1), [-v for v in T0])
        acc = padd(acc, [c[j] * v for v in T2])
        T0, T1 = T1, T2
    return acc


def P_asin(u):
    if u == 0:
        return mp.mpf(1) / 6
    s = mp.sqrt(u)
    return (mp.asin(s) - s) / (u * s)


def Q_atan(v):
    if v == 0:
        return -mp.mpf(1) / 3
    s = mp.sqrt(v)
    return (mp.atan(s) - s) / (v * s)


def horner_double(coefs, u):
    """Evaluate exactly as the C code does: Horner with fma in binary64."""
    import math
    r = coefs[-1]
    for c in reversed(coefs[:-1]):
        r = math.fma(r, u, c) if hasattr(math, "fma") else r * u + c
    return r


def report(name, f_true, coefs, a, b, full):
    cd = [float(c) for c in coefs]
    worst = mp.mpf(0)
    import random
    rnd = random.Random(1)
    for _ in range(20000):
        u = a + (b - a) * mp.mpf(rnd.random())
        ud = float(u)
        approx = mp.mpf(horner_double(cd, ud))
        # error measured on the full function value (x + x*u*P(u))
        x = mp.sqrt(mp.mpf(ud))
        val = x + x * mp.mpf(ud) * approx
        err = abs(val - full(x)) / abs(full(x)) if x != 0 else 0
        worst = max(worst, err)
    print("/* %s: degree %d, max rel err of x+x*u*P(u) in double = %s */" % (name, len(cd) - 1, mp.nstr(worst, 3)))
    for i, c in enumerate(cd):
        print("  %s, /* u^%d */" % (float(c).hex(), i))
    print("  decimal: " + ", ".join("%.17g" % c for c in cd))
    return cd


if __name__ == "__main__":
    a = mp.mpf(0)
    pa = cheb_fit(P_asin, a, mp.mpf("0.25"), 12)
    report("asin P(u), u in [0,0.25]", P_asin, pa, a, mp.mpf("0.25"), mp.asin)
    vmax = (mp.sqrt(2) - 1) ** 2
    qa = cheb_fit(Q_atan, a, vmax * mp.mpf("1.0001"), 10)
    report("atan Q(v), v in [0,(sqrt2-1)^2]", Q_atan, qa, a, vmax, mp.atan)
    print("pi    = %s" % float(mp.pi).hex(), "%.17g" % float(mp.pi))
    print("pi/2  = %s" % float(mp.pi / 2).hex(), "%.17g" % float(mp.pi / 2))
    print("pi/4  = %s" % float(mp.pi / 4).hex(), "%.17g" % float(mp.pi / 4))
    print("sqrt2-1 = %s" % float(mp.sqrt(2) - 1).hex(), "%.17g" % float(mp.sqrt(2) - 1))
