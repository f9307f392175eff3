"""Checks the divided-difference form of FaceReconWENO used by agx_device.hpp (weno_coeffs /
weno) against Shu's formula 2.20 as the reference evaluates it (utility.cpp:449-483), on
random widths: sub-stencil values and linear weights.  Run: python tools/weno_closed_form.py"""
import numpy as np
rng = np.random.default_rng(1)

def stencil_width(w, s, e):
    if e > s: return sum(w[s:e])
    if s > e: return -sum(w[e:s])
    return 0.0
def lagrange(w, degree, rr, ii):
    c = np.zeros(degree + 1)
    for jj in range(degree + 1):
        for mm in range(jj + 1, degree + 2):
            numer, denom = 0.0, 1.0
            for ll in range(degree + 2):
                if ll != mm:
                    prod = 1.0
                    for qq in range(degree + 2):
                        if qq != mm and qq != ll:
                            prod *= stencil_width(w, ii - rr + qq, ii + 1)
                    numer += prod
                    denom *= stencil_width(w, ii - rr + ll, ii - rr + mm)
            c[jj] += numer / denom
        c[jj] *= w[ii - rr + jj]
    return c
def ref(w, u):
    c0 = lagrange(w, 2, 2, 2); c1 = lagrange(w, 2, 1, 2); c2 = lagrange(w, 2, 0, 2)
    cf = lagrange(w, 4, 2, 2)
    s0 = c0 @ u[0:3]; s1 = c1 @ u[1:4]; s2 = c2 @ u[2:5]
    lw0 = cf[0] / c0[0]; lw1 = cf[4] / c2[2]; lw2 = 1 - lw0 - lw1
    return s0, s1, s2, lw0, lw1, lw2
def new(w, u):
    r = [1.0 / (w[a] + w[a + 1]) for a in range(4)]
    G = [(u[a + 1] - u[a]) * r[a] for a in range(4)]
    T = [1.0 / (w[m] + w[m + 1] + w[m + 2]) for m in range(3)]
    E = [G[m + 1] - G[m] for m in range(3)]
    s0 = u[2] + w[2] * G[1] + w[2] * (w[1] + w[2]) * T[0] * E[0]
    s1 = u[2] + w[2] * G[2] - w[2] * w[3] * T[1] * E[1]
    s2 = u[2] + w[2] * G[2] - w[2] * w[3] * T[2] * E[2]
    W4 = w[0] + w[1] + w[2] + w[3]; W5 = W4 + w[4]
    lw0 = w[3] * (w[3] + w[4]) / (W4 * W5)
    lw1 = (w[0] + w[1] + w[2]) * (w[1] + w[2]) / (W5 * (w[1] + w[2] + w[3] + w[4]))
    return s0, s1, s2, lw0, lw1, 1 - lw0 - lw1
worst = 0
for t in range(2000):
    w = rng.uniform(0.2, 3.0, 5); u = rng.normal(size=5) + 3
    a, b = np.array(ref(w, u)), np.array(new(w, u))
    worst = max(worst, np.abs(a - b).max() / np.abs(a).max())
print("worst rel diff", worst)
print(ref(np.ones(5), np.arange(5.0)**2)[3:])
