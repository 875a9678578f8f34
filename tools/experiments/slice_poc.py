"""Proof of concept (CPU, numpy) of the int8-slice weighted Gram: M[p,q] = sum_i c_i x_ip x_iq with integer counts c and the
products z_i = x_ip x_iq cut into S balanced base-256 digits relative to the column's largest |z|.  Compares with an fp64
accumulation and with an exact (Fraction / longdouble) reference."""
import sys, numpy as np
sys.path.insert(0, "tools")
from synthetic import synth, satisfaction_C
from fractions import Fraction

def slices_of(z, S):
    zmax = np.abs(z).max()
    f, E = np.frexp(zmax)                      # zmax = f * 2^E, f in [0.5, 1)
    k = 8 * S - 1 - int(E) - (1 if (f >= 0.99 or S >= 8) else 0)      # |z| 2^k <= 127.5 x 256^(S-1)
    q = np.rint(np.ldexp(z, k)).astype(np.int64)
    digs = []
    for s in range(S):
        d = ((q + 128) & 255) - 128
        digs.append(d.astype(np.int8))
        q = (q - d) >> 8
    assert np.all(q == 0), "top digit overflow"
    return digs, k

X, _ = synth(10000, satisfaction_C(), 10, seed=0)
Xa = X - X.mean(0)
rng = np.random.default_rng(1)
idx = rng.integers(0, 10000, 10000)
c = np.bincount(idx, minlength=10000).astype(np.int64)
for (p, q) in [(0, 0), (0, 1), (3, 57), (59, 59)]:
    z = Xa[:, p] * Xa[:, q]
    exact = sum((Fraction(float(zz)) * int(cc) for zz, cc in zip(z, c) if cc), Fraction(0))
    f64 = float(np.dot(c.astype(np.float64), z))
    seq = 0.0
    for zz, cc in zip(z, c):
        if cc: seq = seq + cc * zz
    out = {}
    for S in (5, 6, 7, 8):
        digs, k = slices_of(z, S)
        acc = [int(np.dot(c, d.astype(np.int64))) for d in digs]
        assert all(abs(a) < 2**31 for a in acc)
        lo = sum(acc[s] << (8 * s) for s in range(min(4, S)))
        hi = sum(acc[s] << (8 * (s - 4)) for s in range(4, S))
        val = np.ldexp(float(hi) * 4294967296.0 + float(lo), -k)
        out[S] = abs(Fraction(val) - exact) / abs(exact)
    print((p, q), "exact %.17g" % float(exact), "| rel err: numpy dot %.2e, sequential fp64 %.2e" % (abs(Fraction(f64) - exact) / abs(exact), abs(Fraction(seq) - exact) / abs(exact)),
          "| slices", {S: "%.2e" % float(e) for S, e in out.items()})
