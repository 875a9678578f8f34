"""CPU check of the arithmetic behind the int8 digit-plane Gram (csrc/kernels_gram_i8.h), restated in NumPy with the kernel's own
steps -- scale exponent from the column maximum, balanced base-256 digits, exact int32 dot products with the integer resample
multiplicities, fp64 fma-chain recombination -- against EXACT rational sums (fractions.Fraction) and against the fp64 accumulation
the round-1 kernel performs.  Shows that the path is not a reduced-precision one: with 7 planes the result is the exact sum of the
fp64 products rounded ~once; with 6 it already matches a sequential fp64 accumulation."""
from fractions import Fraction

import numpy as np
import pytest

import plspm_oracle as orc


def digit_planes(z, S):
    """zs_max_kernel / zs_scale_kernel / zs_build_kernel: k = 8S - 1 - exponent(max |z|) (one less when the maximum fills its binade);
    z 2^k = sum_s d_s 256^s, d_s in [-128, 127]."""
    zmax = np.abs(z).max()
    f, e = np.frexp(zmax)
    k = 8 * S - 1 - int(e) - (1 if (f >= 0.99 or S >= 8) else 0)
    v = np.rint(np.ldexp(z, k)).astype(np.int64)
    planes = []
    for _ in range(S):
        d = ((v + 128) & 255) - 128
        planes.append(d.astype(np.int8))
        v = (v - d) >> 8
    assert not v.any()                                      # the top digit stayed inside int8
    return planes, k


def recombine(acc, k):
    """gram_i8_kernel epilogue: v = acc_0; v = fma(acc_s, 256^s, v); result v 2^-k."""
    v = np.float64(acc[0])
    for s in range(1, len(acc)):
        v = np.float64(acc[s]) * np.float64(1 << (8 * s)) + v      # both operands exact: one rounding, like the fma
    return float(np.ldexp(v, -k))


@pytest.mark.parametrize("pair", [(0, 0), (0, 1), (7, 52), (59, 59), (13, 60)])
def test_digit_plane_sum_is_the_exactly_rounded_weighted_moment(pair):
    X, _ = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
    Xa = np.concatenate((X - X.mean(axis=0), np.ones((10000, 1))), axis=1)         # mean-shifted columns + the ones column
    rng = np.random.default_rng(2)
    c = np.bincount(rng.integers(0, 10000, 10000), minlength=10000).astype(np.int64)     # resample multiplicities
    assert c.max() <= 127
    z = Xa[:, pair[0]] * Xa[:, pair[1]]                                              # the fp64 products the planes are cut from
    exact = sum((Fraction(float(a)) * int(b) for a, b in zip(z, c) if b), Fraction(0))
    scale = float(np.sqrt(np.dot(c, Xa[:, pair[0]] ** 2) * np.dot(c, Xa[:, pair[1]] ** 2)))
    seq = 0.0
    for a, b in zip(z, c):
        if b:
            seq += b * a                                                            # a rounding per term: the fp64 accumulation chain
    err_seq = abs(Fraction(seq) - exact) / Fraction(scale)
    errs = {}
    for S in (5, 6, 7, 8):
        planes, k = digit_planes(z, S)
        acc = [int(np.dot(c, d.astype(np.int64))) for d in planes]
        assert all(abs(a) < 2 ** 31 for a in acc)                                   # exact in the MFMA's int32 accumulators
        errs[S] = float(abs(Fraction(recombine(acc, k)) - exact) / Fraction(scale))
    assert errs[7] <= 2.3e-16 and errs[8] <= 2.3e-16, errs                          # within an ulp of the exact sum
    assert errs[7] <= float(err_seq) + 1e-18, (errs, float(err_seq))               # never worse than the fp64 chain
    assert errs[6] <= 2e-14 and errs[5] <= 5e-12, errs                              # 8 bits per plane


def test_digit_planes_cover_extreme_dynamic_range_and_signs():
    rng = np.random.default_rng(5)
    z = rng.standard_normal(4096) * np.exp(rng.uniform(-30, 30, 4096))             # 26 orders of magnitude in one column
    c = rng.integers(0, 6, 4096).astype(np.int64)
    exact = sum((Fraction(float(a)) * int(b) for a, b in zip(z, c) if b), Fraction(0))
    planes, k = digit_planes(z, 7)
    acc = [int(np.dot(c, d.astype(np.int64))) for d in planes]
    got = recombine(acc, k)
    # absolute error bounded by the plane resolution of the column maximum: sum(c) * 2^-k / 2, plus one rounding of the result
    bound = Fraction(int(c.sum())) * Fraction(1, 2 ** (k + 1)) + abs(exact) * Fraction(1, 2 ** 52)
    assert abs(Fraction(got) - exact) <= bound
