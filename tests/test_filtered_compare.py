"""The plain-groups kernel's FILTERED COMPARE (csrc/render_fast.hip, group block): the reference's hit test z < thr
(main/hmap.cpp:1013-1016, thr a double) is decided from F = (float)thr wherever

    d = z - F,  e = |F| * 2^-22 + 2^-148:    d < -e  =>  z < thr      d > e  =>  not (z < thr)

and only the samples in between load the double.  This checks the two implications -- in exact rational arithmetic for the
claim itself and in IEEE double arithmetic as the device evaluates it (numpy float64 / float32 conversions round like
v_cvt_f32_f64 / v_cvt_f64_f32; the fused multiply-add of e only makes e more accurate) -- over thresholds and positions chosen to
sit on and next to every edge: float ulp boundaries, binade boundaries, float subnormals, the float range's end, infinities, NaN."""
import math
from fractions import Fraction

import numpy as np


def _decide(z, thr):
    """-> (sure_hit, sure_miss) as the kernel computes them (arrays of float64)."""
    with np.errstate(all="ignore"):
        F = thr.astype(np.float32).astype(np.float64)
        d = z - F
        e = np.abs(F) * 2.0 ** -22 + 2.0 ** -148
        sure_hit = d < -e
        unsure = ~(np.abs(d) > e)
        return sure_hit, ~sure_hit & ~unsure, F


@np.errstate(all="ignore")
def _cases(rng, n):
    # thresholds: heights of every scale, float-exact values and their double neighbours, float subnormals, beyond the float range
    mant = rng.uniform(1.0, 2.0, n)
    expo = rng.choice(np.concatenate([np.arange(-160, 140), np.full(300, 5), np.arange(-12, 12).repeat(10)]), n)
    thr = np.ldexp(mant, expo) * rng.choice([1.0, 1.0, 1.0, -1.0], n)
    k = n // 8
    thr[:k] = thr[:k].astype(np.float32).astype(np.float64)                              # exactly a float
    thr[k:2 * k] = np.nextafter(thr[k:2 * k].astype(np.float32).astype(np.float64), np.inf)   # one double ulp off a float
    f32 = thr[2 * k:3 * k].astype(np.float32)
    thr[2 * k:3 * k] = (f32.astype(np.float64) + np.nextafter(f32, np.float32(np.inf)).astype(np.float64)) / 2  # a float tie
    thr[3 * k:3 * k + 6] = [np.inf, -np.inf, np.nan, 0.0, 3.5e38, -3.5e38]                     # (3.5e38 > FLT_MAX)
    thr[3 * k + 6:3 * k + 10] = [2.0 ** -149, 2.0 ** -150, 2.0 ** -126, 1e-320]
    # positions: within a few float ulps of the threshold (where the filter must give up), at its double neighbours, far off
    with np.errstate(all="ignore"):
        ulp32 = np.abs(thr) * 2.0 ** -23 + 2.0 ** -149
        z = thr + ulp32 * rng.choice([0.0, 0.49, -0.49, 0.5, -0.5, 1.0, -1.0, 3.9, -3.9, 4.1, -4.1, 17.0, -17.0, 1e6, -1e6], n) * rng.uniform(0.9, 1.1, n)
    z[::7] = np.nextafter(thr[::7], np.inf)
    z[1::7] = np.nextafter(thr[1::7], -np.inf)
    z[2::7] = thr[2::7]
    z[3::11] = rng.choice([np.inf, -np.inf, 0.0, 1e308, -1e308], len(z[3::11]))
    return z, thr


def test_filtered_compare_never_contradicts_the_double_compare():
    rng = np.random.RandomState(11)
    decided = total = 0
    for _ in range(40):
        z, thr = _cases(rng, 200000)
        with np.errstate(all="ignore"):
            want = z < thr
        hit, miss, _ = _decide(z, thr)
        assert not (hit & ~want).any(), (z[hit & ~want][:3], thr[hit & ~want][:3])
        assert not (miss & want).any(), (z[miss & want][:3], thr[miss & want][:3])
        ok = np.isfinite(thr) & (np.abs(thr) < 3e38)
        decided += int((hit | miss)[ok].sum())
        total += int(ok.sum())
    # positions more than ~5 float ulps from a threshold are decided without the double (the filter does filter)
    assert decided > 0.3 * total
    far = rng.uniform(-1000, 1000, 100000)
    thr = rng.uniform(0, 256, 100000)
    hit, miss, _ = _decide(far, thr)
    assert (hit | miss).mean() > 0.9999


def test_the_margin_in_rational_arithmetic():
    """|thr - F| <= half a float ulp <= |F| 2^-24 (normal floats) or 2^-150 (subnormal ones): e = |F| 2^-22 + 2^-148 covers
    it four times over, so the rounding of d = z - F and of e themselves (relative 2^-53 each) cannot matter."""
    rng = np.random.RandomState(3)
    mant = rng.uniform(1.0, 2.0, 20000)
    expo = rng.randint(-155, 127, 20000)
    for thr in np.ldexp(mant, expo):
        F = float(np.float32(thr))
        err = abs(Fraction(thr) - Fraction(F))
        e = Fraction(abs(F)) / 2 ** 22 + Fraction(1, 2 ** 148)
        assert 4 * err <= e, (thr, F)
        if F != 0.0 and math.isfinite(F):
            assert err <= max(Fraction(abs(F)) / 2 ** 24, Fraction(1, 2 ** 150))
