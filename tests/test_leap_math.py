"""CPU property tests of the arithmetic the production kernel's exact leaps rely on
(csrc/render_fast.hip, modelled in tests/leap_model.py):

  whenever axis_refresh accepts (p, s) and axis_landing_ok accepts p + n*delta, the
  value p + n*delta computed with ONE multiply and ONE add equals the reference's n
  sequential additions p += s (main/hmap.cpp:1037) bit for bit.

Brute force against sequential stepping, on random inputs and on the adversarial ones:
exact rounding ties, start values of either parity, binade boundaries in both
directions, absorbed steps, negative coordinates.
"""
import math
import random
import struct

import leap_model as L


def bits(v):
    return struct.unpack("<Q", struct.pack("<d", v))[0]


def check(p, s, n):
    """Returns (accepted, ok)."""
    a = L.Axis()
    L.axis_refresh(a, p, s)
    if a.key == 0xFFFFFFFF:
        return False, True
    pn = p + float(n) * a.delta
    if not L.axis_landing_ok(a, pn):
        return False, True
    return True, bits(pn) == bits(L.sequential(p, s, n))


def test_random_positions_and_steps():
    rng = random.Random(1)
    accepted = 0
    for _ in range(20000):
        e = rng.randint(-6, 12)
        p = math.ldexp(1.0 + rng.random(), e) * rng.choice((1.0, -1.0))
        s = math.ldexp(rng.random() - 0.5, rng.randint(-8, 1))
        n = rng.randint(1, 400)
        acc, ok = check(p, s, n)
        assert ok, (p, s, n)
        accepted += acc
    assert accepted > 5000


def _tie_step(p, q, sign):
    """A step that is an exact rounding tie in p's binade: s = sign * (q + 1/2) * u."""
    e = math.frexp(abs(p))[1] - 1
    u = math.ldexp(1.0, e - 52)
    return sign * (q + 0.5) * u


def test_exact_ties_both_parities():
    rng = random.Random(2)
    accepted = rejected = 0
    for _ in range(4000):
        e = rng.randint(0, 11)
        m = rng.randint(2 ** 52 + 2 ** 40, 2 ** 53 - 2 ** 40)   # well inside the binade
        p = math.ldexp(float(m), e - 52) * rng.choice((1.0, -1.0))
        q = rng.randint(1, 2 ** 30)
        s = _tie_step(p, q, rng.choice((1.0, -1.0)))
        n = rng.randint(1, 300)
        acc, ok = check(p, s, n)
        assert ok, (p, s, n, m & 1, q & 1)
        accepted += acc
        rejected += not acc
    # one start parity is steady, the other is rejected: both happen
    assert accepted > 1000 and rejected > 1000


def test_binade_boundaries():
    rng = random.Random(3)
    for _ in range(4000):
        e = rng.randint(-2, 11)
        lo = math.ldexp(1.0, e)
        s_mag = math.ldexp(rng.random() + 0.01, e - rng.randint(3, 12))
        k = rng.randint(0, 40)
        toward_zero = rng.random() < 0.5
        if toward_zero:
            p, s = lo + k * s_mag * rng.random(), -s_mag   # about to cross 2^e downwards
        else:
            p, s = 2 * lo - k * s_mag * rng.random(), s_mag  # about to cross 2^(e+1) upwards
        if not (lo <= p < 2 * lo):
            continue
        sign = rng.choice((1.0, -1.0))
        for n in (1, 2, 3, 5, 8, 13, 21, 34, 55):
            acc, ok = check(sign * p, sign * s, n)
            assert ok, (p, s, n, toward_zero)


def test_landing_exactly_on_lower_boundary_is_rejected():
    # p = 2^e + 3u, s = -1.25u: p + 2*delta may be computed as 2^e while the real sequence
    # rounds on the finer grid below 2^e -> such landings must not be trusted
    e = 5
    u = math.ldexp(1.0, e - 52)
    lo = math.ldexp(1.0, e)
    for frac in (1.25, 1.5, 0.75, 2.5):
        for start in range(1, 12):
            p, s = lo + start * u, -frac * u
            for n in range(1, 12):
                acc, ok = check(p, s, n)
                assert ok, (start, frac, n)


def test_absorbed_and_zero_steps():
    a = L.Axis()
    L.axis_refresh(a, 1000.0, 0.0)
    assert a.key != 0xFFFFFFFF and a.delta == 0.0 and L.axis_landing_ok(a, 1000.0)
    L.axis_refresh(a, 1024.0, 0.0)   # exactly on a boundary but not moving: still fine
    assert a.key != 0xFFFFFFFF and L.axis_landing_ok(a, 1024.0)
    tiny = math.ldexp(1.0, -60)
    L.axis_refresh(a, 1000.0, tiny)  # absorbed: p never changes, like the reference
    assert a.delta == 0.0 and L.sequential(1000.0, tiny, 50) == 1000.0
    for p in (0.0, 1e-300, float("inf"), float("nan")):
        L.axis_refresh(a, p, 0.25)
        assert a.key == 0xFFFFFFFF
