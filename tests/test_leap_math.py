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
import pytest

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


@pytest.mark.parametrize("short", [2, 0])
def test_steps_left_count_is_a_safe_bound(short):
    """(short = 0: the count of HMRM_CROSS builds, as tight as the estimate allows.)  kStepsLeft (csrc/leap_common.hpp): after a refresh, every one of the next `left` positions -- the reference's
    sequential p += s -- equals p + j * delta bit for bit, stays inside the binade and off its boundary; so a jump of
    n <= left steps needs no test at its landing point.  Random and adversarial starts (close to either boundary, exact
    ties, reciprocal errors of either sign at the measured 2^-24); brute force against sequential stepping."""
    rng = random.Random(7)
    checked = long_runs = 0
    for trial in range(6000):
        e = rng.randint(-3, 11)
        lo = math.ldexp(1.0, e)
        kind = trial % 4
        if kind == 0:      # anywhere in the binade; now and then with a tiny step (10^6 .. 10^9 steps to the boundary)
            p = lo * (1.0 + rng.random())
            s = math.ldexp(rng.random() - 0.5, e - rng.randint(2, 9) - (rng.choice((20, 25, 30)) if trial % 16 == 0 else 0))
        elif kind == 1:    # a few steps before the upper boundary, moving up
            s = math.ldexp(rng.random() + 0.01, e - rng.randint(3, 10))
            p = 2 * lo - s * rng.uniform(0.0, 30.0)
        elif kind == 2:    # a few steps above the lower boundary, moving down
            s = -math.ldexp(rng.random() + 0.01, e - rng.randint(3, 10))
            p = lo - s * rng.uniform(0.0, 30.0)
        else:              # exact rounding ties
            m = rng.randint(2 ** 52 + 2 ** 30, 2 ** 53 - 2 ** 30)
            p = math.ldexp(float(m), e - 52)
            s = _tie_step(p, rng.randint(1, 2 ** 20), rng.choice((1.0, -1.0)))
        if not (lo <= p < 2 * lo):
            continue
        sign = rng.choice((1.0, -1.0))
        p, s = sign * p, sign * s
        a = L.Axis()
        L.axis_refresh(a, p, s, rcp_err=rng.choice((0.0, 2.0 ** -24, -2.0 ** -24, rng.uniform(-1, 1) * 2.0 ** -24)), short=short)
        if a.left < 0:
            continue
        q, key = p, L.hi32(p) >> 20
        for j in range(1, min(a.left, 3000) + 1):
            q = q + s
            assert bits(q) == bits(p + float(j) * a.delta), (p, s, j, a.left)
            assert (L.hi32(q) >> 20) == key and ((L.hi32(q) & 0xFFFFF) != 0 or L.lo32(q) != 0 or a.delta == 0.0), (p, s, j)
        if a.left <= 3000 and a.delta != 0.0:
            # ... and the count is tight: within a few steps of the true number that stay inside
            more = 0
            while (L.hi32(q + s) >> 20) == key and more < 10:
                q = q + s
                more += 1
            assert more <= 6, (p, s, a.left, more)
        elif a.delta != 0.0:
            # long stays (tiny steps): never given up, and short by no more than 2^-21 of the stay plus a few steps
            true_left = abs((a.lim - p) / a.delta)
            assert a.left > 0 and (a.left == 1 << 30 or true_left - a.left <= true_left * 2.0 ** -21 + 8), (p, s, a.left, true_left)  # (capped at 2^30)
        checked += 1
        long_runs += a.left > 3000
    assert checked > 3000 and long_runs > 100


def test_jumps_ending_in_a_real_step_cross_binades_exactly():
    """HMRM_CROSS (csrc/render_fast.hip): a jump is `left` multiplied steps and then one real step fl(p + s).  A ray
    that only ever moves that way -- refresh, jump to the binade's end, cross with the real step, refresh ... -- visits
    positions of the reference's sequence p += s, bit for bit, through many binades in either direction; and the
    real step after the counted ones does leave the binade nearly always (else the next refresh fails and a group
    of real steps is marched, as before)."""
    rng = random.Random(11)
    crossings = stuck = 0
    for trial in range(300):
        e = rng.randint(2, 9)
        p = math.ldexp(1.0 + rng.random(), e) * rng.choice((1.0, -1.0))
        s = math.ldexp(rng.random() + 0.05, e - rng.randint(5, 9)) * rng.choice((1.0, -1.0))
        if trial % 5 == 0:
            s = _tie_step(p, rng.randint(1, 2 ** 12), rng.choice((1.0, -1.0)))
        q = p            # the reference's sequence
        taken = 0
        a = L.Axis()
        while taken < 4000 and 2.0 ** -3 < abs(p) < 2.0 ** 12:
            L.axis_refresh(a, p, s, rcp_err=rng.uniform(-1, 1) * 2.0 ** -24, short=0)
            if a.left < 0:
                # (not exact here: the kernel marches real steps; a group of 4)
                n = 4
                for _ in range(n):
                    p = p + s
                stuck += 1
            else:
                mult = min(a.left, 700)                 # (any count up to `left` is a valid jump)
                n = mult + 1
                key = L.hi32(p) >> 20
                p = (p + float(mult) * a.delta) + s     # multiplied steps, then the real one
                crossings += (L.hi32(p) >> 20) != key
            for _ in range(n):
                q = q + s
            taken += n
            assert bits(p) == bits(q), (trial, taken)
    assert crossings > 500 and stuck < crossings // 2
