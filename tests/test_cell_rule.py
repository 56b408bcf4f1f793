"""CPU property tests of the general-grid-width cell rule of the production kernel (csrc/leap_common.hpp
cell_coord_fast<2>, csrc/render_fast.hip): the reference computes (int)((x - c0.x) / grid_width)
(main/hmap.cpp:1001-1004); the kernel computes q'' = fma(v, fl(1 / gw), 2^-20) and

  (1) off the neighbourhood of an integer -- fract(q'') >= 2^-19 -- trunc(q'') equals the reference's cell;
  (2) inside it -- q'' in [k, k + 2^-19) -- the reference's cell is k or k - 1 (what the landing test of a jump relies
      on: it accepts such a landing only if both lie inside the window; the start of an attempt and every sampled
      position divide for real).

Checked with exact rational arithmetic (fractions) for the fma and math.fsum-free correctly rounded division (Python's
float division IS correctly rounded), on random positions, on positions planted next to cell boundaries at every
distance from 2^-52 to 2^-18 cells, and for grid widths whose reciprocal rounds to an integer (0.05, 0.01, 0.2, 0.1)."""
import math
import random
from fractions import Fraction

GWS = (0.05, 0.01, 0.2, 0.1, 0.3, 3.0, 0.07, 1.7, 1e-3, 123.456)


def fma(a, b, c):
    """Correctly rounded a * b + c (what v_fma_f64 returns)."""
    exact = Fraction(a) * Fraction(b) + Fraction(c)
    f = float(exact)  # (int / int -> float conversion of a Fraction is correctly rounded)
    return f


def ref_cell(v, gw):
    q = v / gw  # correctly rounded, like the reference's fdiv
    return int(q)  # C truncation toward zero (|q| < 2^31 here)


def kernel(v, gw):
    inv = 1.0 / gw
    q = fma(v, inv, 2.0 ** -20)
    near = not ((q - math.floor(q)) >= 2.0 ** -19)
    return int(q), near


def check(v, gw):
    cell, near = kernel(v, gw)
    want = ref_cell(v, gw)
    if not near:
        assert cell == want, (v, gw, cell, want)
    else:
        assert want in (cell, cell - 1), (v, gw, cell, want)
    return near


def test_random_positions():
    rng = random.Random(7)
    nears = 0
    for gw in GWS:
        for _ in range(4000):
            cells = rng.uniform(-3.0, 5000.0)
            v = cells * gw * (1.0 + rng.uniform(-1e-9, 1e-9))
            nears += check(v, gw)
    assert nears < 40  # (2^-19 of the positions)


def test_positions_planted_next_to_cell_boundaries():
    rng = random.Random(8)
    nears = 0
    for gw in GWS:
        for _ in range(600):
            k = rng.randint(0, 1 << rng.randint(1, 24))
            for e in range(-52, -17):
                for sign in (-1.0, 1.0):
                    # a position e binary orders of a cell away from the boundary k * gw (as well as doubles allow)
                    v = k * gw + sign * math.ldexp(rng.uniform(1.0, 2.0), e) * gw
                    nears += check(v, gw)
                    nears += check(math.nextafter(k * gw, sign * math.inf), gw)
            nears += check(k * gw, gw)
    assert nears > 1000


def test_binade_boundaries_are_cell_boundaries_when_the_reciprocal_is_an_integer():
    """1 / 0.05 rounds to 20.0 and 1 / 0.01 to 100.0: every power of two >= 1 is then a multiple of the cell to the last
    bit, and the first position behind it -- where a binade-limited jump lands -- is `near` whenever the crossing step
    ends within 2^-20 cells of it."""
    for gw, inv in ((0.05, 20.0), (0.01, 100.0), (0.2, 5.0), (0.1, 10.0)):
        assert 1.0 / gw == inv
        for e in range(0, 12):
            b = math.ldexp(1.0, e)
            for d in (0.0, 2.0 ** -40, 2.0 ** -30, 2.0 ** -26):
                v = b + d * gw
                cell, near = kernel(v, gw)
                assert near and ref_cell(v, gw) in (cell, cell - 1)
            cell, near = kernel(b + 1e-3 * gw, gw)
            assert not near and cell == ref_cell(b + 1e-3 * gw, gw) == int(b * inv)


# ---- The exact-residual rule, worked out with rationals -------------------------------------------------
#
# Asked: replace the real division behind the guard by a residual.  With Q = RN(v / gw) (the reference's fdiv), v >= 0,
# gw > 0 and an integer m >= 1 below 2^52 (m's significand is even, its predecessor's odd, so a tie rounds to m):
#
#     trunc(Q) >= m   <=>   v / gw >= m - h(m)   <=>   m * gw - v <= h(m) * gw        (all exact)
#
# where h(m) is half the gap between m and the double below it: 2^(e - 53) for 2^e < m < 2^(e + 1), 2^(e - 54) for m = 2^e.
# In doubles: r = fma(m, gw, -v) is the correctly rounded residual, T = h(m) * gw is exact (a power of two times gw), and
# rounding is monotone, so  r < T  proves  trunc(Q) >= m,  r > T  proves  trunc(Q) < m,  and only  r == T  is undecided.
# The kernel's guard already narrows the reference's cell to {k, k - 1} (k = trunc(q'')); the residual at m = k decides.
#
# What the rule is worth: it replaces the DIVISION, which only a `near` position takes (2^-19 of the
# positions: one wave-group in a thousand) -- it cannot replace the GUARD, because deciding a position with it costs
# int -> double, fma, the exponent of k, a scaled gw, the power-of-two case and a compare against fract + compare of the
# guard.  The multiplication and the guard are the 34 instructions per trip a general width costs; the rule removes none.


def half_gap_below(m):
    """h(m): half the distance from the integer m (as a double) to the double below it."""
    assert 1 <= m < (1 << 52)
    e = m.bit_length() - 1
    return Fraction(2) ** (e - 54) if m & (m - 1) == 0 else Fraction(2) ** (e - 53)


def residual_rule_exact(v, gw, m):
    """trunc(RN(v / gw)) >= m, decided with rationals only."""
    return Fraction(m) * Fraction(gw) - Fraction(v) <= half_gap_below(m) * Fraction(gw)


def residual_rule_doubles(v, gw, m):
    """The same from one fma and one exact scaling; None where the rounded residual equals the threshold."""
    r = fma(float(m), gw, -v)
    t = float(half_gap_below(m) * Fraction(gw))
    assert Fraction(t) == half_gap_below(m) * Fraction(gw)  # (a power of two times gw: exact)
    if r < t:
        return True
    if r > t:
        return False
    return None


def check_residual(v, gw, counts):
    cell, near = kernel(v, gw)
    want = ref_cell(v, gw)
    for m in (cell - 1, cell, cell + 1):
        if m < 1:
            continue
        assert residual_rule_exact(v, gw, m) == (want >= m), (v, gw, m, want)
        d = residual_rule_doubles(v, gw, m)
        if d is None:
            counts["undecided"] += 1
        else:
            assert d == (want >= m), (v, gw, m, want)
    if near and cell >= 1:
        # what the kernel would do with the rule: the guard says {cell, cell - 1}, the residual at m = cell picks one
        d = residual_rule_doubles(v, gw, cell)
        if d is not None:
            assert (cell if d else cell - 1) == want
            counts["near_decided"] += 1
        else:
            counts["near_undecided"] += 1


def test_exact_residual_rule_decides_the_cell_without_a_division():
    rng = random.Random(9)
    counts = {"undecided": 0, "near_decided": 0, "near_undecided": 0}
    for gw in GWS:
        for _ in range(1500):
            cells = rng.uniform(0.0, 5000.0)
            check_residual(cells * gw * (1.0 + rng.uniform(-1e-9, 1e-9)), gw, counts)
        for _ in range(150):
            k = rng.randint(1, 1 << rng.randint(1, 24))
            for e in range(-56, -17, 2):
                for sign in (-1.0, 1.0):
                    v = k * gw + sign * math.ldexp(rng.uniform(1.0, 2.0), e) * gw
                    if v >= 0.0:
                        check_residual(v, gw, counts)
            # the rounding boundary itself and its neighbours: the doubles around (k - h(k)) * gw
            b = float((Fraction(k) - half_gap_below(k)) * Fraction(gw))
            for v in (math.nextafter(b, -math.inf), b, math.nextafter(b, math.inf)):
                check_residual(v, gw, counts)
            # powers of two: the gap below them is half the gap above
            p = 1 << rng.randint(0, 24)
            b = float((Fraction(p) - half_gap_below(p)) * Fraction(gw))
            for v in (math.nextafter(b, -math.inf), b, math.nextafter(b, math.inf), p * gw):
                check_residual(v, gw, counts)
    assert counts["near_decided"] > 2000
    # r == T needs the exact residual within half an ulp of T: it happens (only) on positions planted on the boundary
    assert counts["near_undecided"] <= counts["near_decided"] // 10
