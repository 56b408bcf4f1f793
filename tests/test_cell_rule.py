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
