"""CPU property test of the shortcut the production kernel takes through the slab test
(csrc/device_common.hpp slab_classify): from reciprocal-multiply approximations of the six slab parameters it
either proves a miss, or names the ONE quotient that is distance()'s result (AABB.cpp:49-77) and divides once.

Model: numpy float64 (IEEE division = the reference's), the approximate reciprocal perturbed by up to +-8 ulp
(v_rcp_f64 is far better), verdicts exactly as the device code forms them.  Checked against a literal restatement
of distance() on random rays and on the adversarial ones: rays through edges and corners of the box (equal lower
ends), origins on a face, thin boxes, boxes behind the origin, huge and tiny direction components."""
import numpy as np

INF = np.inf


def distance_literal(o, d, c0, c1):
    """AABB.cpp:49-77, axis order x, y, z, same comparisons (NaN => every test false)."""
    lo, hi = -INF, INF
    for i in range(3):
        with np.errstate(all="ignore"):
            a, b = (c0[i] - o[i]) / d[i], (c1[i] - o[i]) / d[i]
        if a > b:
            a, b = b, a
        if b < lo or a > hi:
            return INF
        if a > lo:
            lo = a
        if b < hi:
            hi = b
    return INF if lo > hi else lo


def classify(o, d, c0, c1, may_report_miss, rng):
    """device_common.hpp slab_classify: (verdict, d)."""
    LO, LO2, HI, mag = -INF, -INF, INF, 0.0
    num, den, span = 0.0, 1.0, 0.0
    fine = True
    for i in range(3):
        fine = fine and (abs(d[i]) > 2.0 ** -500) and (abs(d[i]) < 2.0 ** 500)
        with np.errstate(all="ignore"):
            inv = np.float64(1.0) / d[i]
            inv = inv * np.float64(1.0 + rng.randint(-8, 9) * 2.0 ** -52)  # an approximate reciprocal
            n0, n1 = c0[i] - o[i], c1[i] - o[i]
            t0, t1 = n0 * inv, n1 * inv
        fine = fine and (abs(t0) < 2.0 ** 500) and (abs(t1) < 2.0 ** 500)
        first = t0 < t1
        lo_i, hi_i = (t0, t1) if first else (t1, t0)
        bigger = lo_i > LO
        LO2 = LO if bigger else max(LO2, lo_i)
        if bigger:
            num, den, span, LO = (n0 if first else n1), d[i], hi_i - lo_i, lo_i
        HI = min(HI, hi_i)
        mag = max(mag, abs(t0), abs(t1))
    if not fine:
        return 0, None
    if may_report_miss and (HI < 0.0 or (LO - HI) > mag * 2.0 ** -12):
        return 1, None
    margin = mag * 2.0 ** -30
    if (HI - LO) > margin and (LO - LO2) > margin and span > margin:
        return 2, num / den
    return 0, None


def bits(v):
    return np.float64(v).view(np.uint64)


def run(cases, rng):
    with np.errstate(all="ignore"):  # (non-finite intermediates are part of the cases)
        return _run(cases, rng)


def _run(cases, rng):
    decided = misses = 0
    for o, d, c0, c1 in cases:
        o, d, c0, c1 = (np.asarray(v, dtype=np.float64) for v in (o, d, c0, c1))
        want = distance_literal(o, d, c0, c1)
        for may_miss in (False, True):
            verdict, got = classify(o, d, c0, c1, may_miss, rng)
            if verdict == 1:
                assert want == INF or want < 0.0, (o, d, c0, c1, want)
                misses += 1
            elif verdict == 2:
                assert bits(got) == bits(want), (o, d, c0, c1, got, want)
                decided += 1
    return decided, misses


def test_random_rays_and_boxes():
    rng = np.random.RandomState(3)
    cases = []
    for _ in range(40000):
        c0 = np.array([0.0, 0.0, rng.uniform(-5, 5)])
        c1 = np.array([rng.uniform(0.01, 5000), -rng.uniform(0.01, 5000), c0[2] + rng.uniform(1e-6, 300)])
        o = np.array([rng.uniform(-6000, 6000), rng.uniform(-6000, 6000), rng.uniform(-400, 2000)])
        if rng.randint(2):  # aimed at a point inside the box
            v = c0 + rng.uniform(0, 1, size=3) * (c1 - c0) - o
        else:
            v = rng.normal(size=3) * rng.choice([1.0, 1e-3, 1e-9, 1e6], size=3)
        cases.append((o, v / np.sqrt((v * v).sum()), c0, c1))
    decided, misses = run(cases, rng)
    assert decided > 30000 and misses > 5000  # (the shortcut does decide most rays, both ways)


def test_edges_corners_faces_and_degenerate_directions():
    rng = np.random.RandomState(4)
    c0, c1 = np.array([0.0, 0.0, 0.0]), np.array([64.0, -64.0, 8.0])
    cases = []
    corners = [np.array([x, y, z]) for x in (0.0, 64.0) for y in (0.0, -64.0) for z in (0.0, 8.0)]
    for _ in range(4000):
        o = np.array([rng.uniform(-100, 160), rng.uniform(-160, 100), rng.uniform(-20, 60)])
        target = corners[rng.randint(8)].copy()
        kind = rng.randint(4)
        if kind == 1:  # a point on an edge
            target[rng.randint(3)] = rng.uniform(-64, 64)
        elif kind == 2:  # a point on a face
            target[rng.randint(3)] = rng.uniform(-64, 64)
            target[rng.randint(3)] = rng.uniform(-64, 64)
        elif kind == 3:  # origin on a face plane: a zero numerator
            o[rng.randint(3)] = rng.choice([0.0, 64.0, -64.0, 8.0])
        v = target - o
        if rng.randint(5) == 0:
            v[rng.randint(3)] = rng.choice([0.0, -0.0, 1e-320, 1e-200])  # axis-parallel and nearly so
        n = np.sqrt((v * v).sum())
        cases.append((o, v / n if n > 0 else v, c0, c1))
        cases.append((o, v, c0, c1))  # (un-normalised: the orthographic plane's directions are not unit vectors either)
    # thin boxes and a box of zero height
    for _ in range(1000):
        z = rng.uniform(-3, 3)
        t0, t1 = np.array([0.0, 0.0, z]), np.array([32.0, -32.0, z + rng.choice([0.0, 1e-12, 1e-300, 1e-3])])
        o = np.array([rng.uniform(-50, 80), rng.uniform(-80, 50), rng.uniform(-10, 10)])
        v = rng.normal(size=3)
        cases.append((o, v / np.sqrt((v * v).sum()), t0, t1))
    decided, misses = run(cases, rng)
    assert decided > 1000 and misses > 1000


def test_huge_and_tiny_magnitudes():
    rng = np.random.RandomState(5)
    cases = []
    for _ in range(5000):
        s = 2.0 ** rng.randint(-300, 300)
        c0, c1 = np.array([0.0, 0.0, 0.0]), np.array([s, -s, s * 0.125])
        o = np.array([rng.uniform(-3, 4), rng.uniform(-4, 3), rng.uniform(-1, 2)]) * s
        v = rng.normal(size=3) * (2.0 ** rng.randint(-600, 600, size=3).astype(np.float64))
        cases.append((o, v, c0, c1))
    run(cases, rng)
