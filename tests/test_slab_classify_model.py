"""CPU property test of the shortcut the production kernel takes through the slab test
(csrc/device_common.hpp slab_classify): from reciprocal-multiply approximations of the six slab parameters it
either proves a miss, or names the ONE quotient that is distance()'s result (AABB.cpp:49-77) and divides once.

Model: numpy float64 (IEEE division = the reference's), the approximate reciprocal perturbed by the error MEASURED
for v_rcp_f64 on gfx950 -- up to 2^-24 relative (profiles/r03_rcp_accuracy.txt: 2^-24.36 over 1.6e11 inputs; the GPU
suite re-measures it, test_rcp_f64_accuracy_bound) -- drawn uniformly or pinned to either extreme per axis, verdicts
exactly as the device code forms them.  Checked against a literal restatement of distance() on random rays and on
the adversarial ones: rays through edges and corners of the box (equal lower ends) and past them at relative
offsets 2^-14 .. 2^-50, origins on a face, thin boxes, boxes behind the origin, huge and tiny direction components.
A control shows the test has teeth: round 2's margin (2^-30) fails it under the measured error."""
import numpy as np

INF = np.inf
RCP_REL_ERR = 2.0 ** -24        # device_common.hpp kRcpRelErr
SLAB_MARGIN = 16 * RCP_REL_ERR  # device_common.hpp kSlabMargin


def approx_rcp(x, rng):
    """1/x with a relative error of up to RCP_REL_ERR: uniform, or -- half of the time -- exactly at a bound."""
    with np.errstate(all="ignore"):
        inv = np.float64(1.0) / x
    k = rng.randint(4)
    e = RCP_REL_ERR * (rng.uniform(-1.0, 1.0) if k < 2 else (1.0 if k == 2 else -1.0))
    return inv * np.float64(1.0 + e)


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


def classify(o, d, c0, c1, may_report_miss, rng, margin_frac=SLAB_MARGIN):
    """device_common.hpp slab_classify: (verdict, d)."""
    LO, LO2, HI, mag, least = -INF, -INF, INF, 0.0, INF
    num, den, span = 0.0, 1.0, 0.0
    fine = True
    for i in range(3):
        fine = fine and (abs(d[i]) > 2.0 ** -500) and (abs(d[i]) < 2.0 ** 500)
        with np.errstate(all="ignore"):
            inv = approx_rcp(d[i], rng)
            n0, n1 = c0[i] - o[i], c1[i] - o[i]
            t0, t1 = n0 * inv, n1 * inv
        fine = fine and not (np.isnan(t0) or np.isnan(t1))
        first = t0 < t1
        lo_i, hi_i = (t0, t1) if first else (t1, t0)
        bigger = lo_i > LO
        LO2 = LO if bigger else max(LO2, lo_i)
        if bigger:
            num, den, span, LO = (n0 if first else n1), d[i], hi_i - lo_i, lo_i
        HI = min(HI, hi_i)
        mag = max(mag, abs(t0), abs(t1))
        least = min(least, abs(t0), abs(t1))
    if not (fine and mag < 2.0 ** 500 and least > 2.0 ** -500):
        return 0, None
    if may_report_miss and (HI < 0.0 or (LO - HI) > mag * 2.0 ** -12):
        return 1, None
    margin = mag * margin_frac
    if (HI - LO) > margin and (LO - LO2) > margin and span > margin:
        return 2, num / den
    return 0, None


def bits(v):
    return np.float64(v).view(np.uint64)


def run(cases, rng, margin_frac=SLAB_MARGIN, count_wrong=False):
    with np.errstate(all="ignore"):  # (non-finite intermediates are part of the cases)
        return _run(cases, rng, margin_frac, count_wrong)


def _run(cases, rng, margin_frac, count_wrong):
    decided = misses = wrong = 0
    for o, d, c0, c1 in cases:
        o, d, c0, c1 = (np.asarray(v, dtype=np.float64) for v in (o, d, c0, c1))
        want = distance_literal(o, d, c0, c1)
        for may_miss in (False, True):
            for _ in range(3):  # (several draws of the reciprocal errors per ray)
                verdict, got = classify(o, d, c0, c1, may_miss, rng, margin_frac)
                if verdict == 1:
                    ok = want == INF or want < 0.0
                    misses += 1
                elif verdict == 2:
                    ok = bits(got) == bits(want)
                    decided += 1
                else:
                    ok = True
                if count_wrong:
                    wrong += 0 if ok else 1
                else:
                    assert ok, (o, d, c0, c1, verdict, got, want)
    return (decided, misses, wrong) if count_wrong else (decided, misses)


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


def grazing_cases(rng, n):
    """Rays aimed past an edge or a corner of the box at a relative offset 2^-14 .. 2^-50 (either side)."""
    c0, c1 = np.array([0.0, 0.0, 0.0]), np.array([4096.0, -4096.0, 256.0])
    corners = [np.array([x, y, z]) for x in (0.0, 4096.0) for y in (0.0, -4096.0) for z in (0.0, 256.0)]
    cases = []
    for _ in range(n):
        o = np.array([rng.uniform(-6000, 10000), rng.uniform(-10000, 6000), rng.uniform(-500, 3000)])
        target = corners[rng.randint(8)].copy()
        if rng.randint(2):  # a point on an edge instead of the corner itself
            ax = rng.randint(3)
            target[ax] = c0[ax] + rng.uniform(0, 1) * (c1[ax] - c0[ax])
        v = target - o
        # tilt the direction by a tiny relative amount: the entry point moves across the edge by about that much
        eps = 2.0 ** -rng.uniform(14, 50) * rng.choice([-1.0, 1.0], size=3) * (rng.randint(2, size=3))
        v = v * (1.0 + eps)
        if rng.randint(2):
            v = v / np.sqrt((v * v).sum())
        cases.append((o, v, c0, c1))
    return cases


def test_rays_grazing_edges_and_corners_at_tiny_offsets():
    rng = np.random.RandomState(6)
    decided, misses = run(grazing_cases(rng, 12000), rng)
    assert decided > 2000  # (most of these are undecided by design; the decided ones must be right)


def test_control_round2_margin_fails_under_the_measured_error():
    """The same grazing rays with round 2's margin (2^-30 of the largest parameter, justified by an asserted
    2^-48 reciprocal error) DO produce wrong entry distances once the reciprocal has its measured 2^-24 error."""
    rng = np.random.RandomState(7)
    _, _, wrong = run(grazing_cases(rng, 12000), rng, margin_frac=2.0 ** -30, count_wrong=True)
    assert wrong > 0


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


def test_edge_fuzzer_cameras_do_graze(oracle):
    """tests/deep_fuzz_edges.py (a slice of it runs in the GPU suite): well over 1 % of its rays pass an edge or a
    corner of the box within the shortcut's margin -- lower ends, or entry and exit, within 2^-20 of the largest
    slab parameter -- i.e. they land on the hand-over between the one-division and the six-division path."""
    import deep_fuzz_edges as e
    rays = grazing = frames_with = 0
    for seed in range(90):
        mw, mh, _, params, cam, _ = e.grazing_case(seed)
        cfg = oracle.make_cfg(cam, params, mw, mh)
        c0 = np.array([0.0, 0.0, params.min_height])
        c1 = np.array([mw * params.grid_width, -(mh * params.grid_width), params.max_height])
        here = 0
        for py in range(0, cam.height, 3):
            for px in range(0, cam.width, 3):
                o, d, _ = oracle.probe_ray(cfg, px, py)
                with np.errstate(all="ignore"):
                    t0, t1 = (c0 - o) / d, (c1 - o) / d
                if not (np.isfinite(t0).all() and np.isfinite(t1).all()):
                    continue
                lo, hi = np.sort(np.minimum(t0, t1)), np.maximum(t0, t1).min()
                mag = max(np.abs(t0).max(), np.abs(t1).max())
                rays += 1
                if (lo[2] - lo[1]) <= mag * SLAB_MARGIN or abs(hi - lo[2]) <= mag * SLAB_MARGIN:
                    here += 1
        grazing += here
        frames_with += 1 if here else 0
    assert grazing > 0.05 * rays and frames_with > 45, (grazing, rays, frames_with)
