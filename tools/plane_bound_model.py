"""Offline estimate: attempts per ray with flat window maxima vs plane upper bounds (oracle level policy: best level each attempt)."""
import importlib, sys, math
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
wlname = sys.argv[1] if len(sys.argv) > 1 else "C3"
wl = hmrm.synth.WORKLOADS[wlname]
S = wl.map_size
rgb, _ = hmrm.synth.synth_maps(S)
par = wl.scene_params()
thr = rgb[:, :, 0].astype(np.float64) / 255.0 * (par.max_height - par.min_height) + 2 * par.min_height  # (grey map)
cam = wl.camera()
rec = hmrm.debug_frame(cam, par, S, S)
LEVELS = [(4 << l) for l in range(7)]  # window sizes
cache_flat, cache_plane = {}, {}

def window(level, gx, gy, dirx, diry):
    """window origin (cells) for a ray at cell (gx, gy) moving with sign dirx, diry: quarter-stride placement, >= 3S/4 room ahead"""
    Sz = LEVELS[level]
    st = max(Sz // 4, 1) if level > 0 else 2
    back = 3 if level > 0 else 1
    ix = gx // st - (back if dirx < 0 else 0)
    iy = gy // st - (back if diry < 0 else 0)
    return max(ix, 0) * st, max(iy, 0) * st, Sz

def flat_bound(level, wx0, wy0, Sz):
    k = (level, wx0, wy0)
    if k not in cache_flat:
        cache_flat[k] = thr[wy0:wy0 + Sz, wx0:wx0 + Sz].max()
    return cache_flat[k]

def plane_bound(level, wx0, wy0, Sz):
    k = (level, wx0, wy0)
    if k not in cache_plane:
        t = thr[wy0:wy0 + Sz, wx0:wx0 + Sz]
        h, w = t.shape
        # gradient from half-window means
        bx = (t[:, w // 2:].mean() - t[:, :w // 2].mean()) / max(w / 2.0, 1) if w > 1 else 0.0
        by = (t[h // 2:, :].mean() - t[:h // 2, :].mean()) / max(h / 2.0, 1) if h > 1 else 0.0
        u = np.arange(w)[None, :]; v = np.arange(h)[:, None]
        # the plane must bound thr over the whole cell: take the cell corner where the linear part is smallest
        lin_min = np.minimum(bx * u, bx * (u + 1)) + np.minimum(by * v, by * (v + 1))
        a = (t - lin_min).max()
        cache_plane[k] = (a, bx, by)
    return cache_plane[k]

def simulate(px, py, mode):
    sva = rec["row_sin_va"][py]; cva = rec["row_cos_va"][py]; cha = rec["col_cos_ha"][px]; sha = rec["col_sin_ha"][px]
    d = np.array([sva * cha, sva * sha, cva]) if cam.projection == 2 else None
    pos = np.array(rec["cam"]); c0 = rec["c0"]; c1 = rec["c1"]
    lo, hi = -np.inf, np.inf
    for i in range(3):
        if d[i] == 0: continue
        a = (c0[i] - pos[i]) / d[i]; b = (c1[i] - pos[i]) / d[i]
        if a > b: a, b = b, a
        lo = max(lo, a); hi = min(hi, b)
    if not (lo <= hi) or lo < 0: return None
    p = pos + lo * d + rec["nudge"] * d
    s = cam.step_dist * d
    attempts = groups = steps = 0
    while True:
        gx, gy = int(p[0]), int(-p[1])
        if not (0 <= gx < S and 0 <= gy < S and p[0] >= 0 and -p[1] >= 0): return attempts, groups, steps, False
        best = 0
        if s[2] >= 0 and p[2] >= thr.max(): return attempts + 1, groups, steps, False
        for level in range(7):
            wx0, wy0, Sz = window(level, gx, gy, s[0], -s[1])
            # lateral room in steps
            rx = ((wx0 + Sz if s[0] > 0 else wx0) - p[0]) / s[0] if s[0] != 0 else 1e30
            ry = ((-(wy0 + Sz) if s[1] < 0 else -wy0) - p[1]) / s[1] if s[1] != 0 else 1e30
            room = min(rx, ry)
            if mode == 0:
                m = flat_bound(level, wx0, wy0, Sz)
                if p[2] < m: continue
                rz = (m - p[2]) / s[2] if s[2] < 0 else 1e30
            else:
                a, bx, by = plane_bound(level, wx0, wy0, Sz)
                g0 = p[2] - (a + bx * (p[0] - wx0) + by * (-p[1] - wy0))
                if g0 < 0: continue
                rate = s[2] - bx * s[0] - by * (-s[1])
                rz = g0 / -rate if rate < 0 else 1e30
            n = int(min(room, rz) * 0.998) - 1
            best = max(best, n)
        # whole map level
        m = thr.max()
        if p[2] >= m and s[2] < 0:
            n = int((m - p[2]) / s[2] * 0.998) - 1
            best = max(best, n)
        attempts += 1
        if best >= 2:
            p = p + best * s; steps += best
            continue
        # group of 4 real steps
        groups += 1
        for _ in range(4):
            gx, gy = int(p[0]), int(-p[1])
            if not (0 <= gx < S and 0 <= gy < S and p[0] >= 0 and -p[1] >= 0): return attempts, groups, steps, False
            steps += 1
            if p[2] < thr[gy, gx]: return attempts, groups, steps, True
            p = p + s

rng = np.random.RandomState(1)
tot = {0: [0, 0, 0], 1: [0, 0, 0]}
n = 0
rows = range(700, 2160, 29) if wlname == "C3" else range(0, 2160, 43)
for py in rows:
    for px in range(7, 3840, 97):
        r0 = simulate(px, py, 0)
        if r0 is None: continue
        r1 = simulate(px, py, 1)
        assert r0[3] == r1[3] and (not r0[3] or r0[2] == r1[2]), (px, py, r0, r1)
        for m, r in ((0, r0), (1, r1)):
            tot[m][0] += r[0]; tot[m][1] += r[1]; tot[m][2] += r[0] + r[1]
        n += 1
print(wlname, "rays", n)
for m in (0, 1):
    print(("flat max " if m == 0 else "plane bound"), "attempts", tot[m][0], "groups", tot[m][1], "trips", tot[m][2], "per ray %.2f" % (tot[m][2] / n))
