"""Offline estimate: trips per ray under (a) the kernel's level policy (ported from render_fast.hip), (b) the same with the best of
{finer, lev, coarser} looked up in one attempt, (c) the best of all levels (oracle)."""
import importlib, sys, math
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
wlname = sys.argv[1] if len(sys.argv) > 1 else "C3"
wl = hmrm.synth.WORKLOADS[wlname]
S = wl.map_size
rgb, _ = hmrm.synth.synth_maps(S)
par = wl.scene_params()
thr = rgb[:, :, 0].astype(np.float64) / 255.0 * (par.max_height - par.min_height) + 2 * par.min_height
THRMAX = thr.max()
cam = wl.camera()
rec = hmrm.debug_frame(cam, par, S, S)
K = 7; TOP = 7
cells_per_step = abs(cam.step_dist / par.grid_width)
MINLEV = 2 if cells_per_step > 0.35 else 0
FINEST_PAUSE = 6 if cells_per_step > 0.35 else 0
cache = {}

def wmax(level, wx0, wy0, Sz):
    k = (level, wx0, wy0)
    if k not in cache:
        cache[k] = thr[wy0:wy0 + Sz, wx0:wx0 + Sz].max()
    return cache[k]

def look(level, p, s):
    """-> (n steps possible (estimate, after verification semantics), above, z_bound, crossed-ish)"""
    if level == TOP:
        m = THRMAX; room = 1e30
    else:
        gx, gy = int(p[0]), int(-p[1])
        Sz = 4 << level
        st = 2 if level == 0 else Sz // 4
        back = 1 if level == 0 else 3
        ix = gx // st - (back if s[0] < 0 else 0); iy = gy // st - (back if -s[1] < 0 else 0)
        wx0, wy0 = max(ix, 0) * st, max(iy, 0) * st
        m = wmax(level, wx0, wy0, Sz)
        ex = min(wx0 + Sz, S) if s[0] >= 0 else wx0
        ey = min(wy0 + Sz, S) if -s[1] >= 0 else wy0
        rx = (ex - p[0]) / s[0] if s[0] != 0 else 1e30
        ry = (-ey - p[1]) / s[1] if s[1] != 0 else 1e30
        room = min(abs(rx), abs(ry))
    above = p[2] >= m
    rz = (m - p[2]) / s[2] if s[2] < 0 else 1e30
    z_bound = rz < room
    n = int(min(room, rz) * 0.998) - 1
    return n, above, z_bound, room, rz

def ray(px, py):
    sva = rec["row_sin_va"][py]; cva = rec["row_cos_va"][py]; cha = rec["col_cos_ha"][px]; sha = rec["col_sin_ha"][px]
    d = np.array([sva * cha, sva * sha, cva])
    pos = np.array(rec["cam"]); c0 = rec["c0"]; c1 = rec["c1"]
    lo, hi = -np.inf, np.inf
    for i in range(3):
        if d[i] == 0: continue
        a = (c0[i] - pos[i]) / d[i]; b = (c1[i] - pos[i]) / d[i]
        if a > b: a, b = b, a
        lo = max(lo, a); hi = min(hi, b)
    if not (lo <= hi) or lo < 0: return None
    return pos + lo * d + rec["nudge"] * d, cam.step_dist * d

def inb(p):
    return p[0] >= 0 and -p[1] >= 0 and int(p[0]) < S and int(-p[1]) < S

def simulate(p, s, mode):
    p = p.copy()
    lev = TOP
    if s[2] < 0:
        descent = (p[2] - 0.0) / -s[2]
        lateral = descent * max(abs(s[0]), abs(s[1]))
        for l in range(K - 1, -1, -2):
            st = 2 if l == 0 else (4 << l) // 4
            nstr = 2 if l == 0 else 4
            if l >= MINLEV and lateral <= (nstr - 1) * st: lev = l
    cooldown = fails = jumps = 0
    attempts = groups = trips = 0
    while True:
        trips += 1
        skip = False
        if not inb(p): return attempts, groups, trips - 1, False
        if cooldown == 0:
            attempts += 1
            cands = [lev]
            young = jumps <= 8
            lstep = 2 if young else 1
            coarser = min(lev + lstep, K - 1) if lev != TOP else TOP
            if mode == 1:
                finer_c = (K - 1) if lev == TOP else max(lev - lstep, MINLEV)
                cands = sorted({finer_c, lev, coarser if lev != TOP else TOP, TOP if lev >= K - 2 else lev})
            elif mode == 2:
                cands = list(range(MINLEV, K)) + [TOP]
            best = None
            for L in cands:
                r = look(L, p, s)
                if best is None or (r[0] if r[1] else -1) > (best[1][0] if best[1][1] else -1): best = (L, r)
            L, (n, above, z_bound, room, rz) = best
            lev_used = L
            ok = above and n >= 2
            if ok:
                p = p + n * s
            top = lev_used == TOP
            height_limited = (not above) or z_bound
            if ok: jumps += 1
            drop = lstep
            finer = (K - 1) if top else max(lev_used - drop, MINLEV)
            at_finest = lev_used == MINLEV
            crossed = ok and not z_bound
            hl = (not crossed) and height_limited
            other = (not crossed) and (not hl)
            room_up = room * (2 ** lstep)
            coars = min(lev_used + lstep, K - 1) if not top else TOP
            go_up = (crossed and rz >= room_up) or other
            fails_before = fails
            lev = finer if hl else (coars if go_up else lev_used)
            if lev_used == K - 1 and go_up and crossed: lev = TOP if rz >= 4 * room else lev
            fails = 0 if (crossed or (hl and ok)) else fails + (1 if other else 0)
            cooldown = FINEST_PAUSE if (hl and not ok and at_finest) else ((min(fails_before, 3)) if other else 0)
            skip = (hl and not ok and not at_finest) or ok
        else:
            cooldown -= 1
        if skip: continue
        groups += 1
        for _ in range(4):
            if not inb(p): return attempts, groups, trips, False
            if p[2] < thr[int(-p[1]), int(p[0])]: return attempts, groups, trips, True
            p = p + s

tot = {m: [0, 0, 0] for m in (0, 1, 2)}
n = 0
rows = range(700, 2160, 17) if wlname in ("C3", "C3h") else range(0, 2160, 23)
for py in rows:
    for px in range(7, 3840, 61):
        r = ray(px, py)
        if r is None: continue
        res = [simulate(r[0], r[1], m) for m in (0, 1, 2)]
        assert len({x[3] for x in res}) == 1
        for m in (0, 1, 2):
            for k in range(3): tot[m][k] += res[m][k]
        n += 1
print(wlname, "rays", n, "minlev", MINLEV)
for m, name in ((0, "kernel policy (ported)"), (1, "best of finer / same / coarser"), (2, "best of all levels")):
    print(f"{name:32s} attempts {tot[m][0]:7d} groups {tot[m][1]:6d} trips {tot[m][2]:7d} per ray {tot[m][2] / n:.2f}")
