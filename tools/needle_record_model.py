"""Offline model (no GPU) of TWO-VALUE WINDOW RECORDS for content that defeats a flat window maximum.

A record per window holds the window's maximum with its K highest cells removed (`max2`) and those cells' places.  A ray
whose heights over a leap stay at or above `max2` may cross the window if its path misses the K cells -- proved by a slab
test of the leap's segment against each cell's box, blown up by a margin.  Anything else is marched by the plain groups
(6 positions a trip, every height loaded: the kernel that ships for such maps today).

Waves are the kernel's 8 x 8 pixel tiles.  A wave's iteration issues the record block once if ANY live lane attempts a
leap, and the group block once if ANY live lane marches: that is what the kernel's time follows.  Printed
per variant: the issue cost of the sampled frame relative to the plain groups alone, for a few prices of the record block.
Every variant must end each ray at the same position index as the plain march (asserted).

usage: python tools/needle_record_model.py [needles|white|spikes|smooth] [C3|C5] [tile-row stride] [tile-col stride]
"""
import importlib
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")

kind = sys.argv[1] if len(sys.argv) > 1 else "needles"
wlname = sys.argv[2] if len(sys.argv) > 2 else "C3"
ROW_STRIDE = int(sys.argv[3]) if len(sys.argv) > 3 else 6
COL_STRIDE = int(sys.argv[4]) if len(sys.argv) > 4 else 16
wl = hmrm.synth.WORKLOADS[wlname]
S = wl.map_size
v8 = hmrm.synth.content_heights_u8(S, kind)
par = wl.scene_params()
assert par.grid_width == 1.0
thr = v8.astype(np.float64) / 255.0 * (par.max_height - par.min_height) + 2 * par.min_height
cam = wl.camera()
rec = hmrm.debug_frame(cam, par, S, S)
GROUP = 6          # positions of one plain group (kGroupPlain)
SETUP = 2.0        # ray set-up + shading of a wave, in group-block units
MARGIN = 1e-3      # cells: what the slab test blows a recorded cell up by


def ray(px, py):
    """(p0, step) of pixel (px, py) or None if it misses the grid's box -- as tools/coop_model.py make_lane"""
    if cam.projection == 2:
        sva = rec["row_sin_va"][py]; cva = rec["row_cos_va"][py]; cha = rec["col_cos_ha"][px]; sha = rec["col_sin_ha"][px]
        d = np.array([sva * cha, sva * sha, cva])
    else:
        w = px / (cam.width - 1); h = py / (cam.height - 1)
        vv = np.array(rec["upper_left"]) + w * np.array(rec["plane_right"]) + h * np.array(rec["plane_down"]) - np.array(rec["cam"])
        d = vv / math.sqrt(float(vv @ vv))
    pos = np.array(rec["cam"]); c0 = rec["c0"]; c1 = rec["c1"]
    lo, hi = -np.inf, np.inf
    for i in range(3):
        if d[i] == 0:
            continue
        a = (c0[i] - pos[i]) / d[i]; b = (c1[i] - pos[i]) / d[i]
        if a > b:
            a, b = b, a
        lo = max(lo, a); hi = min(hi, b)
    if not (lo <= hi) or lo < 0:
        return None
    return pos + (lo + rec["nudge"]) * d, cam.step_dist * d


def march(p, s):
    """the literal loop: positions until the hit / the grid's edge.  -> (xs, ys, zs, n) with n positions visited (the last one decides)"""
    n_cap = int(2.0 * S / max(abs(s[0]), abs(s[1]), 1e-9)) + 8
    n_cap = min(n_cap, 200000)
    k = np.arange(n_cap, dtype=np.float64)
    xs = p[0] + k * s[0]; ys = p[1] + k * s[1]; zs = p[2] + k * s[2]
    gx = np.floor(xs).astype(np.int64); gy = np.floor(-ys).astype(np.int64)
    inside = (xs >= 0) & (-ys >= 0) & (gx < S) & (gy < S)
    out = np.nonzero(~inside)[0]
    n_in = int(out[0]) if len(out) else n_cap
    t = thr[gy[:n_in], gx[:n_in]]
    hit = np.nonzero(zs[:n_in] < t)[0]
    n = int(hit[0]) + 1 if len(hit) else n_in + 1          # the position that hits, or the first one outside
    n = min(n, n_cap)
    return xs[:n], ys[:n], zs[:n], gx[:n], gy[:n], n


_records = {}


def record(level, wx0, wy0, K):
    """(max2, [(cx, cy)...]) of the window at (wx0, wy0): max with the K highest cells removed, and their cells"""
    key = (level, wx0, wy0, K)
    r = _records.get(key)
    if r is None:
        Sz = 4 << level
        t = thr[wy0:wy0 + Sz, wx0:wx0 + Sz]
        flat = t.ravel()
        if K >= flat.size:
            r = (-np.inf, [(wx0 + i % t.shape[1], wy0 + i // t.shape[1]) for i in range(flat.size)])
        else:
            idx = np.argpartition(flat, flat.size - K)[flat.size - K:] if K else np.array([], dtype=np.int64)
            rest = np.delete(flat, idx)
            m2 = float(rest.max())
            cells = [(wx0 + int(i) % t.shape[1], wy0 + int(i) // t.shape[1]) for i in idx if flat[i] > m2]
            r = (m2, cells)
        _records[key] = r
    return r


def window(level, gx, gy, sx, sy):
    """the kernel's window for a ray at cell (gx, gy): quarter-stride placement, >= 3/4 of the window ahead (render_fast.hip)"""
    Sz = 4 << level
    st = 2 if level == 0 else Sz // 4
    back = 1 if level == 0 else 3
    ix = gx // st - (back if sx < 0 else 0)
    iy = gy // st - (back if sy > 0 else 0)
    return max(ix, 0) * st, max(iy, 0) * st, Sz


def misses(x0, y0, x1, y1, cells):
    """segment (x0, -y0) -> (x1, -y1) in cell units against each cell's box blown up by MARGIN: True if it misses all"""
    ax, ay, bx, by = x0, -y0, x1, -y1
    dx, dy = bx - ax, by - ay
    for cx, cy in cells:
        t0, t1 = 0.0, 1.0
        for a, d, lo, hi in ((ax, dx, cx - MARGIN, cx + 1 + MARGIN), (ay, dy, cy - MARGIN, cy + 1 + MARGIN)):
            if d == 0.0:
                if a < lo or a > hi:
                    t0, t1 = 1.0, 0.0
                continue
            u, w = (lo - a) / d, (hi - a) / d
            if u > w:
                u, w = w, u
            t0 = max(t0, u); t1 = min(t1, w)
        if t0 <= t1:
            return False
    return True


def schedule(R, level, K, cooldown):
    """per-trip record of one lane under the record scheme: list of (attempted, marched) per trip"""
    xs, ys, zs, gxs, gys, n, s = R
    i = 0
    trips = []
    cool = 0
    while i < n:
        attempted = marched = False
        leaped = 0
        if level >= 0 and cool == 0 and i < n - 1:
            attempted = True
            wx0, wy0, Sz = window(level, int(gxs[i]), int(gys[i]), s[0], s[1])
            m2, cells = record(level, wx0, wy0, K)
            # positions that stay inside the window (counted on the sequence; the kernel's estimate is a step or two short of it)
            j = i
            while j < n - 1 and wx0 <= gxs[j] < wx0 + Sz and wy0 <= gys[j] < wy0 + Sz:
                j += 1
            m = j - i - 1                       # leap to position i + m, itself inside the window and still to be tested
            # heights: z is monotone along the ray, so the ends decide
            if zs[i] < m2:
                m = 0
            while m >= 2 and min(zs[i], zs[i + m - 1]) < m2:
                m -= 1 if m < 8 else m // 4    # the kernel would take the z bound's estimate; a model may search
            if m >= 2 and misses(xs[i], ys[i], xs[i + m - 1], ys[i + m - 1], cells):
                leaped = m
        if leaped:
            assert i + leaped <= n - 1, "a leap passed the deciding position"
            i += leaped
        else:
            marched = True
            i += GROUP
            if attempted:
                cool = cooldown
            elif cool:
                cool -= 1
        trips.append((attempted, marched))
    return trips


def frame(level, K, cooldown, rays):
    """issue cost of the sampled frame: sum over waves and iterations of (record block if any lane attempts) + (group block if any marches)"""
    att_blocks = grp_blocks = 0
    lane_trips = 0
    for tile in rays:
        scheds = [schedule(R, level, K, cooldown) for R in tile]
        lane_trips += sum(len(x) for x in scheds)
        for t in range(max((len(x) for x in scheds), default=0)):
            live = [x[t] for x in scheds if len(x) > t]
            att_blocks += any(a for a, _ in live)
            grp_blocks += any(g for _, g in live)
    return att_blocks, grp_blocks, lane_trips


def main():
    rays = []
    n_rays = n_steps = 0
    for ty in range(0, cam.height // 8, ROW_STRIDE):
        for tx in range(0, cam.width // 8, COL_STRIDE):
            tile = []
            for ly in range(8):
                for lx in range(8):
                    r = ray(tx * 8 + lx, ty * 8 + ly)
                    if r is None:
                        continue
                    p, s = r
                    xs, ys, zs, gx, gy, n = march(p, s)
                    tile.append((xs, ys, zs, gx, gy, n, s))
                    n_rays += 1; n_steps += n
            rays.append(tile)
    waves = len(rays)
    print(f"{wlname}/{kind}: {waves} waves sampled (every {ROW_STRIDE}th tile row, {COL_STRIDE}th tile column), {n_rays} rays in the box, "
          f"{n_steps / max(n_rays, 1):.1f} positions per ray")
    a0, g0, lt0 = frame(-1, 0, 0, rays)
    base = g0 + SETUP * waves
    print(f"plain groups: {g0} group blocks ({g0 / waves:.1f} per wave), {lt0 / max(n_rays, 1):.2f} trips per ray")
    print("window  K  cooldown | record blocks  group blocks  trips/ray | cost / plain groups at a record block of 1.0  1.5  2.0 group blocks")
    for level in (1, 2, 3):
        for K in (0, 2, 4, 8):
            for cooldown in (0, 2):
                a, g, lt = frame(level, K, cooldown, rays)
                costs = [(a * c + g + SETUP * waves) / base for c in (1.0, 1.5, 2.0)]
                print(f"{4 << level:4d}   {K:2d}  {cooldown:4d}     | {a:10d}   {g:10d}   {lt / max(n_rays, 1):8.2f}   | "
                      + "   ".join(f"{c:5.2f}" for c in costs), flush=True)


if __name__ == "__main__":
    main()
