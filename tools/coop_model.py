"""Offline model (no GPU) of COOPERATIVE TRAVERSAL for the production kernel's tail: once a wave has
few live rays, the idle lanes take over disjoint step ranges [k_j, k_j+1) of the live rays -- inside a binade
p_k = p_0 + k * delta exactly (render_fast.hip header), so a helper lane can start k_j steps down the ray without having
walked there -- and the lowest range that finds a hit / the grid's edge decides.

The kernel's traversal (level policy, windows, 4-step groups; binade stays as exact step counts) is ported to Python;
waves are the kernel's 8 x 8 pixel tiles on the workload's scene.  Printed: wave-trips (what a launch's VALU time follows)
and the trips of the longest waves (what its tail follows) today and under the cooperative schedule, for a few
thresholds / segment-length rules.  Every variant must find the same hit step for every ray (asserted).

usage: python tools/coop_model.py [C3|C5|C2] [tile-row stride] [tile-col stride]
"""
import importlib
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")

wlname = sys.argv[1] if len(sys.argv) > 1 else "C3"
ROW_STRIDE = int(sys.argv[2]) if len(sys.argv) > 2 else 6
COL_STRIDE = int(sys.argv[3]) if len(sys.argv) > 3 else 16
wl = hmrm.synth.WORKLOADS[wlname]
S = wl.map_size
rgb, _ = hmrm.synth.synth_maps(S)
par = wl.scene_params()
assert par.grid_width == 1.0
thr = rgb[:, :, 0].astype(np.float64) / 255.0 * (par.max_height - par.min_height) + 2 * par.min_height
THRMAX = float(thr.max())
cam = wl.camera()
rec = hmrm.debug_frame(cam, par, S, S)
K = 7
TOP = 7
cells_per_step = abs(cam.step_dist / par.grid_width)
MINLEV = 2 if cells_per_step > 0.35 else 0
FINEST_PAUSE = 6 if cells_per_step > 0.35 else 0
ADAPT_AFTER = 8


def build_pyramid():
    """levels[l][iy, ix] = max of thr over the S x S window at (ix * st, iy * st), S = 4 << l, st = 2 (l = 0) or S / 4."""
    levels = []
    for l in range(K):
        Sz = 4 << l
        st = 2 if l == 0 else Sz // 4
        nb = (S + st - 1) // st
        pad = nb * st - S
        t = np.pad(thr, ((0, pad), (0, pad)), constant_values=-np.inf)
        bm = t.reshape(nb, st, nb, st).max(axis=(1, 3))
        k = Sz // st
        bp = np.pad(bm, ((0, k - 1), (0, k - 1)), constant_values=-np.inf)
        w = bm.copy()
        for dy in range(k):
            for dx in range(k):
                if dx or dy:
                    w = np.maximum(w, bp[dy:dy + nb, dx:dx + nb])
        levels.append(w)
    return levels


PYR = build_pyramid()
_tp = np.pad(thr, ((0, 1), (0, 1)), constant_values=-np.inf)
M2 = np.maximum(np.maximum(_tp[:-1, :-1], _tp[:-1, 1:]), np.maximum(_tp[1:, :-1], _tp[1:, 1:]))   # max over cells (x..x+1, y..y+1), clipped at the edge


def binade_left(p, s):
    """further steps that certainly stay strictly inside p's binade (approximate arithmetic is fine for a model)"""
    if s == 0.0:
        return 1 << 30
    a = abs(p)
    if a < 2.0 ** -900 or not math.isfinite(a):
        return 0
    e = math.floor(math.log2(a))
    lo, hi = 2.0 ** e, 2.0 ** (e + 1)
    away = (s > 0) == (p > 0)
    room = (hi - a) if away else (a - lo)
    k = int(room / abs(s) * (1 - 2.0 ** -22))
    return max(0, min(k, 1 << 30))


CELLS = 0      # > 0: at the finest level a trip walks CELLS cells of the ray's path with their exact thresholds (loads issued together) instead of a 4-cell window look-up
WALK2 = False   # the walk is U fixed-stride segments, each bounded by the 2x2-cell maximum around its two end points (kernel design)
WALK_RESOLVE_TRIP = False
WALK_AFTER_ZJUMP = True
CELL_FLOOR = 0  # the finest window level attempted when cell walks are on (the cell level sits below it)
CELL_UP = 2.0  # ... and goes back to window look-ups (one level up) when the ray cleared all of them by this much
DUAL = 0       # > 0: an attempt whose window refuses the ray (below its maximum) also looks DUAL level moves finer, same trip
MAX_LEGS = 1   # legs of one jump (chained across binade boundaries inside ONE attempt); 1 = today's kernel
KINDS = ("jump to a binade end", "jump across the window", "jump to the window max", "refused: height", "refused: other", "pause")


class Lane:
    __slots__ = ("x", "y", "z", "sx", "sy", "sz", "lev", "cooldown", "fails", "jumps", "lx", "ly", "lz", "steps", "limit",
                 "status", "trips", "attempts", "groups", "legs", "kinds", "log", "duals", "cellwalks")

    def clone(self):
        o = Lane()
        for k in Lane.__slots__:
            setattr(o, k, getattr(self, k))
        o.kinds = list(self.kinds)
        o.log = list(self.log)
        return o


def refresh(L):
    if L.lx < 0:
        L.lx = binade_left(L.x, L.sx)
    if L.ly < 0:
        L.ly = binade_left(L.y, L.sy)
    if L.lz < 0:
        L.lz = binade_left(L.z, L.sz)


def make_lane(px, py):
    if cam.projection == 2:
        sva = rec["row_sin_va"][py]; cva = rec["row_cos_va"][py]; cha = rec["col_cos_ha"][px]; sha = rec["col_sin_ha"][px]
        d = np.array([sva * cha, sva * sha, cva])
    elif cam.projection == 1:
        w = px / (cam.width - 1); h = py / (cam.height - 1)
        v = np.array(rec["upper_left"]) + w * np.array(rec["plane_right"]) + h * np.array(rec["plane_down"]) - np.array(rec["cam"])
        d = v / math.sqrt(float(v @ v))
    else:
        raise SystemExit("orthographic: not modelled")
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
    p = pos + lo * d + rec["nudge"] * d
    s = cam.step_dist * d
    L = Lane()
    L.x, L.y, L.z = float(p[0]), float(p[1]), float(p[2])
    L.sx, L.sy, L.sz = float(s[0]), float(s[1]), float(s[2])
    L.lev = TOP
    if L.sz < 0:
        descent = (L.z - 0.0) / -L.sz
        lateral = descent * max(abs(L.sx), abs(L.sy))
        for l in range(K - 1, -1, -2):
            st = 2 if l == 0 else (4 << l) // 4
            nstr = 2 if l == 0 else 4
            if l >= MINLEV and lateral <= (nstr - 1) * st:
                L.lev = l
    L.cooldown = L.fails = L.jumps = 0
    L.lx = L.ly = L.lz = -1
    L.steps = 0
    L.limit = 1 << 60   # helper lanes: stop once this many steps are covered
    L.status = 0        # 0 running, 1 hit, 2 left the grid, 3 segment covered
    L.trips = L.attempts = L.groups = L.legs = L.duals = L.cellwalks = 0
    L.kinds = [0, 0, 0, 0, 0, 0]
    L.log = []   # per trip: (attempt made, legs of its jump, group marched)
    return L


def inb(x, y):
    return x >= 0 and -y >= 0 and int(x) < S and int(-y) < S


def window_of(L, lev, in0):
    """the window the kernel looks up at `lev` for this position: (maximum, lateral room in steps)"""
    room = 1e30
    m = THRMAX
    if in0 and lev != TOP:
        gx, gy = int(L.x), int(-L.y)
        Sz = 4 << lev
        st = 2 if lev == 0 else Sz // 4
        back = 1 if lev == 0 else 3
        ix = gx // st - (back if L.sx < 0 else 0)
        iy = gy // st - (back if L.sy > 0 else 0)
        ix = max(ix, 0); iy = max(iy, 0)
        m = float(PYR[lev][iy, ix])
        wx0, wy0 = ix * st, iy * st
        ex = min(wx0 + Sz, S) if L.sx >= 0 else wx0
        ey = min(wy0 + Sz, S) if L.sy <= 0 else wy0
        rx = abs((ex - L.x) / L.sx) if L.sx != 0 else 1e30
        ry = abs((-ey - L.y) / L.sy) if L.sy != 0 else 1e30
        room = min(rx, ry)
    elif in0:
        ex = S if L.sx >= 0 else 0
        ey = S if L.sy <= 0 else 0
        rx = abs((ex - L.x) / L.sx) if L.sx != 0 else 1e30
        ry = abs((-ey - L.y) / L.sy) if L.sy != 0 else 1e30
        room = min(rx, ry)
    return m, room


def trip(L):
    """one trip of the kernel's loop for one lane (attempt and / or group); returns nothing, updates L"""
    L.trips += 1
    skip = False
    trip_legs = -1
    if CELLS and L.lev == -1 and inb(L.x, L.y) and WALK2:
        L.cellwalks += 1
        refresh(L)
        a = int(0.999 / max(abs(L.sx), abs(L.sy), 1e-30))
        a = max(1, min(a, 1 << 20))
        kmax = min(L.lx, L.ly, L.lz)
        done_segs = 0
        margin = 1e30
        for c in range(CELLS):
            k0, k1 = c * a, (c + 1) * a
            if k1 > kmax:
                break
            x0, y0, z0 = L.x + k0 * L.sx, L.y + k0 * L.sy, L.z + k0 * L.sz
            x1, y1, z1 = L.x + k1 * L.sx, L.y + k1 * L.sy, L.z + k1 * L.sz
            if not (inb(x0, y0) and inb(x1, y1)):
                break
            gx, gy = min(int(x0), int(x1)), min(int(-y0), int(-y1))
            mm = M2[gy, gx]
            if min(z0, z1) < mm:
                break
            margin = min(margin, min(z0, z1) - mm)
            done_segs += 1
        adv = done_segs * a
        if adv:
            L.x += adv * L.sx; L.y += adv * L.sy; L.z += adv * L.sz
            L.steps += adv
            L.lx -= adv; L.ly -= adv; L.lz -= adv
        L.log.append((True, -CELLS, done_segs < CELLS))
        if done_segs < CELLS:
            # stopped early: the group of real steps runs in the same trip
            L.groups += 1
            for _ in range(4):
                if not inb(L.x, L.y):
                    L.status = 2
                    return
                L.steps += 1
                if L.z < thr[int(-L.y), int(L.x)]:
                    L.status = 1
                    return
                L.x += L.sx; L.y += L.sy; L.z += L.sz
            L.lx -= 4; L.ly -= 4; L.lz -= 4
        elif margin > CELL_UP:
            L.lev = CELL_FLOOR
        if L.steps >= L.limit:
            L.status = 3
        return
    if CELLS and L.lev == -1 and inb(L.x, L.y):
        # cell walk: per cell the steps the ray stays in it, exact threshold; hit = first step with z < T (hmap.cpp:1016)
        L.cellwalks += 1
        margin = 1e30
        for _ in range(CELLS):
            if not inb(L.x, L.y):
                L.status = 2
                break
            cx, cy = int(L.x), int(-L.y)
            T = thr[cy, cx]
            while inb(L.x, L.y) and int(L.x) == cx and int(-L.y) == cy:
                L.steps += 1
                if L.z < T:
                    L.status = 1
                    break
                margin = min(margin, L.z - T)
                L.x += L.sx; L.y += L.sy; L.z += L.sz
                L.lx -= 1; L.ly -= 1; L.lz -= 1
            if L.status:
                break
        L.log.append((True, -CELLS, False))
        if L.status == 0 and not inb(L.x, L.y):
            L.status = 2
        if L.status and WALK_RESOLVE_TRIP:   # (variant: the walk stops in front of that cell and a group of real steps resolves it: one more trip)
            L.trips += 1
            L.log.append((False, 0, True))
        if L.status == 0:
            if margin > CELL_UP:
                L.lev = CELL_FLOOR
            if L.steps >= L.limit:
                L.status = 3
        return
    if L.cooldown == 0:
        L.attempts += 1
        refresh(L)
        lev = L.lev
        in0 = inb(L.x, L.y)
        m, room = window_of(L, lev, in0)
        if DUAL and in0 and not (L.z >= m):
            # second look-up of the same attempt (loads issued together): DUAL levels finer, used when the first one refuses
            young0 = L.jumps <= ADAPT_AFTER
            lst = 2 if young0 else 1
            alt = (K - 1) if lev == TOP else max(lev - lst * DUAL, MINLEV)
            if alt != lev:
                m2, room2 = window_of(L, alt, in0)
                L.duals += 1
                if L.z >= m2:
                    lev, m, room = alt, m2, room2
        top = lev == TOP
        above = L.z >= m
        rz = (m - L.z) / L.sz if L.sz < 0 else 1e30
        z_bound = rz < room
        n_want = int(min(room, rz, 2.0e9) * 0.998) - 1
        left_lim = min(L.lx, L.ly, L.lz) + 1
        binade_bound = left_lim <= n_want
        z_bound = z_bound and not binade_bound
        n = min(n_want, left_lim)
        ok = in0 and above and n >= 2
        if ok:
            # landing verification: inside the window (by the estimate, conservative enough for a model) and z >= m
            zn = L.z + n * L.sz
            if zn < m:
                ok = False
        if ok:
            L.x += n * L.sx; L.y += n * L.sy; L.z += n * L.sz
            L.steps += n
            L.lx -= n; L.ly -= n; L.lz -= n
            L.legs += 1
            taken = n
            legs = 1
            # chained legs: the jump goes on across the binade boundary inside the same window, no new look-up
            while legs < MAX_LEGS and binade_bound and n_want - taken >= 2:
                refresh(L)
                left_lim = min(L.lx, L.ly, L.lz) + 1
                n2 = min(n_want - taken, left_lim)
                if n2 < 1 or L.z + n2 * L.sz < m:
                    break
                binade_bound = left_lim <= n_want - taken
                L.x += n2 * L.sx; L.y += n2 * L.sy; L.z += n2 * L.sz
                L.steps += n2
                L.lx -= n2; L.ly -= n2; L.lz -= n2
                taken += n2
                legs += 1
                L.legs += 1
            if MAX_LEGS > 1:
                z_bound = (rz < room) and not binade_bound
        L.kinds[(0 if binade_bound else (2 if z_bound else 1)) if ok else (3 if in0 and ((not above) or z_bound) else 4)] += 1
        trip_legs = legs if ok else 0
        height_limited = in0 and ((not above) or z_bound)
        young = L.jumps <= ADAPT_AFTER
        lstep = 2 if young else 1
        if ok:
            L.jumps += 1
        coarser = min(lev + lstep, K - 1)
        drop = lstep
        finer = (K - 1) if top else max(lev - drop, CELL_FLOOR if CELLS else MINLEV)
        minlev = CELL_FLOOR if CELLS else MINLEV
        at_finest = lev == minlev
        crossed = ok and not z_bound
        hl = (not crossed) and height_limited
        other = (not crossed) and (not hl)
        room_up = min(room, rz) * (2 ** lstep)
        go_up = (crossed and rz >= room_up and not binade_bound) or other
        fails_before = L.fails
        L.lev = finer if hl else (coarser if go_up else lev)
        if top and not hl:
            L.lev = TOP
        L.fails = 0 if (crossed or (hl and ok)) else L.fails + (1 if other else 0)
        L.cooldown = FINEST_PAUSE if (hl and not ok and at_finest) else (min(fails_before, 3) if other else 0)
        skip = (hl and not ok and not at_finest) or ok
        if CELLS and hl and at_finest and (not ok or WALK_AFTER_ZJUMP):   # down to the cell level: no pause, no group
            L.lev = -1
            L.cooldown = 0
            skip = True
    else:
        L.cooldown -= 1
        L.kinds[5] += 1
    L.log.append((trip_legs >= 0, max(trip_legs, 0), not skip))
    if not skip:
        L.groups += 1
        for _ in range(4):
            if not inb(L.x, L.y):
                L.status = 2
                return
            L.steps += 1
            if L.z < thr[int(-L.y), int(L.x)]:
                L.status = 1
                return
            L.x += L.sx; L.y += L.sy; L.z += L.sz
        L.lx -= 4; L.ly -= 4; L.lz -= 4
    if L.steps >= L.limit:
        L.status = 3


def run_alone(L):
    while L.status == 0:
        trip(L)
    return L


def exit_estimate(L):
    """steps until the ray certainly is out of the grid or below the floor: the cooperative horizon's outer limit"""
    ex = S if L.sx >= 0 else 0
    ey = S if L.sy <= 0 else 0
    rx = abs((ex - L.x) / L.sx) if L.sx != 0 else 1e30
    ry = abs((-ey - L.y) / L.sy) if L.sy != 0 else 1e30
    rz = (L.z - 0.0) / -L.sz if L.sz < 0 else 1e30
    return min(rx, ry, rz) + 2


def coop_wave(lanes, thresh, seg_rule, overhead):
    """lanes: the wave's rays (None = no march).  Returns (wave trips, rounds, per-ray final (status, steps))."""
    rays = [l.clone() if l is not None else None for l in lanes]
    live = [r for r in rays if r is not None]
    wave_trips = 0
    # phase 1: one ray per lane
    while True:
        running = [r for r in live if r.status == 0]
        if not running or len(running) <= thresh:
            break
        for r in running:
            trip(r)
        wave_trips += 1
    rounds = 0
    running = [r for r in live if r.status == 0]
    while running:
        rounds += 1
        m = 64 // len(running)
        m = 1 << (m.bit_length() - 1)  # power of two lanes per ray
        helpers = []
        for r in running:
            refresh(r)
            horizon = int(min(min(r.lx, r.ly, r.lz), exit_estimate(r)))
            if seg_rule[0] == "exit":
                L = horizon
            else:  # ("recent", c): c x the ray's mean jump length so far per segment
                mean_jump = max(8.0, r.steps / max(1, r.attempts))
                L = int(min(horizon, seg_rule[1] * mean_jump * m))
            seg = max(L // m, 4)
            hs = []
            for j in range(m):
                k = j * seg
                if k > horizon and j > 0:
                    break
                h = r.clone()
                h.x += k * h.sx; h.y += k * h.sy; h.z += k * h.sz
                h.lx -= k; h.ly -= k; h.lz -= k
                h.steps = 0
                h.limit = seg
                h.trips = 0
                hs.append((k, h))
            helpers.append((r, hs))
        # lockstep until every helper is done
        t = 0
        while True:
            any_running = False
            for r, hs in helpers:
                for k, h in hs:
                    if h.status == 0:
                        trip(h)
                        any_running = True
            if not any_running:
                break
            t += 1
        wave_trips += t + overhead
        for r, hs in helpers:
            decided = False
            for k, h in hs:
                if h.status in (1, 2):
                    r.status = h.status
                    r.steps += k + h.steps
                    decided = True
                    break
            if not decided:
                k, h = hs[-1]
                # the ray goes on from the last helper's end (its steps past the segment's end are valid: no hit on the way)
                adv = k + h.steps
                r.x, r.y, r.z = h.x, h.y, h.z
                r.lx, r.ly, r.lz = h.lx, h.ly, h.lz
                r.lev, r.cooldown, r.fails, r.jumps = h.lev, h.cooldown, h.fails, h.jumps
                r.steps += adv
                r.attempts += sum(hh.attempts - r.attempts for _, hh in hs[-1:])
        running = [r for r in live if r.status == 0]
    return wave_trips, rounds, [(r.status, r.steps) if r is not None else None for r in rays]


COST_ATTEMPT, COST_LEG, COST_GROUP, COST_LOOP = 125.0, 60.0, 100.0, 10.0   # VALU + scalar work per block, rough (isa_blocks.py)


def wave_stats(waves, max_legs, dual=0, cells=0, cell_up=2.0, cell_floor=0, walk2=False):
    """per wave: trips of the longest lane, lane trips, results, kinds of the critical lane, modelled issue cost of the wave"""
    global MAX_LEGS, DUAL
    global CELLS, CELL_UP, CELL_FLOOR, WALK2
    WALK2 = walk2
    CELL_FLOOR = cell_floor
    MAX_LEGS = max_legs
    DUAL = dual
    CELLS = cells
    CELL_UP = cell_up
    out = []
    for ty, tx, lanes in waves:
        done = [run_alone(l.clone()) if l is not None else None for l in lanes]
        live = [d for d in done if d is not None]
        crit = max(live, key=lambda d: d.trips)
        cost = 0.0
        extra_legs = 0
        for t in range(crit.trips):
            rows = [d.log[t] for d in live if t < len(d.log)]
            any_attempt = any(r[0] and r[1] >= 0 for r in rows)
            any_walk = any(r[1] < 0 for r in rows)
            xl = max(max(r[1] - 1, 0) for r in rows)
            any_group = any(r[2] for r in rows)
            cost += COST_LOOP + COST_ATTEMPT * any_attempt + COST_LEG * xl + COST_GROUP * any_group + ((30.0 + 18.0 * CELLS) if WALK2 else (40.0 + 44.0 * CELLS)) * any_walk
            extra_legs += xl
        out.append((crit.trips, sum(d.trips for d in live), extra_legs, [(d.status, d.steps) if d is not None else None for d in done],
                    list(crit.kinds), cost))
    MAX_LEGS = 1
    DUAL = 0
    CELLS = 0
    return out


def main():
    tiles_y = (cam.height + 7) // 8
    tiles_x = (cam.width + 7) // 8
    waves = []
    for ty in range(0, tiles_y, ROW_STRIDE):
        for tx in range(3, tiles_x, COL_STRIDE):
            lanes = []
            for l in range(64):
                px, py = tx * 8 + (l & 7), ty * 8 + (l >> 3)
                lanes.append(make_lane(px, py) if px < cam.width and py < cam.height else None)
            if any(l is not None for l in lanes):
                waves.append((ty, tx, lanes))
    print(f"{wlname}: {len(waves)} marching waves sampled (tile rows every {ROW_STRIDE}, tile columns every {COL_STRIDE})")
    base = wave_stats(waves, 1)
    bt = np.array([b[0] for b in base])
    bc = np.array([b[5] for b in base])
    lane_trips = sum(b[1] for b in base)
    kinds = np.array([b[4] for b in base]).sum(axis=0)
    print(f"today: wave-trips {bt.sum()}  lane utilisation {lane_trips / (64.0 * bt.sum()):.3f}  longest waves {sorted(bt)[-5:]}  "
          f"p50 {np.median(bt):.0f} p90 {np.percentile(bt, 90):.0f} p99 {np.percentile(bt, 99):.0f}  modelled issue cost {bc.sum():.0f}")
    print("   trips of each wave's critical lane by kind: " + ", ".join(f"{k} {v / kinds.sum():.2f}" for k, v in zip(KINDS, kinds)))
    long_k = np.array([b[4] for b in base if b[0] >= np.percentile(bt, 90)]).sum(axis=0)
    print("   ... of the longest tenth of the waves:       " + ", ".join(f"{k} {v / long_k.sum():.2f}" for k, v in zip(KINDS, long_k)))
    for legs in (2, 3, 4, 8):
        st = wave_stats(waves, legs)
        for a, b in zip(st, base):
            assert a[3] == b[3]
        t = np.array([a[0] for a in st])
        c = np.array([a[5] for a in st])
        extra = sum(a[2] for a in st)
        print(f"chained legs <= {legs}: wave-trips {t.sum()} ({t.sum() / bt.sum():.3f})  longest {sorted(t)[-5:]}  p50 {np.median(t):.0f} p90 {np.percentile(t, 90):.0f} "
              f"p99 {np.percentile(t, 99):.0f}   extra legs run by the waves {extra} ({extra / t.sum():.2f} per trip)  modelled issue cost {c.sum():.0f} ({c.sum() / bc.sum():.3f})")
    for dual, legs in ((1, 1), (2, 1), (1, 3)):
        st = wave_stats(waves, legs, dual)
        for a, b in zip(st, base):
            assert a[3] == b[3]
        t = np.array([a[0] for a in st])
        lt = sum(a[1] for a in st)
        print(f"dual look-up ({dual} move(s) finer when refused), legs <= {legs}: wave-trips {t.sum()} ({t.sum() / bt.sum():.3f})  longest {sorted(t)[-5:]}  p50 {np.median(t):.0f} "
              f"p90 {np.percentile(t, 90):.0f} p99 {np.percentile(t, 99):.0f}  lane trips {lt} ({lt / lane_trips:.3f})")
    floors = (0, 1, 2) if MINLEV == 0 else (2, 3)
    for cells, up, legs, floor in [(c, u, l, f) for f in floors for (c, u, l) in ((4, 2.0, 1), (6, 2.0, 1), (8, 2.0, 1), (12, 2.0, 1), (8, 2.0, 3))]:
        st = wave_stats(waves, legs, 0, cells, up, floor)
        for a, b in zip(st, base):
            assert a[3] == b[3]
        t = np.array([a[0] for a in st])
        lt = sum(a[1] for a in st)
        cst = sum(a[5] for a in st)
        print(f"[issue cost {cst / bc.sum():.3f}] cell walks of {cells} cells below level {floor} (back up when cleared by {up}), legs <= {legs}: wave-trips {t.sum()} ({t.sum() / bt.sum():.3f})  longest {sorted(t)[-5:]}  "
              f"p50 {np.median(t):.0f} p90 {np.percentile(t, 90):.0f} p99 {np.percentile(t, 99):.0f}  lane trips {lt} ({lt / lane_trips:.3f})")
    for cells, up, legs, floor in [(c, u, l, f) for f in floors for (c, u, l) in ((4, 2.0, 1), (8, 2.0, 1), (8, 1.0, 1), (8, 4.0, 1), (12, 2.0, 1), (8, 2.0, 3))]:
        st = wave_stats(waves, legs, 0, cells, up, floor, True)
        for a, b in zip(st, base):
            assert a[3] == b[3]
        t = np.array([a[0] for a in st])
        lt = sum(a[1] for a in st)
        cst = sum(a[5] for a in st)
        print(f"[issue cost {cst / bc.sum():.3f}] 2x2-max walks of {cells} segments below level {floor} (back up when cleared by {up}), legs <= {legs}: wave-trips {t.sum()} ({t.sum() / bt.sum():.3f})  longest {sorted(t)[-5:]}  "
              f"p50 {np.median(t):.0f} p90 {np.percentile(t, 90):.0f} p99 {np.percentile(t, 99):.0f}  lane trips {lt} ({lt / lane_trips:.3f})")
    if os.environ.get("COOP", "0") == "0":
        return
    variants = []
    for thresh in (4, 8, 16, 32):
        for rule in (("exit",), ("recent", 2.0), ("recent", 4.0)):
            variants.append((thresh, rule, 1))
    for thresh, rule, ovh in variants:
        ct = []
        rounds = 0
        for (ty, tx, lanes), b in zip(waves, base):
            t, r, res = coop_wave(lanes, thresh, rule, ovh)
            assert res == b[3], (ty, tx)
            ct.append(t)
            rounds += r
        ct = np.array(ct)
        print(f"coop thresh {thresh:2d} seg {str(rule):18s} overhead {ovh}: wave-trips {ct.sum()} ({ct.sum() / bt.sum():.3f})  "
              f"longest {sorted(ct)[-5:]}  p90 {np.percentile(ct, 90):.0f} p99 {np.percentile(ct, 99):.0f}  rounds per wave {rounds / len(ct):.2f}")


if __name__ == "__main__":
    main()


def debug_rounds():
    tiles_x = (cam.width + 7) // 8
    for ty in (150, 160):
        for tx in (100, 298):
            lanes = [make_lane(tx * 8 + (l & 7), ty * 8 + (l >> 3)) for l in range(64)]
            done = [run_alone(l.clone()) for l in lanes if l is not None]
            order = sorted(done, key=lambda d: -d.trips)
            print("tile", ty, tx, "trips", [d.trips for d in order[:8]], "steps", [d.steps for d in order[:8]], "status", [d.status for d in order[:8]])
            # follow the longest ray: where is it after each trip
            L = [l for l in lanes if l is not None][[d.trips for d in done].index(order[0].trips)].clone()
            log = []
            while L.status == 0:
                before = L.steps
                lev = L.lev
                trip(L)
                refresh(L)
                log.append((lev, L.steps - before, min(L.lx, L.ly, L.lz), int(exit_estimate(L))))
            print("   (level, steps taken, binade left, exit estimate) per trip:", log)


def debug_longest():
    tiles_y = (cam.height + 7) // 8
    tiles_x = (cam.width + 7) // 8
    best = []
    for ty in range(0, tiles_y, ROW_STRIDE):
        for tx in range(3, tiles_x, COL_STRIDE):
            lanes = [make_lane(tx * 8 + (l & 7), ty * 8 + (l >> 3)) for l in range(64)]
            live = [l for l in lanes if l is not None]
            if not live:
                continue
            done = [run_alone(l.clone()) for l in live]
            crit = max(range(len(done)), key=lambda i: done[i].trips)
            best.append((done[crit].trips, ty, tx, live[crit]))
    best.sort(key=lambda b: -b[0])
    for trips, ty, tx, lane in best[:4]:
        L = lane.clone()
        log = []
        while L.status == 0:
            before, lev, z0 = L.steps, L.lev, L.z
            a0, g0 = L.attempts, L.groups
            trip(L)
            log.append((lev, L.steps - before, "A" if L.attempts > a0 else "-", "G" if L.groups > g0 else "-", round(z0 - thr[min(int(-L.y), S - 1), min(int(L.x), S - 1)], 1)))
        print("tile", ty, tx, "trips", trips, "steps", L.steps, "dir", round(L.sx, 3), round(L.sy, 3), round(L.sz, 4))
        print("   (level, steps, attempt, group, height above the cell) per trip:", log)
