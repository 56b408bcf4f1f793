"""CPU model of the production kernel's traversal (csrc/render_fast.hip) against plain sequential marching
(main/hmap.cpp:1000-1038): window maxima, estimated jump lengths, exact verification of the end point, binade stays
counted by tests/leap_model.py, jumps that end in one real step (HMRM_CROSS), groups of real steps otherwise.
Same hit cell, same step count and the same final position (bit for bit) for every ray -- with ANY level policy, which is why the policy here is a crude one.
No GPU: this checks the scheme; tests/test_parity_gpu.py checks the kernel."""
import math
import random

import numpy as np

import leap_model as L

TOP = 7  # levels 0..6 are windows of 4 << l cells, 7 is the whole map


def trunc_cell(v):
    q = int(v)  # C truncation (hmap.cpp:1001-1004); callers keep |v| small
    return q


def march_plain(p, s, thr, gw, cap):
    """The reference loop: returns (hit, cell, steps, final position)."""
    x, y, z = p
    h, w = thr.shape
    steps = 0
    while True:
        gx, gy = trunc_cell(x / gw), trunc_cell(-y / gw)
        if not (0 <= gx < w and 0 <= gy < h) or x / gw <= -1.0 or -y / gw <= -1.0:
            return False, None, steps, (x, y, z)
        if steps >= cap:
            return False, "cap", steps, (x, y, z)
        steps += 1
        if z < thr[gy, gx]:
            return True, (gy, gx), steps, (x, y, z)
        x, y, z = x + s[0], y + s[1], z + s[2]


def in_grid(x, y, gw, w, h):
    qx, qy = x / gw, -y / gw
    return qx > -1.0 and qy > -1.0 and 0 <= trunc_cell(qx) < w and 0 <= trunc_cell(qy) < h


def march_leaping(p, s, thr, gw, cap, rng):
    x, y, z = p
    sx, sy, sz = s
    h, w = thr.shape
    ax, ay, az = L.Axis(), L.Axis(), L.Axis()
    offx, offy = sx < 0.0, sy > 0.0
    steps = jumps = groups = 0
    lev = TOP
    while True:
        moved = False
        if in_grid(x, y, gw, w, h) and rng.random() < 0.8:  # (the kernel pauses attempts now and then: any schedule is valid)
            for a, pp, ss in ((ax, x, sx), (ay, y, sy), (az, z, sz)):
                if a.left < 0:
                    L.axis_refresh(a, pp, ss, rcp_err=rng.uniform(-1, 1) * 2.0 ** -24, short=0)
            gx, gy = trunc_cell(x / gw), trunc_cell(-y / gw)
            exact = ax.left >= 0 and ay.left >= 0 and az.left >= 0
            top = lev == TOP
            sparse = lev < 1
            hs = 28 if top else lev + (1 if sparse else 0)
            back = 1 if sparse else 3
            ix = max((gx >> hs) - (back if offx else 0), 0)
            iy = max((gy >> hs) - (back if offy else 0), 0)
            wx0, wy0 = ix << hs, iy << hs
            wcells = (1 << 30) if top else (4 << lev)
            wsx, wsy = min(wcells, w - wx0), min(wcells, h - wy0)
            m = float(thr[wy0:wy0 + wsy, wx0:wx0 + wsx].max())
            ok = False
            if exact and z >= m:
                ex = float(wx0 if offx else wx0 + wsx) * gw
                ey = -float(wy0 if offy else wy0 + wsy) * gw
                room = min(abs((ex - x) * ax.rdel), abs((ey - y) * ay.rdel))
                room_z = (m - z) * az.rdel if sz < 0.0 else 2.0 ** 40
                room = min(room, room_z)
                # only speed may depend on the estimates: spoil them, the verification below has to hold the line
                room *= rng.choice((1.0, 1.0, rng.uniform(0.3, 2.5)))
                n = min(L.cvt_i32_sat(room * 0.998), cap - steps) - 1
                n = min(n, min(ax.left, ay.left, az.left) + 1)
                if n >= 2:
                    k = float(n - 1)
                    xn, yn, zn = (x + k * ax.delta) + sx, (y + k * ay.delta) + sy, (z + k * az.delta) + sz
                    gxn, gyn = (trunc_cell(xn / gw), trunc_cell(-yn / gw)) if (xn / gw > -1.0 and -yn / gw > -1.0) else (-5, -5)
                    ok = 0 <= gxn - wx0 < wsx and 0 <= gyn - wy0 < wsy and zn >= m
                    if ok:
                        x, y, z = xn, yn, zn
                        steps += n
                        jumps += 1
                        for a in (ax, ay, az):
                            a.left -= n
                        moved = True
            # any level sequence is valid: a crude random walk over the levels
            lev = min(TOP, lev + 1) if ok and rng.random() < 0.5 else (max(0, lev - rng.choice((1, 2))) if not ok else lev)
        if moved:
            continue
        groups += 1
        for _ in range(4):  # a group of real steps, tests in order
            if not in_grid(x, y, gw, w, h):
                return False, None, steps, (x, y, z), jumps, groups
            if steps >= cap:
                return False, "cap", steps, (x, y, z), jumps, groups
            steps += 1
            gx, gy = trunc_cell(x / gw), trunc_cell(-y / gw)
            if z < thr[gy, gx]:
                return True, (gy, gx), steps, (x, y, z), jumps, groups
            x, y, z = x + sx, y + sy, z + sz
        for a in (ax, ay, az):
            a.left -= 4


def _scene(rng, w, h):
    base = rng.random((h // 8 + 2, w // 8 + 2)) * 6.0
    t = np.kron(base, np.ones((8, 8)))[:h, :w] + rng.random((h, w)) * 1.5
    t[rng.integers(0, h, 6), rng.integers(0, w, 6)] += 9.0  # a few spikes
    return np.ascontiguousarray(t)


def test_leaping_traversal_equals_sequential_marching():
    nrng = np.random.default_rng(5)
    rng = random.Random(5)
    total_jumps = total_steps = leaped_rays = crossing_rays = 0
    for scene in range(12):
        w, h = rng.choice(((96, 64), (160, 160), (257, 131)))
        thr = _scene(nrng, w, h)
        gw = rng.choice((1.0, 1.0, 0.3, 0.05, 3.0))
        for ray in range(150):
            # a start inside the grid, well above or just above the terrain, a direction that descends slowly (long marches)
            cx, cy = rng.uniform(0.5, w - 0.5), rng.uniform(0.5, h - 0.5)
            ang = rng.uniform(0, 2 * math.pi)
            step = rng.choice((0.11, 0.25, 0.5)) * gw
            dz = -rng.choice((0.004, 0.02, 0.08, 0.3)) if rng.random() < 0.85 else rng.uniform(0.0, 0.05)
            norm = math.sqrt(1.0 + dz * dz)
            s = (step * math.cos(ang) / norm, step * math.sin(ang) / norm, step * dz / norm)
            if ray % 9 == 0:
                s = (s[0], 0.0, s[2])  # a coordinate that never moves
            p = (cx * gw, -cy * gw, float(thr.max()) + rng.uniform(0.01, 6.0))
            if ray % 7 == 3:  # climbing out of a valley: starts BELOW most window maxima and ends above them
                dz = rng.uniform(0.05, 0.4)
                norm = math.sqrt(1.0 + dz * dz)
                s = (step * math.cos(ang) / norm, step * math.sin(ang) / norm, step * dz / norm)
                p = (cx * gw, -cy * gw, float(thr[int(cy), int(cx)]) + rng.uniform(0.01, 0.5))
            cap = 200000
            want = march_plain(p, s, thr, gw, cap)
            got = march_leaping(p, s, thr, gw, cap, rng)
            # same verdict, same cell, same number of steps -- and the same position, bit for bit: both walked the
            # reference's sequence p += s (struct comparison of floats is exact)
            assert got[:4] == want, (scene, ray, gw, p, s, got, want)
            total_jumps += got[4]
            total_steps += want[2]
            leaped_rays += got[4] > 0
            crossing_rays += got[4] > 2
    # the model did leap: most rays jumped, and jumps carried the bulk of the steps' work
    assert leaped_rays > 1200 and crossing_rays > 600 and total_steps > 400000 and total_jumps > 8000
