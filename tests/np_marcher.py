"""A second, independent restatement of the reference's hot path in numpy
(vectorised over pixels), used only to cross-check the C oracle on small cases.
numpy's float64 ufuncs are plain IEEE operations (no FMA contraction), python's
math.sin/cos/tan are glibc's.  References: main/hmap.cpp:661-672, :952-1058,
src/*.cpp (see oracle/hmrm_oracle.c for the line-by-line citations)."""
import math

import numpy as np


def _v(x, y, z):
    return np.array([x, y, z], dtype=np.float64)


def _cross(a, b):
    return _v(a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1])


def _dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def render(cam, params, heights, cmap, step_cap=1 << 26):
    """cam: hmrm.Camera, params: hmrm.SceneParams, heights: HxW float64 (heightmap_buf),
    cmap: HxWx4 uint8 -> (frame HxWx4 uint8, steps HxW int64, entry_d HxW float64)."""
    W, H = cam.width, cam.height
    mh, mw = heights.shape
    gw, sd = params.grid_width, cam.step_dist
    pos = _v(cam.pos[0], cam.pos[1], cam.pos[2])
    hang, vang = cam.hang, cam.vang
    look = _v(math.sin(vang) * math.cos(hang), math.sin(vang) * math.sin(hang), math.cos(vang))
    up_vang = vang - (math.pi / 2.0)
    up = _v(math.sin(up_vang) * math.cos(hang), math.sin(up_vang) * math.sin(hang), math.cos(up_vang))
    ar = float(W) / H
    with np.errstate(all="ignore"):
        px = np.arange(W, dtype=np.float64)[None, :].repeat(H, 0)
        py = np.arange(H, dtype=np.float64)[:, None].repeat(W, 1)
        w = px / np.float64(W - 1)
        h = py / np.float64(H - 1)
        if cam.projection == 1:
            hpw = math.tan(cam.hfov / 2.0)
            hph = hpw / ar
            right = _cross(look, up)
            right = right * (1.0 / math.sqrt(_dot(right, right)))
            ul = ((pos + look) + hph * up) - hpw * right
            ll = ((pos + look) - hph * up) - hpw * right
            ur = ((pos + look) + hph * up) + hpw * right
            pr, pd = ur - ul, ll - ul
            v = [((ul[i] + w * pr[i]) + h * pd[i]) - pos[i] for i in range(3)]
            inv = 1.0 / np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
            d = [v[i] * inv for i in range(3)]
            o = [np.full((H, W), pos[i]) for i in range(3)]
        elif cam.projection == 2:
            vfov = cam.hfov / ar
            ul_hang = hang + (cam.hfov / 2.0)
            ul_vang = vang - (vfov / 2.0)
            ha = ul_hang - w * cam.hfov
            va = ul_vang + h * vfov
            sin = np.vectorize(math.sin, otypes=[np.float64])
            cos = np.vectorize(math.cos, otypes=[np.float64])
            ha_safe, va_safe = np.where(np.isfinite(ha), ha, 0.0), np.where(np.isfinite(va), va, 0.0)
            sha, cha = np.where(np.isfinite(ha), sin(ha_safe), np.nan), np.where(np.isfinite(ha), cos(ha_safe), np.nan)
            sva, cva = np.where(np.isfinite(va), sin(va_safe), np.nan), np.where(np.isfinite(va), cos(va_safe), np.nan)
            d = [sva * cha, sva * sha, cva + 0.0 * w]
            o = [np.full((H, W), pos[i]) for i in range(3)]
        else:
            lk = look.astype(np.float32).astype(np.float64)
            u = up.astype(np.float32).astype(np.float64)
            right = _cross(lk, u)
            ow = cam.ortho_width
            ul = (pos - ((W / 2.0) * ow) * right) + ((H / 2.0) * ow) * u
            pr = (W * ow) * right
            pd = (H * ow) * (-u)
            o = [(ul[i] + w * pr[i]) + h * pd[i] for i in range(3)]
            d = [np.full((H, W), lk[i]) for i in range(3)]

        c0 = _v(0.0, 0.0, params.min_height)
        c1 = _v(c0[0] + mw * gw, c0[1] - mh * gw, params.max_height)

        # distance(): AABB.cpp:49-77 with its early returns as masks
        inf = np.inf
        lo = np.full((H, W), -inf)
        hi = np.full((H, W), inf)
        dead = np.zeros((H, W), dtype=bool)
        for i in range(3):
            dl = (c0[i] - o[i]) / d[i]
            dh = (c1[i] - o[i]) / d[i]
            swap = dl > dh
            dl, dh = np.where(swap, dh, dl), np.where(swap, dl, dh)
            dead |= (~dead) & ((dh < lo) | (dl > hi))
            lo = np.where((~dead) & (dl > lo), dl, lo)
            hi = np.where((~dead) & (dh < hi), dh, hi)
        dist = np.where(dead | (lo > hi), inf, lo)
        hit = ~((dist == inf) | (dist < 0.0))

        dd = np.where(hit, dist, 0.0)
        x = o[0] + dd * d[0]
        y = o[1] + dd * d[1]
        z = o[2] + dd * d[2]
        nudge = gw * 0.01
        x = x + nudge * d[0]
        y = y + nudge * d[1]
        z = z + nudge * d[2]
        sx, sy, sz = sd * d[0], sd * d[1], sd * d[2]

        frame = np.zeros((H, W, 4), dtype=np.uint8)
        steps = np.zeros((H, W), dtype=np.int64)
        real_hit = np.zeros((H, W), dtype=bool)
        active = hit.copy()
        flat_h = heights.reshape(-1)
        flat_c = cmap.reshape(-1, 4)
        bg = np.array([cam.bg_r, cam.bg_g, cam.bg_b, 255], dtype=np.uint8)
        it = 0
        while active.any():
            qx = (x - c0[0]) / gw
            qy = -(y - c0[1]) / gw
            inb = (qx > -1.0) & (qx < mw) & (qy > -1.0) & (qy < mh)  # == the int tests, see render.hip
            active &= inb
            if it >= step_cap:
                break
            gx = np.where(active, qx, 0.0).astype(np.int64)
            gy = np.where(active, qy, 0.0).astype(np.int64)
            cell = gx + gy * mw
            hz = flat_h[cell]
            steps += active
            now = active & (z < hz + c0[2])
            if now.any():
                col = flat_c[cell[now]]
                col = np.where(col[:, 3:4] == 0, bg[None, :], col)
                col[:, 3] = 255
                frame[now] = col
            real_hit |= now
            active &= ~now
            x = np.where(active, x + sx, x)
            y = np.where(active, y + sy, y)
            z = np.where(active, z + sz, z)
            it += 1

        miss = ~real_hit
        dz = d[2]
        skyv = miss & (dz > 0.0)
        r_ = 220.0 * (dz * dz) + float(cam.bg_r)
        g_ = 240.0 * (dz * dz) + float(cam.bg_g)
        b_ = 255.0 * dz + float(cam.bg_b)
        sky = np.stack([np.floor(np.clip(np.where(skyv, c, 0.0), 0.0, 255.0)).astype(np.uint8) for c in (r_, g_, b_)]
                       + [np.full((H, W), 255, dtype=np.uint8)], axis=2)
        frame[skyv] = sky[skyv]
        frame[miss & ~skyv] = bg
    return frame, steps, dist
