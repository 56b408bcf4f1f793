#!/usr/bin/env python3
"""Offline model: several pixels per lane, one after the other (input: gpurun_out/trip_maps.npz from trip_map_dump.py).

Today a wave owns 8x8 pixels and runs max-over-lanes loop trips.  If a lane marched P pixels in sequence (the next
ray starts when the previous one ends, other lanes do not wait), the wave would run max-over-lanes of the SUM of its
lanes' trips: lanes average out.  Which pixels share a lane decides how much: rows mirrored inside the tile cancel the
smooth trend of the march length with the screen row.  Prints wave-trips (work) and the longest wave (critical path)."""
import sys
import numpy as np
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trip_maps.npz")
for name in d.files:
    packed = d[name]
    it = (packed >> 16).astype(np.int64) + (packed & 0xffff).astype(np.int64)
    H, W = it.shape
    def waves8x8(a):
        h, w = a.shape[0] // 8 * 8, a.shape[1] // 8 * 8
        return a[:h, :w].reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(h // 8, w // 8, 64)
    base = waves8x8(it).max(axis=2)
    print(f"{name}: lane-trips {int(it.sum())}; one pixel per lane: wave-trips {int(base.sum())}, longest wave {int(base.max())}, "
          f"lane utilisation {it.sum() / 64 / base.sum():.3f}")
    def report(label, lane_sums):  # lane_sums: [..., 64] per wave
        wt = lane_sums.max(axis=-1)
        print(f"   {label:58s} wave-trips {int(wt.sum()):8d} = {wt.sum() / base.sum():.3f} of today, longest wave {int(wt.max()):4d}, "
              f"lane utilisation {lane_sums.sum() / 64 / wt.sum():.3f}")
    for P in (2, 4):
        th = 8 * P  # tile height
        h, w = H // th * th, W // 8 * 8
        t = it[:h, :w].reshape(h // th, th, w // 8, 8)  # [tile_y, row, tile_x, col]
        # (a) lane (c, r) takes rows r, r+8, ...
        a = t.reshape(h // th, P, 8, w // 8, 8).sum(axis=1).transpose(0, 2, 1, 3).reshape(h // th, w // 8, 64)
        report(f"{P} pixels per lane, rows r, r+8, ..", a)
        # (b) mirrored: rows r and th-1-r (P = 2); for P = 4: r, 15-r, 16+r, 31-r
        idx = []
        for r in range(8):
            rows = [r, th - 1 - r] if P == 2 else [r, 15 - r, 16 + r, 31 - r]
            idx.append(rows)
        b = np.stack([t[:, rows].sum(axis=1) for rows in idx], axis=1)  # [tile_y, 8, tile_x, 8]
        report(f"{P} pixels per lane, rows mirrored inside the tile", b.transpose(0, 2, 1, 3).reshape(h // th, w // 8, 64))
        # (c) adjacent rows 2r, 2r+1 (P = 2) / 4r..4r+3 (P = 4)
        c = t.reshape(h // th, 8, P, w // 8, 8).sum(axis=2).transpose(0, 2, 1, 3).reshape(h // th, w // 8, 64)
        report(f"{P} pixels per lane, adjacent rows", c)
    # (d) the second pixel half a frame away (rows y and y + H/2): pairs a long tile with a short one
    h2 = H // 16 * 8
    far = it[:h2] + it[h2:2 * h2]
    report("2 pixels per lane, rows y and y + H/2", waves8x8(far))
