#!/usr/bin/env python3
"""Offline model: what would merging half-empty waves of a workgroup save?  (input: gpurun_out/trip_maps.npz)

Cost model: a wave executes max-over-lanes loop iterations (iterations of a lane ~ attempts + groups).
Waves of a workgroup run in lockstep; at every K-th iteration two waves whose live lanes fit in one wave
are merged (greedy, fullest first).  Prints wave-iterations without and with merging: an UPPER bound of the
gain (no barrier / repack cost, perfect lockstep)."""
import sys
import numpy as np
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trip_maps.npz")
for name in d.files:
    packed = d[name]
    it = (packed >> 16).astype(np.int64) + (packed & 0xffff).astype(np.int64)
    H, W = it.shape
    H8, W8 = H // 8 * 8, W // 8 * 8
    t = it[:H8, :W8].reshape(H8 // 8, 8, W8 // 8, 8).transpose(0, 2, 1, 3).reshape(H8 // 8, W8 // 8, 64)  # [tile_y, tile_x, lane]
    base = int(t.max(axis=2).sum())
    lane_total = int(t.sum())
    print(f"{name}: lane-iterations {lane_total}, wave-iterations {base} (lane utilisation {lane_total / 64 / base:.3f})")
    for (gy, gx) in ((2, 1), (2, 2), (4, 2), (4, 4)):
        for K in (2, 4, 8):
            ty, tx = t.shape[0] // gy * gy, t.shape[1] // gx * gx
            g = t[:ty, :tx].reshape(ty // gy, gy, tx // gx, gx, 64).transpose(0, 2, 1, 3, 4).reshape(-1, gy * gx, 64)
            g = g[g.max(axis=(1, 2)) > 0]
            total = 0
            base_g = int(g.max(axis=2).sum())
            for wg in g:
                waves = [w[w > 0] for w in wg if (w > 0).any()]  # remaining iterations per live lane
                i = 0
                while waves:
                    # run K iterations
                    step = min(K, max(int(w.max()) for w in waves))
                    for _ in range(step):
                        total += len(waves)
                        waves = [w - 1 for w in waves]
                        waves = [w[w > 0] for w in waves]
                        waves = [w for w in waves if len(w)]
                        if not waves:
                            break
                    # merge greedily: fullest with the largest one that still fits
                    waves.sort(key=len, reverse=True)
                    merged = []
                    while waves:
                        a = waves.pop(0)
                        for j in range(len(waves)):
                            if len(a) + len(waves[j]) <= 64:
                                a = np.concatenate([a, waves.pop(j)])
                                break
                        merged.append(a)
                    waves = merged
            print(f"   workgroup {gy}x{gx} waves, merge check every {K}: {total} wave-iterations = {total / base_g:.3f} of {base_g}")
