#!/usr/bin/env python3
"""Per-wave duration (s_memtime cycles, instrumented kernel) vs iterations, C3."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
os.environ["HMRM_KERNEL"] = "leap"
wl = hmrm.synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
scene = hmrm.Scene(rgb, cmap, wl.scene_params())
cam = wl.camera()
def diag(mode):
    os.environ["HMRM_DIAG_ITERS"] = str(mode)
    scene.render_stats(cam, per_pixel=True)
    _, st, out, _ = scene.render_stats(cam, per_pixel=True)
    os.environ.pop("HMRM_DIAG_ITERS")
    return out
packed = diag(1); cyc = diag(2).astype(np.int64); t0 = diag(3).astype(np.int64)
it = (packed >> 16).astype(np.int64) + (packed & 0xffff).astype(np.int64)
H, W = it.shape
def waves(a, f): return f(a[:H // 8 * 8, :W // 8 * 8].reshape(H // 8, 8, W // 8, 8), axis=(1, 3))
wit, wcyc, wt0 = waves(it, np.max), waves(cyc, np.max), waves(t0, np.min)
print("kernel ms (stats variant):", hmrm.lib.lib.hmrm_last_kernel_ms())
print("wave cycles: max %d  p99.9 %d  p99 %d  median(heavy) %d ; sum %.3e" % (wcyc.max(), np.percentile(wcyc, 99.9), np.percentile(wcyc, 99), np.median(wcyc[wit > 0]), wcyc.sum()))
for lo, hi in ((0, 1), (1, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 4096)):
    m = (wit >= lo) & (wit < hi)
    if m.any(): print(f" waves with {lo:4d}..{hi:4d} iterations: n={int(m.sum()):7d}  cycles/wave median {np.median(wcyc[m]):9.0f}  cycles/iteration {np.median(wcyc[m] / np.maximum(wit[m], 1)):8.0f}")
span = (wt0.max() - wt0.min()) & 0xffffffff
print("start-time span (cycles, mod 2^32):", span, " latest start + its duration vs earliest start:", int(((wt0 - wt0.min()) + wcyc).max()))
i = np.unravel_index(np.argmax((wt0 - wt0.min()) + wcyc), wcyc.shape)
print(" last-finishing wave: tile", i, "iterations", int(wit[i]), "cycles", int(wcyc[i]), "start offset", int(wt0[i] - wt0.min()))
# where the heaviest waves' time goes: cycles waiting for the pyramid look-up (mode 17) and for the group's
# height loads (mode 18), each forced to complete right after its issue (so nothing overlaps it)
wl17, wl18 = waves(diag(17).astype(np.int64), np.max), waves(diag(18).astype(np.int64), np.max)
cyc17, cyc18 = waves(cyc, np.max), None
order = np.argsort(wcyc, axis=None)[::-1][:8]
print("heaviest waves: iterations, cycles (plain run), pyramid-load wait, group-load wait (instrumented runs)")
for o in order:
    i = np.unravel_index(o, wcyc.shape)
    print(f"  tile {i}: iterations {int(wit[i]):4d}  cycles {int(wcyc[i]):8d}  pyramid wait {int(wl17[i]):8d}  group wait {int(wl18[i]):8d}")
m = wit >= 32
print("waves with >= 32 iterations: cycles %.3e  pyramid wait %.3e  group wait %.3e" % (wcyc[m].sum(), wl17[m].sum(), wl18[m].sum()))
