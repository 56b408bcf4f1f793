#!/usr/bin/env python3
"""Where does the production kernel spend its iterations?  Per-pixel (attempts, groups)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
os.environ["HMRM_KERNEL"] = "leap"
for name in sys.argv[1:] or ["C3"]:
    wl = hmrm.synth.WORKLOADS[name]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
    scene = hmrm.Scene(rgb, cmap, wl.scene_params())
    cam = wl.camera()
    os.environ.pop("HMRM_DIAG_ITERS", None)
    _, st, steps, _ = scene.render_stats(cam, per_pixel=True)
    os.environ["HMRM_DIAG_ITERS"] = "1"
    _, st2, packed, _ = scene.render_stats(cam, per_pixel=True)
    os.environ.pop("HMRM_DIAG_ITERS", None)
    att, grp = (packed >> 16).astype(np.int64), (packed & 0xffff).astype(np.int64)
    it = att + grp
    print(name, "attempts", st2.leap_attempts, "leaps", st2.leaps, "groups", st2.groups, "leaped", st2.leaped_steps, "of", st2.steps)
    e = steps > 0
    print(" per entering ray: attempts %.1f groups %.1f ; iterations pct [50,90,99,99.9,max]:" % (att[e].mean(), grp[e].mean()),
          np.percentile(it[e], [50, 90, 99, 99.9]).tolist(), int(it.max()))
    H, W = it.shape
    t = it[:H // 8 * 8, :W // 8 * 8].reshape(H // 8, 8, W // 8, 8)
    wmax = t.max(axis=(1, 3)); wsum = t.sum(axis=(1, 3))
    print(" wave iterations: sum of wave-max %d, max wave %d, lane utilisation %.3f" % (wmax.sum(), wmax.max(), wsum.sum() / (64.0 * wmax.sum())))
    worst = np.unravel_index(np.argmax(it), it.shape)
    print(" worst pixel", worst, "steps", int(steps[worst]), "attempts", int(att[worst]), "groups", int(grp[worst]))
    rows = wmax.sum(axis=1)
    top = np.argsort(rows)[-5:][::-1]
    print(" heaviest wave-rows (8px):", [(int(r) * 8, int(rows[r])) for r in top])
    # histogram of wave-max
    print(" wave-max histogram:", np.histogram(wmax[wmax > 0], bins=[1, 8, 16, 32, 64, 128, 256, 512, 1024, 4096])[0].tolist())
    scene.close()
