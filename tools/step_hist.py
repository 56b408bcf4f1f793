#!/usr/bin/env python3
"""Distribution of per-ray step counts and per-tile imbalance for a workload (GPU)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
for name in sys.argv[1:] or ["C3"]:
    wl = hmrm.synth.WORKLOADS[name]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
    scene = hmrm.Scene(rgb, cmap, wl.scene_params())
    cam = wl.camera()
    fb, st, steps, entry = scene.render_stats(cam, per_pixel=True)
    s = steps.astype(np.int64)
    print(name, "rays", st.rays, "steps", st.steps, "hits", st.hits, "kernel_ms(stats)", hmrm.lib.lib.hmrm_last_kernel_ms())
    print(" entering rays:", int((s > 0).sum()), "mean steps of entering:", s[s > 0].mean(), "max", s.max())
    print(" percentiles of entering rays:", np.percentile(s[s > 0], [10, 50, 90, 99, 99.9]).tolist())
    H, W = s.shape
    t = s[:H // 8 * 8, :W // 8 * 8].reshape(H // 8, 8, W // 8, 8)
    wave_max = t.max(axis=(1, 3)); wave_sum = t.sum(axis=(1, 3))
    print(" wave(8x8) lane utilisation = sum/(64*max):", wave_sum.sum() / (64 * wave_max.sum()))
    print(" sum of wave max (serial iterations over all waves):", int(wave_max.sum()), " max wave:", int(wave_max.max()))
    rows = s.sum(axis=1)
    print(" row-sum steps: first nonzero row", int(np.argmax(rows > 0)), "peak row", int(rows.argmax()), "peak/mean", rows.max() / rows.mean())
    print(" kernel ms:", scene.bench_kernel_ms(cam, 5))
