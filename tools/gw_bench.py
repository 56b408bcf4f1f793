#!/usr/bin/env python3
"""Kernel time of the C3 scene scaled to other grid widths (GWM 0 / 1 / 2 cell-coordinate paths)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
wl = hmrm.synth.WORKLOADS["C3"]
rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
for gw in (1.0, 0.5, 0.05, 0.3, 3.0):
    s = wl.map_size * gw
    params = hmrm.SceneParams.make(0.0, s / 16.0, grid_width=gw)
    cam = wl.camera()
    cam.pos[0], cam.pos[1], cam.pos[2] = -s / 8.0, s / 8.0, s / 4.0
    cam.step_dist = wl.step_dist * gw
    scene = hmrm.Scene(rgb, cmap, params)
    _, st, _, _ = scene.render_stats(cam)
    ms = np.median([scene.bench_kernel_ms(cam, 10) for _ in range(5)])
    print(f"grid_width {gw:5.2f}: {ms:.3f} ms  steps {st.steps} attempts {st.leap_attempts} groups {st.groups}", flush=True)
    scene.close()
