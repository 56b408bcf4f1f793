#!/usr/bin/env python3
"""Kernel time of the C3 scene scaled to other grid widths (GWM 0 / 1 / 2 cell-coordinate paths), and of REFDEF: the
reference's own operating point (sample_config.txt:5-7: grid_width 0.01, step_dist 0.05 = 5 cells per step) on the C5
frame.  Same cells at every grid width (positions, heights and the step scale with it: synth.grid_workload)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
synth = hmrm.synth
rgb, cmap = synth.synth_maps(4096)
base_ms = {}
for base in (sys.argv[1:] or ["C3", "C5"]):
    for gw in (1.0, 0.5, 0.05, 0.3, 3.0):
        wl = synth.grid_workload(base, gw)
        cam = wl.camera()
        scene = hmrm.Scene(rgb, cmap, wl.scene_params())
        _, st, _, _ = scene.render_stats(cam)
        for _ in range(12):
            scene.bench_kernel_ms(cam, 1)
        ms = float(np.median([scene.bench_kernel_ms(cam, 10) for _ in range(7)]))
        base_ms.setdefault(base, ms)
        print(f"{base} grid_width {gw:5.2f}: {ms:.4f} ms ({ms / base_ms[base]:.3f} x gw 1)  steps {st.steps} attempts {st.leap_attempts} "
              f"jumps {st.leaps} groups {st.groups}", flush=True)
        scene.close()
wl = synth.WORKLOADS["REFDEF"]
cam = wl.camera()
scene = hmrm.Scene(rgb, cmap, wl.scene_params())
fb, st, _, _ = scene.render_stats(cam)
for v in ("leap", "group"):
    os.environ["HMRM_KERNEL"] = v
    for _ in range(12):
        scene.bench_kernel_ms(cam, 1)
    assert np.array_equal(scene.render(cam), fb)
    ms = float(np.median([scene.bench_kernel_ms(cam, 10) for _ in range(7)]))
    print(f"REFDEF (gw 0.01, step_dist 0.05 = 5 cells, perspective 3840x2160, 4096^2) {v:5s}: {ms:.4f} ms  steps {st.steps} "
          f"attempts {st.leap_attempts} jumps {st.leaps} groups {st.groups} leaped {st.leaped_steps}", flush=True)
os.environ["HMRM_KERNEL"] = "leap"
scene.close()
