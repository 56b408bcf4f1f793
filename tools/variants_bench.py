#!/usr/bin/env python3
"""Kernel-variant A/B on one device: interleaved rounds, median/min kernel ms (HIP events)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
variants = os.environ.get("VARIANTS", "leap,group,simple").split(",")
for name in sys.argv[1:] or ["C3"]:
    wl = hmrm.synth.WORKLOADS[name] if "/" not in name else hmrm.synth.content_workload(*name.split("/"))  # e.g. C3/white
    rgb, cmap = wl.maps()
    scene = hmrm.Scene(rgb, cmap, wl.scene_params())
    cam = wl.camera()
    os.environ["HMRM_KERNEL"] = "leap"
    _, st, _, _ = scene.render_stats(cam)
    print(f"{name} diag: attempts {st.leap_attempts} leaps {st.leaps} groups {st.groups} leaped_steps {st.leaped_steps} "
          f"of {st.steps}; entering rays ~{st.hits}+; per hit-ray: attempts {st.leap_attempts/max(st.hits,1):.1f} "
          f"leaps {st.leaps/max(st.hits,1):.1f} groups {st.groups/max(st.hits,1):.1f}")
    times = {v: [] for v in variants}
    for rnd in range(7):
        for v in variants:
            os.environ["HMRM_KERNEL"] = v
            times[v].append(scene.bench_kernel_ms(cam, 5))
    for v in variants:
        t = np.array(times[v][1:])
        print(f"{name} {v:7s} median {np.median(t):8.3f} ms  min {t.min():8.3f} ms  "
              f"-> {st.steps / np.median(t) / 1e6:10.1f} Gsteps/s  ({st.steps} steps, {st.rays} rays)", flush=True)
    scene.close()
