#!/usr/bin/env python3
"""Kernel ms against HMRM_PASS1_TRIPS (two-pass hand-over point; 0 = single pass), interleaved rounds.
usage: trips_sweep.py <workloads...>   (TRIPS="0,8,16,24,32,48" env to choose)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
trips = [int(t) for t in os.environ.get("TRIPS", "0,8,12,16,24,32,48").split(",")]
for name in sys.argv[1:] or ["C3"]:
    wl = hmrm.synth.WORKLOADS[name]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
    scene = hmrm.Scene(rgb, cmap, wl.scene_params())
    cam = wl.camera()
    times = {t: [] for t in trips}
    for rnd in range(6):
        for t in trips:
            os.environ["HMRM_PASS1_TRIPS"] = str(t)
            times[t].append(scene.bench_kernel_ms(cam, 10))
    for t in trips:
        a = np.array(times[t][1:])
        os.environ["HMRM_PASS1_TRIPS"] = str(t)
        _, st, _, _ = scene.render_stats(cam)
        print(f"{name} trips {t:3d}: median {np.median(a):.4f} ms  min {a.min():.4f} ms   "
              f"(attempts {st.leap_attempts} groups {st.groups})", flush=True)
    scene.close()
