#!/usr/bin/env python3
"""Dump per-pixel (attempts, groups) maps of the production kernel to gpurun_out/trip_maps.npz
(input of tools/merge_sim.py).  usage: trip_map_dump.py [workloads...]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
os.environ["HMRM_KERNEL"] = "leap"
out = {}
for name in sys.argv[1:] or ["C3", "C5", "C2"]:
    wl = hmrm.synth.WORKLOADS[name]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
    scene = hmrm.Scene(rgb, cmap, wl.scene_params())
    cam = wl.camera()
    os.environ["HMRM_DIAG_ITERS"] = "1"
    _, st, packed, _ = scene.render_stats(cam, per_pixel=True)
    os.environ.pop("HMRM_DIAG_ITERS", None)
    out[name] = packed
    print(name, "attempts", st.leap_attempts, "groups", st.groups, flush=True)
    scene.close()
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/trip_maps.npz", **out)
