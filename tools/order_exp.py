#!/usr/bin/env python3
"""Launch-order experiments: kernel ms of a workload with explicit tile-row pieces started first (HMRM_TILE_SEGMENTS,
a tool knob: "b0:c0,b1:c1,b2:c2" in 16-row tile rows) against the default rotation.  usage: order_exp.py WORKLOAD "segs" "segs" ..."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
wl = hmrm.synth.WORKLOADS[sys.argv[1]]
variants = [""] + sys.argv[2:]
rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
scene = hmrm.Scene(rgb, cmap, wl.scene_params())
cam = wl.camera()
ref = scene.render(cam)
times = {v: [] for v in variants}
for rnd in range(6):
    for v in variants:
        if v:
            os.environ["HMRM_TILE_SEGMENTS"] = v
        else:
            os.environ.pop("HMRM_TILE_SEGMENTS", None)
        if rnd == 0:
            assert np.array_equal(scene.render(cam), ref), v  # (order never changes a pixel)
        times[v].append(scene.bench_kernel_ms(cam, 10))
for v in variants:
    t = np.array(times[v][1:])
    print(f"{wl.name} [{v or 'default rotation'}]: median {np.median(t):.4f} ms  min {t.min():.4f} ms", flush=True)
scene.close()
