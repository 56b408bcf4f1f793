#!/usr/bin/env python3
"""Kernel-time decomposition experiments on C3's scene (GPU)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
wl = hmrm.synth.WORKLOADS["C3"]
rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
scene = hmrm.Scene(rgb, cmap, wl.scene_params())
D = hmrm.degrees_to_rads
def run(label, cam, variants=("leap",)):
    for v in variants:
        os.environ["HMRM_KERNEL"] = v
        _, st, _, _ = scene.render_stats(cam)
        ms = np.median([scene.bench_kernel_ms(cam, 10) for _ in range(5)])
        print(f"{label:34s} {v:6s} {ms:7.3f} ms  steps {st.steps:12d} hits {st.hits:8d} attempts {st.leap_attempts:9d} groups {st.groups:9d}", flush=True)
cam = wl.camera()
run("C3 (spherical)", cam, ("leap",))
c = wl.camera(); c.vang = D(40.0)
run("C3 looking up: all sky", c, ("leap", "simple"))
c = wl.camera(); c.projection = 1; c.hfov = D(90)
run("C3 perspective 90", c)
c = wl.camera(); c.projection = 1; c.hfov = D(90); c.vang = D(40.0)
run("perspective all sky", c, ("leap", "simple"))
c = wl.camera(); c.projection = 1; c.hfov = D(60); c.vang = D(140.0); c.pos[0], c.pos[1], c.pos[2] = 1000.0, -1000.0, 1500.0
run("perspective looking down: all terrain", c, ("leap", "group"))
c = wl.camera(); c.projection = 3; c.ortho_width = 0.9; c.vang = D(180.0); c.hang = 0.0; c.pos[0], c.pos[1], c.pos[2] = 2048.0, -2048.0, 1000.0
run("ortho top-down: all terrain", c, ("leap", "group"))
c = wl.camera(); c.width, c.height = 1920, 1080
run("C3 at 1920x1080", c)
c = wl.camera(); c.width, c.height = 7680, 4320
run("C3 at 7680x4320", c)
