#!/usr/bin/env python3
"""Launch-order policies side by side (interleaved): HMRM_TILE_ORDER=1 plain rotation, 2 rotation calibrated from
measurement (launch_order.cpp plan_order_from_measurement; HMRM_ORDER_VERBOSE=1 reports each calibration on stderr).
BASELINE workloads and other cameras over the 4096^2 scene.  usage: order_modes.py [workloads...]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
import torch
D = hmrm.degrees_to_rads
os.environ["HMRM_ORDER_VERBOSE"] = "1"
POLICIES = [("rotation", "1"), ("calibrated", "2")]
def bench(scene, label, cam):
    ref = None
    times = {p: [] for p, _ in POLICIES}
    for rnd in range(6):
        for p, mode in POLICIES:
            os.environ["HMRM_TILE_ORDER"] = mode
            if rnd == 0:
                for _ in range(9):  # (calibration: plain, then up to three trials measured twice, settled)
                    fb = scene.render(cam)
                    scene.bench_kernel_ms(cam, 3)
                assert ref is None or np.array_equal(fb, ref), (label, p)
                ref = fb
            times[p].append(scene.bench_kernel_ms(cam, 10))
    print(f"{label:40s} " + "  ".join(f"{p}: {np.median(times[p][1:]):.4f}" for p, _ in POLICIES), flush=True)
names = sys.argv[1:] or ["C3", "C5", "C2", "C4"]
for name in names:
    wl = hmrm.synth.WORKLOADS[name]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
    scene = hmrm.Scene(rgb, cmap, wl.scene_params())
    bench(scene, name, wl.camera())
    if name in ("C3", "C5"):
        for k in (8, 24, 40):
            bench(scene, f"{name} orbit frame {k}/64", wl.camera(k, 64))
    if name == "C3":
        c = wl.camera(); c.projection = 1; c.hfov = D(90)
        bench(scene, "C3 scene, perspective 90", c)
        c = wl.camera(); c.projection = 1; c.hfov = D(60); c.vang = D(140.0); c.pos[0], c.pos[1], c.pos[2] = 1000.0, -1000.0, 1500.0
        bench(scene, "perspective looking down: all terrain", c)
        c = wl.camera(); c.projection = 3; c.ortho_width = 0.9; c.vang = D(180.0); c.hang = 0.0; c.pos[0], c.pos[1], c.pos[2] = 2048.0, -2048.0, 1000.0
        bench(scene, "ortho top-down: all terrain", c)
        c = wl.camera(); c.width, c.height = 1920, 1080
        bench(scene, "C3 at 1920x1080", c)
        c = wl.camera(); c.vang = D(100.0)
        bench(scene, "C3 vang 100", c)
        c = wl.camera(); c.vang = D(135.0); c.pos[2] = 600.0
        bench(scene, "C3 vang 135 from z 600", c)
    scene.close()
