#!/usr/bin/env python3
"""Spherical frames with and without the table of whole-row miss colours (DevFrame::sky_rows; HMRM_SKY_ROWS=0 turns it
off): kernel ms, interleaved, plain rotation and the calibrated order.  usage: sky_rows_ab.py [workloads...]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
for name in sys.argv[1:] or ["C3"]:
    wl = hmrm.synth.WORKLOADS[name]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
    scene = hmrm.Scene(rgb, cmap, wl.scene_params())
    cam = wl.camera()
    for order in ("1", "2"):
        os.environ["HMRM_TILE_ORDER"] = order
        times = {"0": [], "1": []}
        for rnd in range(9):
            for v in ("0", "1"):
                os.environ["HMRM_SKY_ROWS"] = v
                if order == "2" and rnd == 0:
                    for _ in range(12):
                        scene.bench_kernel_ms(cam, 1)   # (the launch order settles)
                times[v].append(scene.bench_kernel_ms(cam, 10))
        print(f"{name} order mode {order}: without the table {np.median(times['0'][1:]):.4f} ms, with it {np.median(times['1'][1:]):.4f} ms", flush=True)
    scene.close()
