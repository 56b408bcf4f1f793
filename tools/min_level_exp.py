#!/usr/bin/env python3
"""Finest pyramid level worth an attempt (camera.cpp's hint min_window -> DevFrame::min_level), overridden with the
tool knob HMRM_MIN_LEVEL: kernel ms per workload and level, interleaved.  usage: min_level_exp.py [workloads...]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
for name in sys.argv[1:] or ["C3", "C5", "C2", "C4"]:
    wl = hmrm.synth.WORKLOADS[name]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
    scene = hmrm.Scene(rgb, cmap, wl.scene_params())
    cam = wl.camera()
    levels = ["", "0", "1", "2", "3"]
    times = {l: [] for l in levels}
    for rnd in range(6):
        for l in levels:
            if l:
                os.environ["HMRM_MIN_LEVEL"] = l
            else:
                os.environ.pop("HMRM_MIN_LEVEL", None)
            os.environ["HMRM_TILE_ORDER"] = "1"   # (plain rotation: no calibration noise between the variants)
            times[l].append(scene.bench_kernel_ms(cam, 10))
    print(name, " ".join(f"[{l or 'default'}] {np.median(times[l][1:]):.4f}" for l in levels), flush=True)
    os.environ.pop("HMRM_MIN_LEVEL", None)
    # the pause (groups marched before the next attempt) after a failed height test at the finest level: HMRM_FINEST_PAUSE
    pauses = ["", "0", "1", "3", "6", "8"]
    times = {l: [] for l in pauses}
    for rnd in range(6):
        for l in pauses:
            if l:
                os.environ["HMRM_FINEST_PAUSE"] = l
            else:
                os.environ.pop("HMRM_FINEST_PAUSE", None)
            times[l].append(scene.bench_kernel_ms(cam, 10))
    os.environ.pop("HMRM_FINEST_PAUSE", None)
    print(name, "pause", " ".join(f"[{l or 'default'}] {np.median(times[l][1:]):.4f}" for l in pauses), flush=True)
    scene.close()
