#!/usr/bin/env python3
"""Minimal launch loop for rocprofv3: python tools/prof_run.py <workload> <variant> <launches>."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
name, variant, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
os.environ["HMRM_KERNEL"] = variant
wl = hmrm.synth.WORKLOADS[name]
rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
scene = hmrm.Scene(rgb, cmap, wl.scene_params())
print(name, variant, "kernel ms", scene.bench_kernel_ms(wl.camera(), n))
scene.close()
