#!/usr/bin/env python3
"""Minimal launch loop for rocprofv3: python tools/prof_run.py <workload | workload/content kind> <variant> <launches>."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
name, variant, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
os.environ["HMRM_KERNEL"] = variant
if "/" in name:
    wl = hmrm.synth.content_workload(*name.split("/"))
    rgb, cmap = wl.maps()
else:
    wl = hmrm.synth.WORKLOADS[name]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
scene = hmrm.Scene(rgb, cmap, wl.scene_params())
# the first 12 launches one at a time: the library settles this camera's launch order on them (tools/pmc_summary.py
# leaves them out of its averages: PMC_SKIP_FIRST)
for _ in range(12):
    scene.bench_kernel_ms(wl.camera(), 1)
print(name, variant, "kernel ms", scene.bench_kernel_ms(wl.camera(), n))
scene.close()
