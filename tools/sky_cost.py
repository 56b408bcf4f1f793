#!/usr/bin/env python3
"""Launch loop of C3's scene with the camera looking up (every ray misses the box): under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES ...` this gives the instructions a sky wave costs.
usage: sky_cost.py [spherical|perspective] [launches]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
wl = hmrm.synth.WORKLOADS["C3"]
rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
scene = hmrm.Scene(rgb, cmap, wl.scene_params())
c = wl.camera()
c.vang = hmrm.degrees_to_rads(40.0)
if len(sys.argv) > 1 and sys.argv[1] == "perspective":
    c.projection = 1
    c.hfov = hmrm.degrees_to_rads(90)
print("kernel ms", scene.bench_kernel_ms(c, int(sys.argv[2]) if len(sys.argv) > 2 else 10))
scene.close()
