#!/usr/bin/env python3
"""How long does a launch take when (nearly) every wave only shades a miss?  The BASELINE camera turned to the sky: the
same grid of waves, no marching -- the floor that wave dispatch, the table loads and the frame store put under a launch.
usage: sky_frame_time.py [workload]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
wl = hmrm.synth.WORKLOADS[name]
rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
scene = hmrm.Scene(rgb, cmap, wl.scene_params())
os.environ["HMRM_TILE_ORDER"] = "1"
for label, vang in (("as specified", None), ("turned 60 degrees up", -60.0), ("turned to the zenith", "zenith")):
    cam = wl.camera()
    if vang == "zenith":
        cam.vang = hmrm.degrees_to_rads(1.0)
    elif vang is not None:
        cam.vang = cam.vang + hmrm.degrees_to_rads(vang)
    _, st, *_ = scene.render_stats(cam)
    t = [scene.bench_kernel_ms(cam, 10) for _ in range(5)]
    print(f"{name} {label:22s}: kernel {np.median(t):.4f} ms   rays {st.rays}  hits {st.hits}  steps {st.steps}", flush=True)
scene.close()
