"""Kernel time of the bilinear quality mode next to the reference's nearest-cell mode (tools only)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hm = importlib.import_module("heightmap-ray-marcher_amd")

for name in sys.argv[1:] or ["C3", "C2", "C5"]:
    wl = hm.synth.WORKLOADS[name]
    rgb, cmap = hm.synth.synth_maps(wl.map_size)
    scene = hm.Scene(rgb, cmap, wl.scene_params())
    for sampling in (hm.NEAREST, hm.BILINEAR):
        cam = wl.camera()
        cam.sampling = sampling
        fb, st, *_ = scene.render_stats(cam)
        scene.bench_kernel_ms(cam, 5)
        ms = scene.bench_kernel_ms(cam, 50)
        print(f"{name} sampling={'bilinear' if sampling else 'nearest'}: {ms:.3f} ms/frame, steps {st.steps:.3e}, "
              f"leaped {st.leaped_steps / max(st.steps, 1):.3f}, hits {st.hits}", flush=True)
    scene.close()
