"""Row-major launch order (HMRM_TILE_ORDER=0) against the cost-rotated order, kernel time."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hm = importlib.import_module("heightmap-ray-marcher_amd")
import numpy as np
for name in sys.argv[1:] or ["C3", "C2", "C5", "C4"]:
    wl = hm.synth.WORKLOADS[name]
    rgb, cmap = hm.synth.synth_maps(wl.map_size)
    scene = hm.Scene(rgb, cmap, wl.scene_params())
    cams = [("as configured", wl.camera())]
    if name == "C3":
        for vang in (20.0, 70.0, 90.0, 100.0, 140.0):
            c = wl.camera(); c.vang = hm.degrees_to_rads(vang); cams.append(("vang %.0f" % vang, c))
    for label, cam in cams:
        res = {}
        for order in ("0", "1"):
            os.environ["HMRM_TILE_ORDER"] = order
            cam.bg_b = int(order)  # a different frame record per arm
            fb = scene.render(cam)
            scene.bench_kernel_ms(cam, 5)
            res[order] = (min(scene.bench_kernel_ms(cam, 50) for _ in range(3)), fb[:, :, :2].copy())
        print(name, label, "row-major %.4f  rotated %.4f  same pixels %s" % (res["0"][0], res["1"][0], np.array_equal(res["0"][1], res["1"][1])), flush=True)
    scene.close()
