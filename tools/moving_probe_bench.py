#!/usr/bin/env python3
"""The scene's shadow probe under a MOVING camera (every frame a camera the library has never seen, so nothing is ever
calibrated): ms per frame of the first 20, the next 20 and the next 120 frames with the probe (HMRM_TRY_GROUP=1, default)
and without, on maps where rays cannot jump and on the smooth terrain -> profiles/r04_raw/moving_camera_probe.txt"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
import torch
for name in ("C5/needles", "C5/white", "C5"):
    wl = hmrm.synth.WORKLOADS[name] if "/" not in name else hmrm.synth.content_workload(*name.split("/"))
    maps = wl.maps()
    for try_group in ("1", "0"):
        os.environ["HMRM_TRY_GROUP"] = try_group
        os.environ["HMRM_ORDER_VERBOSE"] = "1"
        scene = hmrm.Scene(*maps, wl.scene_params())
        out = torch.empty((wl.height, wl.width, 4), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        cams = [wl.camera(k, 100003) for k in range(1, 161)]
        def run(cs):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for c in cs:
                scene.render_rows_device(c, out.data_ptr(), wl.width * 4, 0, wl.height, stream=st)
            torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / len(cs)
        a = run(cams[:20]); b = run(cams[20:40]); c = run(cams[40:160])
        print(f"{name} HMRM_TRY_GROUP={try_group}: moving camera ms per frame: first 20 {a:.4f}, next 20 {b:.4f}, next 120 {c:.4f}; choice {scene.kernel_choice()}", flush=True)
        scene.close()
