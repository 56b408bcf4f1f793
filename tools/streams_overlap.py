#!/usr/bin/env python3
"""Frames in flight on several streams of ONE GPU: do the tails of consecutive frames overlap?
usage: streams_overlap.py [workload] -- frames per second of the static pose / of the 64-frame orbit with 1..4 streams."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
import torch
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
wl = hmrm.synth.WORKLOADS[name]
rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
scene = hmrm.Scene(rgb, cmap, wl.scene_params())
cam = wl.camera()
W, H = cam.width, cam.height
N = 240
for orbit in (False, True):
    cams = [wl.camera(k, 24) for k in range(24)] if orbit else [cam]
    for S in [int(x) for x in os.environ.get("STREAMS", "1,2,3,4").split(",")]:
        streams = [torch.cuda.Stream() for _ in range(S)]
        outs = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(S)]
        def run(n):
            for i in range(n):
                s = i % S
                scene.render_rows_device(cams[i % len(cams)], outs[s].data_ptr(), W * 4, 0, H, stream=streams[s].cuda_stream)
        run(2 * 24 * S if orbit else 20)  # (every camera once on every stream: per-stream frame caches)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            run(N)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / N)
        print(f"{name} {'24-frame orbit' if orbit else 'static pose  '} {S} stream(s): {best * 1e3:.4f} ms per frame", flush=True)
scene.close()
