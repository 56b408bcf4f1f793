#!/usr/bin/env python3
"""Frames in flight on several streams of ONE GPU: do the tails of consecutive frames overlap?
usage: streams_overlap.py [workload] -- ms per frame of the static pose / of a 24-frame orbit with 1..4 streams (cameras
pre-rendered: the per-stream cache holds their host set-up), then UNCACHED: an orbit of cameras the library has never
seen, the host set-up (libm, spherical tables, upload) inside the timed loop."""
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
# uncached: every frame a new camera (positions of a 1000003-frame orbit, never repeated within this process)
fresh_k = [1]
for S in [int(x) for x in os.environ.get("STREAMS", "1,2,3,4").split(",")]:
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(S)]
    best = 1e9
    for _ in range(3):
        cams = [wl.camera(fresh_k[0] + i, 1000003) for i in range(N)]
        fresh_k[0] += N
        for i in range(20):
            scene.render_rows_device(cam, outs[i % S].data_ptr(), W * 4, 0, H, stream=streams[i % S].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            scene.render_rows_device(cams[i], outs[i % S].data_ptr(), W * 4, 0, H, stream=streams[i % S].cuda_stream)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / N)
    print(f"{name} uncached orbit  {S} stream(s): {best * 1e3:.4f} ms per frame", flush=True)
scene.close()
