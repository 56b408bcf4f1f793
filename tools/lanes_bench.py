#!/usr/bin/env python3
"""Small frames through the ticketed entry points: GPU time per frame of a sequence of frames of one
pose -- back to back on one stream, on three caller-managed streams (bench.py's frames_in_flight), and through
hmrm_render_device_begin / _wait with 1, 2, 3 and 4 tickets in flight (the library's three launch lanes, no stream
in the caller's hands); then the host-memory ring (hmrm_render_begin: PCIe-bound) with 1 and 3 tickets in flight.

  python tools/lanes_bench.py C2 C3 > profiles/r04_lanes.txt"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
import torch

N = int(os.environ.get("FRAMES", "300"))
for name in sys.argv[1:] or ["C2"]:
    wl = hmrm.synth.WORKLOADS[name]
    scene = hmrm.Scene(*wl.maps(), wl.scene_params())
    cam = wl.camera()
    W, H = cam.width, cam.height
    fb = scene.render(cam)
    bufs = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    for _ in range(12):
        scene.bench_kernel_ms(cam, 1)

    def timed(fn, n=N):
        fn(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n

    def one_stream(n):
        s0 = torch.cuda.current_stream().cuda_stream
        for _ in range(n):
            scene.render_rows_device(cam, bufs[0].data_ptr(), W * 4, 0, H, stream=s0)

    def three_streams(n):
        for i in range(n):
            scene.render_rows_device(cam, bufs[i % 3].data_ptr(), W * 4, 0, H, stream=streams[i % 3].cuda_stream)

    def tickets(depth):
        def run(n):
            inflight = []
            for i in range(n):
                if len(inflight) == depth:
                    scene.render_device_wait(inflight.pop(0))
                inflight.append(scene.render_device_begin(cam, bufs[i % depth].data_ptr(), W * 4))
            for t in inflight:
                scene.render_device_wait(t)
        return run

    def ring(depth):
        def run(n):
            inflight = []
            for i in range(n):
                if len(inflight) == depth:
                    t = inflight.pop(0)
                    scene.render_wait(t, (H, W), copy=False)
                    scene.render_release(t)
                inflight.append(scene.render_begin(cam))
            for t in inflight:
                scene.render_wait(t, (H, W), copy=False)
                scene.render_release(t)
        return run
    base = timed(one_stream)
    print(f"{name}: one stream, back to back          {base:.4f} ms per frame")
    print(f"{name}: three caller-managed streams      {timed(three_streams):.4f}")
    for d in (1, 2, 3, 4):
        t = timed(tickets(d))
        print(f"{name}: device tickets, {d} in flight       {t:.4f}  ({t / base:.3f} x one stream)", flush=True)
    for b in bufs[:3]:
        assert np.array_equal(b.cpu().numpy(), fb)
    for d in (1, 3):
        t = timed(ring(d), max(30, N // 10))
        print(f"{name}: host ring (PCIe), {d} in flight     {t:.4f}  ({W * H * 4 / t / 1e6:.1f} GB/s of frame data)", flush=True)
    scene.close()
