#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) frame time through hmrm_render vs kernel-only time."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
for name in sys.argv[1:] or ["C3"]:
    wl = hmrm.synth.WORKLOADS[name]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
    t0 = time.perf_counter(); scene = hmrm.Scene(rgb, cmap, wl.scene_params()); t_scene = time.perf_counter() - t0
    cam = wl.camera()
    scene.render(cam)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); fb = scene.render(cam); ts.append(time.perf_counter() - t0)
    _, st, _, _ = scene.render_stats(cam)
    k = scene.bench_kernel_ms(cam, 20)
    e2e = float(np.median(ts)) * 1e3
    # asynchronous ring (hmrm_render_begin/_wait): `depth` frames in flight, kernel k+1 beside the copy of frame k
    shape = (cam.height, cam.width)
    warm = [scene.render_begin(cam) for _ in range(3)]  # (the ring's pinned frames are allocated on first use)
    for t in warm:
        scene.render_wait(t, shape, copy=False)
        scene.render_release(t)
    for depth in (1, 2, 3):
        n = 60
        t0 = time.perf_counter()
        q = []
        for i in range(n):
            q.append(scene.render_begin(cam))
            if len(q) >= depth:
                t = q.pop(0)
                v = scene.render_wait(t, shape, copy=False)
                scene.render_release(t)
        while q:
            t = q.pop(0)
            v = scene.render_wait(t, shape, copy=False)
            scene.render_release(t)
        dt = (time.perf_counter() - t0) * 1e3 / n
        print(f"{name}: async ring, {depth} frame(s) in flight: {dt:.3f} ms per frame sustained "
              f"({fb.nbytes / dt / 1e6:.1f} GB/s of frame data into pinned host memory)", flush=True)
    assert np.array_equal(v, fb)
    print(f"{name}: scene upload+prepare {t_scene*1e3:.1f} ms; hmrm_render end-to-end (kernel + D2H of {fb.nbytes/1e6:.1f} MB into pageable memory) "
          f"{e2e:.2f} ms = {st.steps/e2e/1e6:.1f} Gsteps/s PCIe-inclusive; kernel only {k:.3f} ms = {st.steps/k/1e6:.1f} Gsteps/s")
    scene.close()
