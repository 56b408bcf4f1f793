#!/usr/bin/env python3
"""The record kernel (HMRM_KERNEL=rec: leaps over window records, frame.hpp WindowRecord) against the plain groups and the
production kernel on every map kind of synth.CONTENT_KINDS, C3 and C5 cameras: kernel ms (median of 4 x 10 launches after
the launch-order calibration), frames compared bit for bit with the instrumented production frame.

  python tools/rec_bench.py [C3 C5] > profiles/r05_raw/rec_bench.txt"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
synth = hmrm.synth
bases = sys.argv[1:] or ["C3", "C5"]
kinds = os.environ.get("KINDS", ",".join(synth.CONTENT_KINDS)).split(",")
ENV = {"leap": {"HMRM_TRY_GROUP": "0"}, "group": {"HMRM_KERNEL": "group"}, "rec": {"HMRM_KERNEL": "rec"}}


def setenv(v):
    for k in ("HMRM_KERNEL", "HMRM_TRY_GROUP"):
        os.environ.pop(k, None)
    os.environ.update(ENV[v])


print(f"{'workload':14s} {'leap ms':>9s} {'group ms':>9s} {'rec ms':>9s} {'rec/group':>9s} {'rec/leap':>9s} | scene set-up s (maps uploaded, heights, pyramid, records)")
for kind in kinds:
    maps = {}
    for base in bases:
        wl = synth.content_workload(base, kind) if kind != "smooth" else synth.WORKLOADS[base]
        if wl.map_size not in maps:
            maps[wl.map_size] = wl.maps()
        cam = wl.camera()
        setenv("leap")
        scene = hmrm.Scene(*maps[wl.map_size], wl.scene_params())
        fb, st, _, _ = scene.render_stats(cam)
        scene.close()
        med = {}
        setup = 0.0
        for v in ("leap", "group", "rec"):
            setenv(v)
            t0 = time.perf_counter()
            scene = hmrm.Scene(*maps[wl.map_size], wl.scene_params())
            setup = time.perf_counter() - t0
            for _ in range(14):
                scene.bench_kernel_ms(cam, 1)
            if not np.array_equal(scene.render(cam), fb):
                raise SystemExit(f"{wl.name}: variant {v} renders a different frame")
            med[v] = float(np.median([scene.bench_kernel_ms(cam, 10) for _ in range(5)][1:]))
            scene.close()
        print(f"{wl.name:14s} {med['leap']:9.4f} {med['group']:9.4f} {med['rec']:9.4f} {med['rec'] / med['group']:9.3f} {med['rec'] / med['leap']:9.3f} | {setup:.3f}", flush=True)
