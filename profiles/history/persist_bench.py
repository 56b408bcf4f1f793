#!/usr/bin/env python3
"""Persistent-tile experiment: the production kernel dispatched one workgroup per tile ("classic")
against resident waves pulling wave tiles from per-XCD queue heads (HMRM_PERSIST=1), interleaved rounds, kernel ms by
HIP events (median).  Every configuration's frame is compared with the instrumented kernel's first.

  python tools/persist_bench.py C3 C5 C2 C4      CONFIGS="chunk:waves:single,..." (0 / -1 = default)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
configs = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("CONFIGS", "8:0:-1,4:0:-1,16:0:-1,8:0:0").split(",")]
KEYS = ("HMRM_PERSIST", "HMRM_PERSIST_CHUNK", "HMRM_PERSIST_WAVES", "HMRM_PERSIST_SINGLE")


def setenv(cfg):
    for k in KEYS:
        os.environ.pop(k, None)
    if cfg is not None:
        os.environ["HMRM_PERSIST"] = "1"
        os.environ["HMRM_PERSIST_CHUNK"] = str(cfg[0])
        if cfg[1] > 0:
            os.environ["HMRM_PERSIST_WAVES"] = str(cfg[1])
        if cfg[2] >= 0:
            os.environ["HMRM_PERSIST_SINGLE"] = str(cfg[2])


for name in sys.argv[1:] or ["C3"]:
    wl = hmrm.synth.WORKLOADS[name] if "/" not in name else hmrm.synth.content_workload(*name.split("/"))
    scene = hmrm.Scene(*wl.maps(), wl.scene_params())
    cam = wl.camera()
    setenv(None)
    fb, st, _, _ = scene.render_stats(cam)
    for _ in range(12):
        scene.bench_kernel_ms(cam, 1)
    classic_rot = []
    os.environ["HMRM_TILE_ORDER"] = "1"  # the plain rotation, which is what the persistent kernel follows
    for _ in range(3):
        classic_rot.append(scene.bench_kernel_ms(cam, 10))
    del os.environ["HMRM_TILE_ORDER"]
    for cfg in configs:
        setenv(cfg)
        if not np.array_equal(scene.render(cam), fb):
            raise SystemExit(f"{name}: persistent kernel {cfg} renders a different frame")
    times = {None: [], **{c: [] for c in configs}}
    for rnd in range(6):
        for cfg in [None] + configs:
            setenv(cfg)
            if cfg is None and rnd == 0:
                for _ in range(12):
                    scene.bench_kernel_ms(cam, 1)  # (calibration again: the knobs were reloaded)
            times[cfg].append(scene.bench_kernel_ms(cam, 10))
    base = float(np.median(times[None][1:]))
    print(f"{name}: classic (calibrated order) {base:.4f} ms; classic, plain rotation {np.median(classic_rot[1:]):.4f} ms")
    for cfg in configs:
        t = float(np.median(times[cfg][1:]))
        print(f"{name}: persistent chunk {cfg[0]:2d} waves {cfg[1] or 'all':>5} single rows {cfg[2] if cfg[2] >= 0 else 'marching':>8}: "
              f"{t:.4f} ms  ({t / base:.3f} x classic)", flush=True)
    setenv(None)
    scene.close()
