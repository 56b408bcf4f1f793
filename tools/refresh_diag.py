#!/usr/bin/env python3
"""How often does a wave of the production kernel run axis_refresh blocks (instrumented kernel, HMRM_DIAG_ITERS=16)?"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
for name in sys.argv[1:] or ["C3", "C5"]:
    wl = hmrm.synth.WORKLOADS[name]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
    scene = hmrm.Scene(rgb, cmap, wl.scene_params())
    os.environ["HMRM_DIAG_ITERS"] = "16"
    _, st, _, _ = scene.render_stats(wl.camera())
    os.environ.pop("HMRM_DIAG_ITERS")
    a, r, l, n = st.leap_attempts, st.leaps, st.groups, st.leaped_steps
    print(f"{name}: attempt blocks run by waves {a}; refresh blocks run {r} = {r / max(a, 1):.2f} per attempt block (of 3); "
          f"lanes needing a refresh {l} = {l / max(r, 1):.1f} per block run; lanes per attempt block {n / max(a, 1):.1f}", flush=True)
    scene.close()
