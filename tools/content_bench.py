#!/usr/bin/env python3
"""The traversal on content built to defeat it: for every map kind of synth.CONTENT_KINDS and the C3
and C5 cameras, kernel time of the production kernel (leap), the plain speculative groups (group) and the literal
loop (simple), with the instrumented kernel's traversal counters -- and the worst leap / group ratio.

  python tools/content_bench.py [C3 C5] > profiles/r04_content.txt

Also checks every frame against the instrumented frame (bit-exact pixels whatever the kernel variant); parity
against the oracle on these maps is tests/test_parity_gpu.py::test_hostile_content_full_frames_match_oracle."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
synth = hmrm.synth
bases = sys.argv[1:] or ["C3", "C5"]
kinds = os.environ.get("KINDS", ",".join(synth.CONTENT_KINDS)).split(",")
variants = os.environ.get("VARIANTS", "leap,group,simple").split(",")
worst = (0.0, "")
print(f"{'workload':14s} {'leap ms':>9s} {'group ms':>9s} {'simple ms':>10s} {'leap/group':>10s} | {'ray-steps':>13s} {'attempts':>11s} "
      f"{'jumps':>10s} {'success':>7s} {'groups':>10s} {'leaped %':>8s} {'hits':>9s}")
for kind in kinds:
    maps = {}
    for base in bases:
        wl = synth.content_workload(base, kind) if kind != "smooth" else synth.WORKLOADS[base]
        if wl.map_size not in maps:
            maps[wl.map_size] = wl.maps()
        scene = hmrm.Scene(*maps[wl.map_size], wl.scene_params())
        cam = wl.camera()
        os.environ["HMRM_KERNEL"] = "leap"
        fb, st, _, _ = scene.render_stats(cam)
        times = {v: [] for v in variants}
        for v in variants:
            os.environ["HMRM_KERNEL"] = v
            for _ in range(12 if v != "simple" else 1):  # (launch-order calibration of this variant)
                scene.bench_kernel_ms(cam, 1)
            if not np.array_equal(scene.render(cam), fb):
                raise SystemExit(f"{wl.name}: HMRM_KERNEL={v} renders a different frame")
        for rnd in range(5):
            for v in variants:
                os.environ["HMRM_KERNEL"] = v
                times[v].append(scene.bench_kernel_ms(cam, 2 if v == "simple" else 10))
        med = {v: float(np.median(times[v][1:])) for v in variants}
        ratio = med["leap"] / med["group"] if "group" in med else float("nan")
        if ratio > worst[0]:
            worst = (ratio, wl.name)
        print(f"{wl.name:14s} {med['leap']:9.4f} {med.get('group', float('nan')):9.4f} {med.get('simple', float('nan')):10.3f} {ratio:10.3f} | "
              f"{st.steps:13d} {st.leap_attempts:11d} {st.leaps:10d} {st.leaps / max(st.leap_attempts, 1):7.3f} {st.groups:10d} "
              f"{100.0 * st.leaped_steps / max(st.steps, 1):8.2f} {st.hits:9d}", flush=True)
        os.environ["HMRM_KERNEL"] = "leap"
        scene.close()
print(f"worst leap / group: {worst[0]:.3f} ({worst[1]})")
