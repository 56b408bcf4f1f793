#!/usr/bin/env python3
"""The traversal on content built to defeat it: for every map kind of synth.CONTENT_KINDS and the C3
and C5 cameras, kernel time of the library as shipped (auto: the production kernel unless the scene's one-time probe
measured the plain groups faster), of the production kernel alone (leap: HMRM_TRY_GROUP=0), of the plain speculative
groups (group) and of the literal loop (simple), with the instrumented kernel's traversal counters -- and the worst
auto / group and leap / group ratios.

  python tools/content_bench.py [C3 C5] > profiles/r04_content.txt

Also checks every variant's frame against the instrumented frame (bit-exact pixels whatever the kernel); parity
against the oracle on these maps is tests/test_parity_gpu.py::test_hostile_content_full_frames_match_oracle."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
synth = hmrm.synth
bases = sys.argv[1:] or ["C3", "C5"]
kinds = os.environ.get("KINDS", ",".join(synth.CONTENT_KINDS)).split(",")
variants = os.environ.get("VARIANTS", "auto,leap,group,simple").split(",")
ENV = {"auto": {}, "leap": {"HMRM_TRY_GROUP": "0"}, "group": {"HMRM_KERNEL": "group"}, "simple": {"HMRM_KERNEL": "simple"}}


def setenv(v):
    for k in ("HMRM_KERNEL", "HMRM_TRY_GROUP"):
        os.environ.pop(k, None)
    os.environ.update(ENV[v])


worst_auto, worst_leap = (0.0, ""), (0.0, "")
print(f"{'workload':14s} {'auto ms':>9s} {'chose':>6s} {'leap ms':>9s} {'group ms':>9s} {'simple ms':>10s} {'auto/group':>10s} {'leap/group':>10s} | "
      f"{'ray-steps':>13s} {'attempts':>11s} {'jumps':>10s} {'success':>7s} {'groups':>10s} {'leaped %':>8s} {'hits':>9s}")
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
        med, chose = {}, "-"
        for v in variants:
            # (a scene of its own per variant: the probe's verdict is kept per scene)
            setenv(v)
            scene = hmrm.Scene(*maps[wl.map_size], wl.scene_params())
            for _ in range(14 if v != "simple" else 1):  # (launch-order calibration and, for auto, the kernel probe)
                scene.bench_kernel_ms(cam, 1)
            if not np.array_equal(scene.render(cam), fb):
                raise SystemExit(f"{wl.name}: variant {v} renders a different frame")
            n = 2 if v == "simple" else 10
            med[v] = float(np.median([scene.bench_kernel_ms(cam, n) for _ in range(5)][1:]))
            if v == "auto":
                chose = ("leap", "group", "simple", "rec")[scene.kernel_choice()]
            scene.close()
        nan = float("nan")
        ra, rl = med.get("auto", nan) / med.get("group", nan), med.get("leap", nan) / med.get("group", nan)
        if ra > worst_auto[0]:
            worst_auto = (ra, wl.name)
        if rl > worst_leap[0]:
            worst_leap = (rl, wl.name)
        print(f"{wl.name:14s} {med.get('auto', nan):9.4f} {chose:>6s} {med.get('leap', nan):9.4f} {med.get('group', nan):9.4f} {med.get('simple', nan):10.3f} "
              f"{ra:10.3f} {rl:10.3f} | {st.steps:13d} {st.leap_attempts:11d} {st.leaps:10d} {st.leaps / max(st.leap_attempts, 1):7.3f} {st.groups:10d} "
              f"{100.0 * st.leaped_steps / max(st.steps, 1):8.2f} {st.hits:9d}", flush=True)
setenv("auto")
print(f"worst auto / group (the library as shipped): {worst_auto[0]:.3f} ({worst_auto[1]})")
print(f"worst leap / group (production kernel forced): {worst_leap[0]:.3f} ({worst_leap[1]})")
