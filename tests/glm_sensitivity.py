#!/usr/bin/env python3
"""How much rides on the unpinned glm formulas?  (CPU only, uses the oracle: test infrastructure.)

The reference's vector arithmetic is glm's ("tested with 0.9.9.8", README.md:109), which is not under
/root/reference; oracle/hmrm_oracle.c restates `normalize(v) = v * (1 / sqrt(dot(v, v)))` and
`dot = (x + y) + z` from glm's published source and nothing the reference holds pins them
(call sites: src/Perspective.cpp:13-14,27; src/Orthographic.cpp:11).  This script renders BASELINE
configs C1, C2 (full frames) and C3/C5 (every 8th row) with the assumed formulas and with each
alternative a different glm build could have used, and counts the pixels and rays that change.

usage: python tests/glm_sensitivity.py [out.txt]
"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HMRM_NO_TORCH_PRELOAD", "1")
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
from oracle import oracle_py as oracle

VARIANTS = {1: "normalize = v / sqrt(dot)", 2: "dot = x + (y + z)", 3: "both"}
lines = [__doc__.split("usage:")[0].strip(), ""]
for name, stride in (("C1", 1), ("C2", 1), ("C5", 8), ("C3", 8)):
    wl = hmrm.synth.WORKLOADS[name]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
    params, cam = wl.scene_params(), wl.camera()
    heights = oracle.update_heightmap(rgb, params)
    cfg = oracle.make_cfg(cam, params, wl.map_size, wl.map_size)
    oracle.set_glm_variant(0)
    fb0, total0, _, steps0, entry0 = oracle.render(cfg, heights, cmap, per_pixel=True, row_stride=stride)
    rows = slice(0, cam.height, stride)
    npx = fb0[rows].shape[0] * fb0[rows].shape[1]
    proj = ("perspective", "spherical", "orthographic")[wl.projection - 1]
    lines.append(f"{name} ({proj}, {cam.width}x{cam.height} over {wl.map_size}^2, every {stride}. row: {npx} rays, {total0} ray-steps)")
    for v, what in VARIANTS.items():
        oracle.set_glm_variant(v)
        fb, total, _, steps, entry = oracle.render(cfg, heights, cmap, per_pixel=True, row_stride=stride)
        px = int((fb[rows] != fb0[rows]).any(axis=2).sum())
        st = int((steps[rows] != steps0[rows]).sum())
        en = int((entry[rows].view(np.uint64) != entry0[rows].view(np.uint64)).sum())
        lines.append(f"   {what:28s}: {px:6d} pixels differ ({px / npx:.2e}), {st:7d} rays take another step count, "
                     f"{en:8d} entry distances differ in the last bits, ray-steps {total - total0:+d}")
    oracle.set_glm_variant(0)
    lines.append("")
lines.append("Spherical (C3) never calls glm's normalize/dot/cross (src/Spherical.cpp:17-31), orthographic calls only cross\n"
             "(src/Orthographic.cpp:11, not switched here): the assumption matters for perspective frames only.")
text = "\n".join(lines) + "\n"
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text)
