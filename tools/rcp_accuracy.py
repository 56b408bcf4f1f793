#!/usr/bin/env python3
"""Measure v_rcp_f64's relative error on the GPU at hand (hmrm_debug_rcp_error) and print a report.
slab_classify's margins (csrc/device_common.hpp kRcpRelErr) are derived from the largest figure printed here;
tests/test_parity_gpu.py::test_rcp_f64_accuracy_bound re-measures a slice in the driver's run.
  python tools/rcp_accuracy.py [--quick] > gpurun_out/rcp_accuracy.txt"""
import argparse
import importlib
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")


def log2s(e):
    return "exact" if e == 0 else f"2^{math.log2(e):.2f}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    hmrm.set_device(0)
    full = 1 << (28 if a.quick else 32)
    worst = 0.0
    total = 0
    print("v_rcp_f64 on this device: relative error |rcp(x) * x - 1| (exact fma), worst over each sample")
    runs = []
    for e in ((0,) if a.quick else (0, 1, -1, 52, -300, 300)):
        for low in (0, 1, 2):
            runs.append((f"mode 0: leading 32 mantissa bits exhaustive, exponent {e:+d}, trailing bits "
                         f"{('zero', 'ones', 'hashed')[low]}", (0, full, low, e, e)))
    for lo, hi, seeds in ((-40, 0, 4), (-1, 14, 4), (-500, 500, 4), (-1000, 1000, 2)):
        for s in range(1 if a.quick else seeds):
            runs.append((f"mode 1: hashed mantissa/sign, exponent in [{lo}, {hi}], seed {s}", (1, full, 4 * s + 7, lo, hi)))
    for lo, hi in ((-10, 14), (-60, 60)):
        for s in range(1 if a.quick else 3):
            runs.append((f"mode 2: n * rcp(d) vs n / d, n exponent in [{lo}, {hi}], d in 2^-40..1, seed {s}", (2, full, 4 * s + 9, lo, hi)))
    for name, args in runs:
        m, hist = hmrm.rcp_error(*args)
        total += args[1]
        worst = max(worst, m)
        nz = [(k, int(c)) for k, c in enumerate(hist) if c]
        top = ", ".join(f"[2^-{k},2^-{k - 1}): {c}" for k, c in nz[:4])
        print(f"{name}\n    n = {args[1]:>11d}  max = {m:.6e} = {log2s(m)} = {m * 2.0**52:.3f} x 2^-52   largest bins: {top}")
        sys.stdout.flush()
    print(f"TOTAL {total} samples; WORST relative error {worst:.6e} = {log2s(worst)} = {worst * 2.0**52:.3f} x 2^-52")


if __name__ == "__main__":
    main()
