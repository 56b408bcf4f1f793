#!/usr/bin/env python3
"""Occupancy timeline of ONE launch of the PRODUCTION kernel (tool build -DHMRM_TIMELINE: every wave stores its start /
end time and its XCD, into a device buffer): waves resident per 1/50th of the launch, split into waves that march and waves that only shade a
miss, the frame rows being worked on, per-XCD finish times.  Shows ramp, steady state and tail.
usage (GPU box): python tools/timeline.py [workload] [extra -D flags]   (rebuilds the library twice)"""
import importlib, os, struct, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
extra = sys.argv[2] if len(sys.argv) > 2 else ""
if os.environ.get("HMRM_TIMELINE_CHILD"):
    hmrm = importlib.import_module("heightmap-ray-marcher_amd")
    wl = hmrm.synth.WORKLOADS[name]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
    scene = hmrm.Scene(rgb, cmap, wl.scene_params())
    cam = wl.camera()
    _, st, steps, _ = scene.render_stats(cam, per_pixel=True)
    np.save(os.environ["HMRM_TIMELINE_FILE"] + ".steps.npy", steps)
    os.environ["HMRM_DIAG_ITERS"] = "1"
    _, _, packed, _ = scene.render_stats(cam, per_pixel=True)
    os.environ.pop("HMRM_DIAG_ITERS")
    np.save(os.environ["HMRM_TIMELINE_FILE"] + ".trips.npy", (packed >> 16).astype(np.int64) + (packed & 0xffff).astype(np.int64))
    for _ in range(12):  # (one at a time: the launch order of this camera settles)
        scene.bench_kernel_ms(cam, 1)
    print("kernel ms:", scene.bench_kernel_ms(cam, 10), flush=True)
    scene.close()
    sys.exit(0)
out = "/tmp/hmrm_timeline.bin"
subprocess.run(["bash", os.path.join(ROOT, "tools", "sweep_build.sh"), ("-DHMRM_TIMELINE " + extra).strip()], check=True)
try:
    subprocess.run([sys.executable, __file__, name], check=True, env=dict(os.environ, HMRM_TIMELINE_CHILD="1", HMRM_TIMELINE_FILE=out))
finally:
    subprocess.run(["bash", os.path.join(ROOT, "tools", "sweep_build.sh"), ""], check=True)
raw = open(out, "rb").read()
gx, gy, wpb, used, tiles_y, *segs = struct.unpack("12i", raw[:48])
seg_first, seg_delta = segs[:3], segs[3:]
rot = seg_delta[0]
rec = np.frombuffer(raw[48:], dtype=np.dtype([("t0", "<u8"), ("t1", "<u8"), ("xcc", "<u4"), ("pad", "<u4")]), count=used)
steps = np.load(out + ".steps.npy").astype(np.int64)
H, W = steps.shape
# wave -> frame tile row: blocks are (x fastest, then y); a block holds wpb waves stacked vertically (8 rows each)
blk = np.arange(used) // wpb
by = blk // gx
delta = np.full(used, seg_delta[0])        # RowMap's grid row -> tile row map (device_common.hpp pixel_of_lane)
for k in range(3):
    delta = np.where(by >= seg_first[k], seg_delta[k + 1], delta)
tile_row = (by + delta) % tiles_y
wave_row0 = tile_row * (8 * wpb) + (np.arange(used) % wpb) * 8
bx = blk % gx
tiles = steps[:H // 8 * 8, :W // 8 * 8].reshape(H // 8, 8, W // 8, 8).sum(axis=(1, 3))
ok = (wave_row0 // 8 < tiles.shape[0]) & (bx < tiles.shape[1])
wsteps = np.zeros(used, dtype=np.int64)
wsteps[ok] = tiles[(wave_row0[ok] // 8), bx[ok]]
trips = np.load(out + ".trips.npy")
ttiles = trips[:H // 8 * 8, :W // 8 * 8].reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
wtrips = np.zeros(used, dtype=np.int64)
wtrips[ok] = ttiles[(wave_row0[ok] // 8), bx[ok]]
xcc = rec["xcc"] & 0xf
start = rec["t0"].astype(np.float64)
end = rec["t1"].astype(np.float64)
base = start.min()  # (s_memrealtime: one 100 MHz clock for the whole device)
start -= base
end -= base
T = end.max()
if os.environ.get("HMRM_TIMELINE_NPZ"):
    np.savez_compressed(os.environ["HMRM_TIMELINE_NPZ"], start=start.astype(np.float32), end=end.astype(np.float32), row=wave_row0.astype(np.int32),
                        col=bx.astype(np.int32), steps=wsteps, trips=wtrips.astype(np.int32), xcc=xcc.astype(np.int8))
print(f"{name} [{extra}]: {used} waves, {int((wsteps > 0).sum())} marching; launch span {T:.0f} ticks of s_memrealtime (100 MHz: {T / 100:.1f} us); launch order pieces: first {seg_first}, delta {seg_delta}")
print("per-XCD: waves, marching waves, last finish (fraction of the span); then: when half / 90 % of its waves had started (t/T), "
      "wave-time spent (us summed over waves), median us per trip of its marching waves with 8..32 trips")
for x in np.unique(xcc):
    m = xcc == x
    mid = m & (wtrips >= 8) & (wtrips < 32)
    print(f"  xcc {int(x)}: {int(m.sum()):6d} {int((m & (wsteps > 0)).sum()):6d}  {end[m].max() / T:.3f}   started 50 % {np.percentile(start[m], 50) / T:.3f} "
          f"90 % {np.percentile(start[m], 90) / T:.3f}   wave-time {(end[m] - start[m]).sum() / 100:9.0f} us   "
          f"us/trip {np.median((end[mid] - start[mid]) / 100 / wtrips[mid]) if mid.any() else 0:.3f}")
nb = 50
print("bin   t/T  resident(all)  resident(marching)  median-frame-row(marching)  waves-started")
for b in range(nb):
    a, e = T * b / nb, T * (b + 1) / nb
    ov = np.clip(np.minimum(end, e) - np.maximum(start, a), 0, None) / (e - a)
    mar = (start < e) & (end > a) & (wsteps > 0)
    print(f"{b:3d} {a / T:5.2f} {ov.sum():10.0f} {ov[wsteps > 0].sum():12.0f} {np.median(wave_row0[mar]) if mar.any() else -1:14.0f} "
          f"{int(((start >= a) & (start < e)).sum()):14d}")
dur = end - start
o = [i for i in np.argsort(end)[::-1][:400] if wsteps[i] > 0][:24]
print("last marching waves to finish: frame row, tile column, start t/T, end t/T, duration us, longest lane's trips (attempts + groups), us per trip")
for i in o:
    print(f"  row {int(wave_row0[i]):5d} col {int(bx[i]):4d} start {start[i] / T:.3f} end {end[i] / T:.3f} dur {dur[i] / 100:7.1f} us  trips {int(wtrips[i]):4d}  "
          f"{dur[i] / 100 / max(int(wtrips[i]), 1):.2f} us/trip  xcc {int(xcc[i])}")
m = wtrips >= 32
print(f"waves whose longest lane has >= 32 trips: {int(m.sum())}; us per trip: median {np.median(dur[m] / 100 / wtrips[m]):.2f}, "
      f"p10 {np.percentile(dur[m] / 100 / wtrips[m], 10):.2f}, p90 {np.percentile(dur[m] / 100 / wtrips[m], 90):.2f}")
for lo, hi in ((1, 8), (8, 16), (16, 32), (32, 64), (64, 1000)):
    m = (wtrips >= lo) & (wtrips < hi)
    if m.any():
        print(f"  waves with {lo:3d}..{hi:4d} trips: n={int(m.sum()):6d}  median duration {np.median(dur[m]) / 100:6.1f} us  median us/trip {np.median(dur[m] / 100 / wtrips[m]):.2f}")
