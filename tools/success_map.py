#!/usr/bin/env python3
"""Per-pixel leap attempts and successful jumps (instrumented kernel, HMRM_DIAG_ITERS=19 and 1): how much of the longest
rays' chains are refused attempts?  usage: success_map.py [workload]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
os.environ["HMRM_KERNEL"] = "leap"
wl = hmrm.synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
scene = hmrm.Scene(rgb, cmap, wl.scene_params())
cam = wl.camera()
def diag(mode):
    os.environ["HMRM_DIAG_ITERS"] = str(mode)
    _, st, out, _ = scene.render_stats(cam, per_pixel=True)
    os.environ.pop("HMRM_DIAG_ITERS")
    return out.astype(np.int64)
a = diag(19); b = diag(1)
att, ok, grp = a >> 16, a & 0xffff, b & 0xffff
trips = att + grp
print(f"all rays: attempts {att.sum()} successful {ok.sum()} ({ok.sum() / att.sum():.3f}) groups {grp.sum()}")
for lo, hi in ((1, 8), (8, 16), (16, 32), (32, 64), (64, 1000)):
    m = (trips >= lo) & (trips < hi)
    if m.any():
        print(f" rays with {lo:3d}..{hi:4d} trips: n={int(m.sum()):8d}  attempts/ray {att[m].mean():6.1f}  successful {ok[m].sum() / att[m].sum():.3f}  groups/ray {grp[m].mean():5.1f}")
H, W = trips.shape
t8 = trips[:H // 8 * 8, :W // 8 * 8].reshape(H // 8, 8, W // 8, 8)
wmax = t8.max(axis=(1, 3))
arg = np.argsort(wmax, axis=None)[::-1][:10]
print("longest waves: tile, longest lane's (attempts, successful, groups)")
for o in arg:
    ty, tx = np.unravel_index(o, wmax.shape)
    blk = (slice(ty * 8, ty * 8 + 8), slice(tx * 8, tx * 8 + 8))
    j = np.unravel_index(np.argmax(trips[blk]), (8, 8))
    print(f"  tile ({ty},{tx}) trips {int(wmax[ty, tx])}: attempts {int(att[blk][j])} successful {int(ok[blk][j])} groups {int(grp[blk][j])}")
scene.close()
