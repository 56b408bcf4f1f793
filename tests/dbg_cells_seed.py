import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))  # (test infrastructure: uses the oracle)
import numpy as np
hm = importlib.import_module("heightmap-ray-marcher_amd")
from oracle import oracle_py as oracle
os.environ["HMRM_STEP_CAP"] = "2000000"
seed = int(sys.argv[1])
rng = np.random.RandomState(seed)
mw, mh = int(rng.choice([64, 200, 513, 1024])), int(rng.choice([64, 200, 513, 1024]))
base = rng.randint(0, 40, size=(mh, mw, 1)).astype(np.uint8)
rgb = np.repeat(base, 3, axis=2)
for _ in range(int(rng.randint(1, 6))):
    tx, ty = int(rng.randint(0, mw)), int(rng.randint(0, mh))
    rgb[max(ty - 2, 0):ty + 3, max(tx - 2, 0):tx + 3] = int(rng.randint(120, 256))
cmap = rng.randint(0, 256, size=(mh, mw, 4)).astype(np.uint8)
cmap[:, :, 3] = 255
gw = float(rng.choice([0.05, 0.01, 0.2, 0.1, 1e-3, 0.3, 0.07, 3.0, 1.7, 0.05, 0.01]))
hi = float(rng.choice([4.0, 20.0, 90.0])) * gw
params = hm.SceneParams.make(0.0, hi, grid_width=gw)
frac = float(rng.choice([0.25, 0.5, 0.125, 1.0, 0.2, 0.37]))
sd = frac * gw
ex, ey = mw * gw, mh * gw
kind = int(rng.randint(0, 4))
W, H = int(rng.randint(8, 40)), int(rng.randint(4, 16))
vang = float(hm.degrees_to_rads(90.0 + rng.choice([0.0, 0.5, 3.0, 20.0, 45.0])))
assert kind > 1
side = int(rng.randint(0, 4))
d = float(2.0 ** int(rng.randint(-3, 4)))
pos = [[-d, -ey / 2, hi * 1.5], [ex + d, -ey / 2, hi * 1.5], [ex / 2, d, hi * 1.5], [ex / 2, -ey - d, hi * 1.5]][side]
pos = [float(np.round(v * 8) / 8) for v in pos]
hang = [0.0, np.pi, -np.pi / 2, np.pi / 2][side] + float(rng.choice([0.0, 0.0, 0.01, -0.3]))
cam = hm.Camera.make(width=W, height=H, projection=int(rng.choice([1, 2])), hfov=float(hm.degrees_to_rads(rng.uniform(2.0, 70.0))),
                     hang=float(hang), vang=float(hm.degrees_to_rads(rng.uniform(95.0, 135.0))), pos=tuple(pos), step_dist=sd,
                     bg=(1, 2, 3), sampling=int([0, 0, 0, 2][int(rng.randint(0, 4))]))
print("scene", (mw, mh), "gw", gw, "sd", sd, "proj", cam.projection, "res", (W, H), "pos", pos, "hang", hang, "vang", cam.vang, "hfov", cam.hfov, "sampling", cam.sampling)
heights = oracle.update_heightmap(rgb, params)
cfg = oracle.make_cfg(cam, params, mw, mh, step_cap=2000000)
ofb, total, capped, osteps, oentry = oracle.render(cfg, heights, cmap, per_pixel=True)
for variant in ("leap", "leap", "group", "simple", "leap"):
    os.environ["HMRM_KERNEL"] = variant
    scene = hm.Scene(rgb, cmap, params)
    for rep in range(3):
        fb, st, steps, entry = scene.render_stats(cam, per_pixel=True, allow_capped=True)
        plain = scene.render(cam)
        bad = np.argwhere((fb != ofb).any(axis=2))
        badp = np.argwhere((plain != ofb).any(axis=2))
        print(variant, rep, "stats-frame diffs", len(bad), "plain-frame diffs", len(badp), "steps diffs", int((steps.astype(np.int64) != osteps).sum()),
              "entry diffs", int((entry.view(np.uint64) != oentry.view(np.uint64)).sum()))
        for (r, c) in bad[:4]:
            print("   px", (c, r), "gpu", fb[r, c], "oracle", ofb[r, c], "steps", steps[r, c], osteps[r, c], "entry", entry[r, c], "neighbours gpu", fb[r, max(c-1,0)], fb[r, min(c+1, W-1)])
    scene.close()
