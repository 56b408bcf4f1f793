"""Like deep_fuzz.py, over ONE big synthetic map (all pyramid levels, long jumps, binade crossings at large
coordinates): random cameras, projections, step sizes and grid widths against the CPU oracle.
Test infrastructure, not collected by pytest: python tests/deep_fuzz_big.py <first seed> <cameras> [seconds] [map size]."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
hm = importlib.import_module("heightmap-ray-marcher_amd")
from oracle import oracle_py as oracle

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
budget_s = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
S = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
CAP = 2000000
os.environ["HMRM_STEP_CAP"] = str(CAP)
rgb, cmap = hm.synth.synth_maps(S)
t0 = time.time()
bad = done = leaped = 0
scene = None
cur = None
for k in range(count):
    if time.time() - t0 > budget_s:
        break
    seed = seed0 + k
    rng = np.random.RandomState(seed)
    gw = float(rng.choice([1.0, 1.0, 0.5, 2.0, 0.05, 0.3, 7.0]))
    lo = float(rng.choice([0.0, 0.0, -20.0, 5.0]))
    hi = lo + float(rng.choice([S / 16.0, S / 64.0, S / 4.0])) * gw
    key = (gw, lo, hi)
    if key != cur:  # a new scene only when the parameters change
        if scene is not None:
            scene.close()
        params = hm.SceneParams.make(lo, hi, grid_width=gw)
        scene = hm.Scene(rgb, cmap, params)
        heights = oracle.update_heightmap(rgb, params)
        cur = key
    ext = S * gw
    proj = int(rng.choice([1, 2, 3]))
    ang = rng.uniform(0, 2 * np.pi)
    dist = rng.uniform(0.0, 1.3) * ext
    pos = [ext / 2 + dist * np.cos(ang), -ext / 2 + dist * np.sin(ang), hi + rng.uniform(0.01, 4.0) * (hi - lo)]
    hang = float(np.arctan2(-ext / 2 - pos[1], ext / 2 - pos[0]) + rng.uniform(-0.7, 0.7))
    vang = float(hm.degrees_to_rads(rng.uniform(80, 150)))
    sd = float(rng.choice([0.25, 0.5, 1.0, 0.1, 0.37, 2.5]) * gw)
    cam = hm.Camera.make(width=int(rng.randint(40, 200)), height=int(rng.randint(30, 120)), projection=proj,
                         hfov=float(hm.degrees_to_rads(rng.uniform(20, 178))), hang=hang, vang=vang, pos=tuple(pos),
                         ortho_width=float(rng.uniform(0.5, 12.0) * gw), step_dist=sd,
                         bg=(3, 2, 1), sampling=int([0, 0, 0, 0, 1, 2][int(rng.randint(0, 6))]))
    cfg = oracle.make_cfg(cam, params, S, S, step_cap=CAP)
    ofb, total, capped, osteps, oentry = oracle.render(cfg, heights, cmap, per_pixel=True)
    fb, st, steps, entry = scene.render_stats(cam, per_pixel=True, allow_capped=True)
    live = osteps >= 0
    ok = (np.array_equal(entry.view(np.uint64), oentry.view(np.uint64)) and st.capped == capped and np.array_equal(fb[live], ofb[live])
          and np.array_equal(steps.astype(np.int64)[live], osteps[live]))
    if capped == 0:
        ok = ok and st.steps == total and np.array_equal(scene.render(cam), ofb)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "gw", gw, "heights", (lo, hi), "proj", proj, "sd", sd, "res", (cam.width, cam.height),
              "sampling", cam.sampling, "capped", capped, st.capped, "frame diff", int((fb != ofb).any(axis=2).sum()),
              "steps diff", int((steps.astype(np.int64) != osteps).sum()), flush=True)
    leaped += st.leaped_steps
    done += 1
    if done % 200 == 0:
        print("... %d cameras, %d mismatches, %.3e steps leaped, %.0f s" % (done, bad, leaped, time.time() - t0), flush=True)
print("map %dx%d: cameras %d, mismatches %d, leaped steps %.3e, %.0f s" % (S, S, done, bad, leaped, time.time() - t0))
sys.exit(1 if bad else 0)
