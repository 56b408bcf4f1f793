"""Long seeded fuzz of the HIP path against the CPU oracle (frames, per-ray step counts, caps).
Test infrastructure, not collected by pytest: python tests/deep_fuzz.py <first seed> <scenes> [seconds].  Prints one line per
mismatch with everything needed to reproduce it, and a summary."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
hm = importlib.import_module("heightmap-ray-marcher_amd")
from oracle import oracle_py as oracle
import scenes

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
budget_s = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
os.environ["HMRM_STEP_CAP"] = "300000"
t0 = time.time()
bad = done = leaped = capped_scenes = 0
for k in range(count):
    if time.time() - t0 > budget_s:
        break
    seed = seed0 + k
    rng = np.random.RandomState(seed)
    mw = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 100, 129, 256, 300]))
    mh = int(rng.choice([1, 2, 5, 16, 31, 64, 128, 257]))
    rgb, cmap = scenes.small_maps(mw, mh, seed, color_heights=bool(k % 3 == 0))
    if k % 7 == 0:
        cmap[rng.randint(0, mh), rng.randint(0, mw), 3] = 0
    gw = float(rng.choice([1.0, 0.5, 0.25, 2.0, 4.0, 0.05, 0.3, 1.7, 1e-3, 37.0, 0.1]))
    lo = float(rng.choice([0.0, 0.0, -1.5, 2.0, -100.0, 1e-3]))
    hi = lo + float(rng.uniform(0.05, 0.6 * max(mw, mh))) * gw
    lum = [(0.299, 0.587, 0.114), (1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (2.0, -1.0, 0.3)][int(rng.randint(0, 4))]
    params = hm.SceneParams.make(lo, hi, grid_width=gw, lum=lum)
    ex, ey = mw * gw, mh * gw
    proj = int(rng.choice([1, 2, 3]))
    mode = int(rng.randint(0, 6))
    ang = rng.uniform(0, 2 * np.pi)
    dist = rng.uniform(0.05, 2.5) * max(ex, ey)
    pos = [ex / 2 + dist * np.cos(ang), -ey / 2 + dist * np.sin(ang), hi + rng.uniform(-1.0, 3.0) * (hi - lo + gw)]
    hang = float(np.arctan2(-ey / 2 - pos[1], ex / 2 - pos[0]) + rng.uniform(-0.6, 0.6))
    vang = float(hm.degrees_to_rads(rng.uniform(30, 178)))
    if mode == 0:   # camera inside the box
        pos = [rng.uniform(0, ex), -rng.uniform(0, ey), rng.uniform(lo, hi)]
    elif mode == 1:  # axis-parallel view directions
        hang = float(rng.choice([0.0, np.pi / 2, np.pi, -np.pi / 2]))
        vang = float(rng.choice([np.pi / 2, np.pi, np.pi * 0.75]))
    elif mode == 2:  # straight down from above the map
        pos = [rng.uniform(0, ex), -rng.uniform(0, ey), hi + rng.uniform(0.1, 5.0) * (hi - lo)]
        vang = float(np.pi)
    sd = float(rng.choice([0.05, 0.1, 0.25, 0.5, 1.0, 0.37, 3.0, 0.013]) * gw)
    if rng.randint(0, 25) == 0:
        sd = -sd
    cam = hm.Camera.make(width=int(rng.randint(1, 80)), height=int(rng.randint(1, 60)), projection=proj,
                         hfov=float(hm.degrees_to_rads(rng.uniform(5, 179))), hang=hang, vang=vang, pos=tuple(pos),
                         ortho_width=float(rng.uniform(0.05, 4.0) * gw), step_dist=sd,
                         bg=tuple(int(v) for v in rng.randint(0, 256, size=3)), sampling=int([0, 0, 0, 1, 2][int(rng.randint(0, 5))]))
    heights = oracle.update_heightmap(rgb, params)
    cfg = oracle.make_cfg(cam, params, mw, mh, step_cap=300000)
    ofb, total, capped, osteps, oentry = oracle.render(cfg, heights, cmap, per_pixel=True)
    scene = hm.Scene(rgb, cmap, params)
    fb, st, steps, entry = scene.render_stats(cam, per_pixel=True, allow_capped=True)
    ok = np.array_equal(entry.view(np.uint64), oentry.view(np.uint64)) and st.capped == capped
    if capped == 0:
        ok = ok and np.array_equal(fb, ofb) and np.array_equal(steps.astype(np.int64), osteps) and st.steps == total
        try:
            ok = ok and np.array_equal(scene.render(cam), ofb)
        except hm.HmrmError:
            ok = False
    else:
        capped_scenes += 1
        live = osteps >= 0  # (the oracle marks a capped ray with -(cap+1)) rays that did not reach the cap must agree
        ok = ok and np.array_equal(fb[live], ofb[live]) and np.array_equal(steps.astype(np.int64)[live], osteps[live])
    if not ok and os.environ.get("HMRM_FUZZ_VERBOSE"):
        live = osteps >= 0
        print("  entry equal", np.array_equal(entry.view(np.uint64), oentry.view(np.uint64)), "capped", st.capped, capped,
              "frame diff px", int((fb != ofb).any(axis=2).sum()), "of which live", int(((fb != ofb).any(axis=2) & live).sum()),
              "steps diff", int((steps.astype(np.int64) != osteps).sum()), "of which live", int(((steps.astype(np.int64) != osteps) & live).sum()))
        yy, xx = np.nonzero((steps.astype(np.int64) != osteps) | (fb != ofb).any(axis=2))
        for y, x in list(zip(yy, xx))[:5]:
            print("   px", (x, y), "gpu steps", int(steps[y, x]), "oracle", int(osteps[y, x]), "gpu rgba", fb[y, x].tolist(), "oracle", ofb[y, x].tolist(), "entry", entry[y, x])
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "map", (mw, mh), "gw", gw, "heights", (lo, hi), "proj", proj, "mode", mode, "sd", sd,
              "res", (cam.width, cam.height), "sampling", cam.sampling, "capped", capped, st.capped, flush=True)
    leaped += st.leaped_steps
    done += 1
    if done % 2000 == 0:
        print("... %d scenes, %d mismatches, %.0f s" % (done, bad, time.time() - t0), flush=True)
    scene.close()
print("scenes %d, mismatches %d, scenes with capped rays %d, leaped steps %d, %.0f s" % (done, bad, capped_scenes, leaped, time.time() - t0))
sys.exit(1 if bad else 0)
