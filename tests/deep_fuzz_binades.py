"""Seeded fuzz aimed at binade crossings (csrc/render_fast.hip HMRM_CROSS: a jump ends with one real step that carries a
coordinate into its next binade): long, low maps crossed end to end by shallow rays -- a dozen binades of x or y towards
the map's origin or away from it, several of z on the way down -- with step sizes that include exact rounding ties,
power-of-two and general grid widths.  GPU vs CPU oracle on frames, per-ray step counts and distance() bits.
Test infrastructure, not collected by pytest: python tests/deep_fuzz_binades.py <first seed> <scenes> [seconds]."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
hm = importlib.import_module("heightmap-ray-marcher_amd")
from oracle import oracle_py as oracle

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
budget_s = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
os.environ["HMRM_STEP_CAP"] = "2000000"
t0 = time.time()
bad = done = leaped = jumps = 0
for k in range(count):
    if time.time() - t0 > budget_s:
        break
    seed = seed0 + k
    rng = np.random.RandomState(seed)
    long_side = int(rng.choice([512, 1000, 2048, 4096]))
    short_side = int(rng.choice([2, 5, 16]))
    along_x = bool(rng.randint(0, 2))
    mw, mh = (long_side, short_side) if along_x else (short_side, long_side)
    low = rng.randint(0, 25, size=(mh, mw, 1)).astype(np.uint8)          # low terrain: long jumps
    rgb = np.repeat(low, 3, axis=2)
    wall = int(rng.randint(2, long_side - 2))                             # one wall somewhere stops most rays
    if along_x:
        rgb[:, wall:wall + 3] = 255
    else:
        rgb[wall:wall + 3, :] = 255
    cmap = rng.randint(0, 256, size=(mh, mw, 4)).astype(np.uint8)
    cmap[:, :, 3] = 255
    gw = float(rng.choice([1.0, 1.0, 0.5, 0.125, 0.3, 0.05, 3.0, 1.7]))
    hi = float(rng.choice([8.0, 40.0, 300.0])) * gw
    params = hm.SceneParams.make(0.0, hi, grid_width=gw)
    # step: a plain fraction of a cell, or one that makes x + s an exact rounding tie in some binade on the way
    base = float(rng.choice([0.25, 0.11, 0.5, 0.37]))
    sd = base * gw
    if rng.randint(0, 3) == 0:
        sd = (0.25 + 2.0 ** -int(rng.randint(40, 46))) * gw
    ex, ey = mw * gw, mh * gw
    toward_origin = bool(rng.randint(0, 2))
    # the camera sits past one end of the long side and looks along it, a few degrees below the horizon
    far = float(rng.uniform(1.0, 30.0)) * gw
    mid_short = float(rng.uniform(0.2, 0.8))
    if along_x:
        pos = [ex + far if toward_origin else -far, -ey * mid_short, hi * float(rng.uniform(0.3, 1.6))]
        hang = np.pi if toward_origin else 0.0
    else:
        pos = [ex * mid_short, far if toward_origin else -ey - far, hi * float(rng.uniform(0.3, 1.6))]
        hang = -np.pi / 2 if toward_origin else np.pi / 2
    hang = float(hang + rng.uniform(-0.002, 0.002) * (short_side / long_side) * 50)
    vang = float(hm.degrees_to_rads(90.0 + rng.choice([0.05, 0.3, 1.0, 4.0])))
    proj = int(rng.choice([1, 2, 3]))
    cam = hm.Camera.make(width=int(rng.randint(4, 24)), height=int(rng.randint(3, 14)), projection=proj,
                         hfov=float(hm.degrees_to_rads(rng.uniform(0.5, 12.0))), hang=hang, vang=vang, pos=tuple(pos),
                         ortho_width=float(rng.uniform(0.2, 0.9) * ey if along_x else rng.uniform(0.2, 0.9) * ex), step_dist=sd,
                         bg=(1, 2, 3), sampling=int([0, 0, 0, 1, 2][int(rng.randint(0, 5))]))
    heights = oracle.update_heightmap(rgb, params)
    cfg = oracle.make_cfg(cam, params, mw, mh, step_cap=2000000)
    ofb, total, capped, osteps, oentry = oracle.render(cfg, heights, cmap, per_pixel=True)
    scene = hm.Scene(rgb, cmap, params)
    fb, st, steps, entry = scene.render_stats(cam, per_pixel=True, allow_capped=True)
    ok = np.array_equal(entry.view(np.uint64), oentry.view(np.uint64)) and st.capped == capped
    if capped == 0:
        ok = ok and np.array_equal(fb, ofb) and np.array_equal(steps.astype(np.int64), osteps) and st.steps == total
        ok = ok and np.array_equal(scene.render(cam), ofb)
    else:
        live = osteps >= 0
        ok = ok and np.array_equal(fb[live], ofb[live]) and np.array_equal(steps.astype(np.int64)[live], osteps[live])
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "map", (mw, mh), "gw", gw, "hi", hi, "proj", proj, "sd", repr(sd), "toward origin", toward_origin,
              "res", (cam.width, cam.height), "sampling", cam.sampling, "capped", capped, st.capped,
              "steps diff", int((steps.astype(np.int64) != osteps).sum()), "px diff", int((fb != ofb).any(axis=2).sum()), flush=True)
    leaped += st.leaped_steps
    jumps += st.leaps
    done += 1
    if done % 200 == 0:
        print("... %d scenes, %d mismatches, %.0f s" % (done, bad, time.time() - t0), flush=True)
    scene.close()
print("binade crossings: scenes %d, mismatches %d, jumps %d, leaped steps %d, %.0f s" % (done, bad, jumps, leaped, time.time() - t0))
sys.exit(1 if bad else 0)
