"""Seeded fuzz aimed at CELL BOUNDARIES under general grid widths (csrc/leap_common.hpp cell_coord_fast<2>, the attempt
block of csrc/render_fast.hip): grid widths whose reciprocal rounds to an integer (0.05, 0.01, 0.2, 0.1, 1e-3: every
power of two is then a cell boundary to the last bit) and ones where it does not, with rays that run ALONG cell
boundaries (orthographic cameras looking down an axis, columns a fraction of a cell apart from a corner that is a multiple
of the cell), rays whose steps are exact fractions of a cell (positions ON a boundary every few steps), and perspective /
spherical cameras at dyadic positions.  Every such position is `near` (within 2^-20 of a boundary): the start of an
attempt divides for real, a landing point is accepted only when both candidate cells lie inside the window, every sampled
position divides.  GPU vs CPU oracle on frames, per-ray step counts and distance() bits.
Test infrastructure, not collected by pytest: python tests/deep_fuzz_cells.py <first seed> <scenes> [seconds]."""
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
    mw, mh = int(rng.choice([64, 200, 513, 1024])), int(rng.choice([64, 200, 513, 1024]))
    base = rng.randint(0, 40, size=(mh, mw, 1)).astype(np.uint8)            # low terrain: long jumps
    rgb = np.repeat(base, 3, axis=2)
    for _ in range(int(rng.randint(1, 6))):                                  # a few towers that stop rays
        tx, ty = int(rng.randint(0, mw)), int(rng.randint(0, mh))
        rgb[max(ty - 2, 0):ty + 3, max(tx - 2, 0):tx + 3] = int(rng.randint(120, 256))
    cmap = rng.randint(0, 256, size=(mh, mw, 4)).astype(np.uint8)
    cmap[:, :, 3] = 255
    gw = float(rng.choice([0.05, 0.01, 0.2, 0.1, 1e-3, 0.3, 0.07, 3.0, 1.7, 0.05, 0.01]))
    hi = float(rng.choice([4.0, 20.0, 90.0])) * gw
    params = hm.SceneParams.make(0.0, hi, grid_width=gw)
    frac = float(rng.choice([0.25, 0.5, 0.125, 1.0, 0.2, 0.37]))              # step as a fraction of a cell
    sd = frac * gw
    ex, ey = mw * gw, mh * gw
    kind = int(rng.randint(0, 4))
    W, H = int(rng.randint(8, 40)), int(rng.randint(4, 16))
    vang = float(hm.degrees_to_rads(90.0 + rng.choice([0.0, 0.5, 3.0, 20.0, 45.0])))
    if kind <= 1:
        # orthographic along an axis; columns q cells apart from a corner on a multiple of the cell: rays run along boundaries
        axis = int(rng.randint(0, 4))
        hang = [0.0, np.pi, np.pi / 2, -np.pi / 2][axis]
        q = float(rng.choice([0.25, 0.5, 1.0, 0.125]))
        ow = q * gw * (W - 1) / W                                               # column spacing = W * ow / (W - 1) = q cells
        c0 = float(rng.randint(0, max(1, (mw if axis >= 2 else mh) - int(q * W) - 1))) * gw
        centre = c0 + 0.5 * q * gw * (W - 1)
        far = float(rng.choice([1.0, 3.0, 0.5])) * gw * 4
        if axis == 0:
            pos = [-far, -centre, hi * float(rng.uniform(0.3, 1.4))]
        elif axis == 1:
            pos = [ex + far, -centre, hi * float(rng.uniform(0.3, 1.4))]
        elif axis == 2:
            pos = [centre, -ey - far, hi * float(rng.uniform(0.3, 1.4))]
        else:
            pos = [centre, far, hi * float(rng.uniform(0.3, 1.4))]
        cam = hm.Camera.make(width=W, height=H, projection=3, hang=float(hang), vang=vang, pos=tuple(pos), ortho_width=float(ow),
                             step_dist=sd, bg=(1, 2, 3), sampling=int([0, 0, 0, 2][int(rng.randint(0, 4))]))
    else:
        # perspective / spherical from a dyadic position outside the box, looking across it
        side = int(rng.randint(0, 4))
        d = float(2.0 ** int(rng.randint(-3, 4)))
        pos = [[-d, -ey / 2, hi * 1.5], [ex + d, -ey / 2, hi * 1.5], [ex / 2, d, hi * 1.5], [ex / 2, -ey - d, hi * 1.5]][side]
        pos = [float(np.round(v * 8) / 8) for v in pos]
        hang = [0.0, np.pi, -np.pi / 2, np.pi / 2][side] + float(rng.choice([0.0, 0.0, 0.01, -0.3]))
        cam = hm.Camera.make(width=W, height=H, projection=int(rng.choice([1, 2])), hfov=float(hm.degrees_to_rads(rng.uniform(2.0, 70.0))),
                             hang=float(hang), vang=float(hm.degrees_to_rads(rng.uniform(95.0, 135.0))), pos=tuple(pos), step_dist=sd,
                             bg=(1, 2, 3), sampling=int([0, 0, 0, 2][int(rng.randint(0, 4))]))
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
        print("MISMATCH seed", seed, "map", (mw, mh), "gw", gw, "hi", hi, "kind", kind, "proj", cam.projection, "sd", repr(sd),
              "res", (cam.width, cam.height), "sampling", cam.sampling, "capped", capped, st.capped,
              "steps diff", int((steps.astype(np.int64) != osteps).sum()), "px diff", int((fb != ofb).any(axis=2).sum()), flush=True)
        # what differs, and does it differ again?  (a mismatch that a second render of the same scene does not repeat is a race,
        # not arithmetic)
        for (r, c) in np.argwhere((fb != ofb).any(axis=2))[:4]:
            print("   stats frame px", (int(c), int(r)), "gpu", fb[r, c].tolist(), "oracle", ofb[r, c].tolist(), "steps", int(steps[r, c]), int(osteps[r, c]), flush=True)
        plain = scene.render(cam)
        for (r, c) in np.argwhere((plain != ofb).any(axis=2))[:4]:
            print("   plain frame px", (int(c), int(r)), "gpu", plain[r, c].tolist(), "oracle", ofb[r, c].tolist(), flush=True)
        for rep in range(2):
            fb2, _, steps2, _ = scene.render_stats(cam, per_pixel=True, allow_capped=True)
            print("   rendered again: px diff", int((fb2 != ofb).any(axis=2).sum()), "steps diff", int((steps2.astype(np.int64) != osteps).sum()),
                  "plain px diff", int((scene.render(cam) != ofb).any(axis=2).sum()), flush=True)
    leaped += st.leaped_steps
    jumps += st.leaps
    done += 1
    if done % 200 == 0:
        print("... %d scenes, %d mismatches, %.0f s" % (done, bad, time.time() - t0), flush=True)
    scene.close()
print("cell boundaries: scenes %d, mismatches %d, jumps %d, leaped steps %d, %.0f s" % (done, bad, jumps, leaped, time.time() - t0))
sys.exit(1 if bad else 0)
