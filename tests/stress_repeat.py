"""Hunt for NONDETERMINISM: small random scenes (the shapes of deep_fuzz_cells.py) are created, rendered with the instrumented and the
plain kernel many times and destroyed again, thousands per minute; every render of a scene must equal that scene's first one, bit for
bit (frames, per-ray step counts, distance() bits).  No oracle involved: this looks for races -- copies, tables, stale buffers --,
not for arithmetic.  Test infrastructure: python tests/stress_repeat.py <first seed> <seconds> [renders per scene]."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
hm = importlib.import_module("heightmap-ray-marcher_amd")

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
t0 = time.time()
scenes = renders = bad = 0
k = 0
while time.time() - t0 < budget_s:
    rng = np.random.RandomState(seed0 + k)
    k += 1
    mw, mh = int(rng.choice([64, 200, 513])), int(rng.choice([64, 200, 513]))
    rgb = np.repeat(rng.randint(0, 40, size=(mh, mw, 1)).astype(np.uint8), 3, axis=2)
    for _ in range(3):
        tx, ty = int(rng.randint(0, mw)), int(rng.randint(0, mh))
        rgb[max(ty - 2, 0):ty + 3, max(tx - 2, 0):tx + 3] = int(rng.randint(120, 256))
    cmap = rng.randint(0, 256, size=(mh, mw, 4)).astype(np.uint8)
    cmap[:, :, 3] = 255
    gw = float(rng.choice([0.05, 0.07, 1.0, 0.3, 0.5]))
    hi = float(rng.choice([4.0, 20.0, 90.0])) * gw
    params = hm.SceneParams.make(0.0, hi, grid_width=gw)
    W, H = int(rng.randint(8, 40)), int(rng.randint(4, 16))
    ex, ey = mw * gw, mh * gw
    side = int(rng.randint(0, 4))
    d = float(2.0 ** int(rng.randint(-3, 4)))
    pos = [[-d, -ey / 2, hi * 1.5], [ex + d, -ey / 2, hi * 1.5], [ex / 2, d, hi * 1.5], [ex / 2, -ey - d, hi * 1.5]][side]
    hang = [0.0, np.pi, -np.pi / 2, np.pi / 2][side] + float(rng.choice([0.0, 0.01, -0.3]))
    cam = hm.Camera.make(width=W, height=H, projection=int(rng.choice([1, 2, 2])), hfov=float(hm.degrees_to_rads(rng.uniform(2.0, 70.0))),
                         hang=float(hang), vang=float(hm.degrees_to_rads(rng.uniform(95.0, 135.0))), pos=tuple(pos),
                         step_dist=float(rng.choice([0.25, 0.5, 0.37])) * gw, bg=(1, 2, 3))
    scene = hm.Scene(rgb, cmap, params)
    fb0, st0, steps0, entry0 = scene.render_stats(cam, per_pixel=True, allow_capped=True)
    for r in range(reps):
        fb, st, steps, entry = scene.render_stats(cam, per_pixel=True, allow_capped=True)
        plain = scene.render(cam) if st0.capped == 0 else fb
        renders += 2
        same = np.array_equal(fb, fb0) and np.array_equal(plain, fb0) and np.array_equal(steps, steps0) and np.array_equal(entry.view(np.uint64), entry0.view(np.uint64))
        if not same:
            bad += 1
            print("NONDETERMINISM seed", seed0 + k - 1, "render", r, "map", (mw, mh), "gw", gw, "proj", cam.projection, "res", (W, H),
                  "stats frame px", int((fb != fb0).any(axis=2).sum()), "plain frame px", int((plain != fb0).any(axis=2).sum()),
                  "steps", int((steps != steps0).sum()), "entry", int((entry.view(np.uint64) != entry0.view(np.uint64)).sum()), flush=True)
            for (y, x) in np.argwhere((fb != fb0).any(axis=2))[:4]:
                print("   stats px", (int(x), int(y)), fb[y, x].tolist(), "first render", fb0[y, x].tolist(), flush=True)
            for (y, x) in np.argwhere((plain != fb0).any(axis=2))[:4]:
                print("   plain px", (int(x), int(y)), plain[y, x].tolist(), "first render", fb0[y, x].tolist(), flush=True)
    scene.close()
    scenes += 1
print("repeat stress: scenes %d, renders %d, nondeterministic %d, %.0f s" % (scenes, renders, bad, time.time() - t0))
sys.exit(1 if bad else 0)
