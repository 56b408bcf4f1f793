"""Third fuzzer: cameras aimed so that the rays of a frame GRAZE an edge or a corner of the box -- neighbouring rays
pass it at relative offsets from 2^-14 down to 2^-52, on both sides -- the neighbourhood in which the kernel's
one-division shortcut through distance() (csrc/device_common.hpp slab_classify, AABB.cpp:49-77) must hand over to the
reference's six divisions.  GPU vs oracle on distance() bits, frames, per-ray step counts, all three projections.
Test infrastructure, not collected by pytest: python tests/deep_fuzz_edges.py <first seed> <frames> [seconds].
`grazing_case(seed)` is also what tests/test_parity_gpu.py::test_rays_grazing_box_edges_and_corners iterates."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
hm = importlib.import_module("heightmap-ray-marcher_amd")

MAP_CHOICES = ((64, 64), (48, 80), (96, 33))


def grazing_case(seed):
    """-> (map_w, map_h, map_seed, SceneParams, Camera, log2 of the pixel-to-pixel angle / offset)."""
    rng = np.random.RandomState(seed)
    mw, mh = MAP_CHOICES[int(rng.randint(len(MAP_CHOICES)))]
    gw = float(rng.choice([1.0, 0.5, 0.05, 0.3, 64.0]))
    lo = float(rng.choice([0.0, 0.0, -3.0, 2.5]))
    hi = lo + float(rng.uniform(0.05, 0.5)) * max(mw, mh) * gw
    params = hm.SceneParams.make(lo, hi, grid_width=gw)
    c0 = np.array([0.0, 0.0, lo])
    c1 = np.array([mw * gw, -(mh * gw), hi])
    ext = float(max(mw, mh) * gw)
    # target: a corner, or a point on an edge
    target = np.array([(c0, c1)[int(rng.randint(2))][i] for i in range(3)])
    kind = int(rng.randint(3))
    if kind >= 1:
        ax = int(rng.randint(3))
        target[ax] = c0[ax] + rng.uniform(0.02, 0.98) * (c1[ax] - c0[ax])
    # camera outside the box, any side, any height
    ang, dist = rng.uniform(0, 2 * np.pi), rng.uniform(0.8, 3.0) * ext
    pos = np.array([c1[0] / 2 + dist * np.cos(ang), c1[1] / 2 + dist * np.sin(ang),
                    lo + rng.uniform(-1.0, 4.0) * (hi - lo + gw)])
    v = target - pos
    n = float(np.sqrt((v * v).sum()))
    vang = float(np.arccos(v[2] / n))
    hang = float(np.arctan2(v[1], v[0]))
    k = float(rng.uniform(14, 52))
    a = 2.0 ** -k   # angle (or relative offset) between neighbouring rays
    W, H = int(rng.randint(24, 64)), int(rng.randint(20, 48))
    proj = 1 + seed % 3
    if proj == 3:
        # parallel rays: the centre pixel's ray passes through the target, neighbours a * n apart
        look = np.array([np.sin(vang) * np.cos(hang), np.sin(vang) * np.sin(hang), np.cos(vang)])
        pos = target - look * n
        ow = a * n
    else:
        ow = 0.1
    cam = hm.Camera.make(width=W, height=H, projection=proj, hfov=a * (W - 1), hang=hang, vang=vang, pos=tuple(pos),
                         ortho_width=ow, step_dist=float(rng.choice([0.25, 0.5, 0.13]) * gw),
                         bg=(7, 8, 9), sampling=0)
    return mw, mh, 1000 + seed % 5, params, cam, -k


def straddles(oentry):
    """The frame holds rays on both sides of a silhouette edge (hits and misses of the box)."""
    fin = np.isfinite(oentry)
    return bool(fin.any() and (~fin).any())


def run_case(hm_scene_cache, oracle, scenes, seed):
    mw, mh, mseed, params, cam, lg = grazing_case(seed)
    key = (mw, mh, mseed, tuple(bytes(params)))
    if key not in hm_scene_cache:
        rgb, cmap = scenes.small_maps(mw, mh, mseed)
        hm_scene_cache.clear()  # (one live scene at a time)
        hm_scene_cache[key] = (hm.Scene(rgb, cmap, params), oracle.update_heightmap(rgb, params), cmap)
    scene, heights, cmap = hm_scene_cache[key]
    cfg = oracle.make_cfg(cam, params, mw, mh)
    ofb, total, capped, osteps, oentry = oracle.render(cfg, heights, cmap, per_pixel=True)
    fb, st, steps, entry = scene.render_stats(cam, per_pixel=True, allow_capped=True)
    ok = (np.array_equal(entry.view(np.uint64), oentry.view(np.uint64)) and st.capped == capped
          and np.array_equal(fb, ofb) and np.array_equal(steps.astype(np.int64), osteps)
          and np.array_equal(scene.render(cam), ofb))
    return ok, straddles(oentry), cam, lg


if __name__ == "__main__":
    from oracle import oracle_py as oracle
    import scenes
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    budget_s = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
    t0 = time.time()
    bad = done = strad = 0
    cache = {}
    for i in range(count):
        if time.time() - t0 > budget_s:
            break
        ok, s, cam, lg = run_case(cache, oracle, scenes, seed0 + i)
        strad += int(s)
        done += 1
        if not ok:
            bad += 1
            print("MISMATCH seed", seed0 + i, "proj", cam.projection, "res", (cam.width, cam.height), "log2 offset", lg, flush=True)
        if done % 1000 == 0:
            print("... %d frames, %d mismatches, %.0f s" % (done, bad, time.time() - t0), flush=True)
    print("edge-grazing: cameras %d, mismatches %d, frames straddling a silhouette edge %d, %.0f s" % (done, bad, strad, time.time() - t0))
    sys.exit(1 if bad else 0)
