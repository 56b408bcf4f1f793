"""Long seeded fuzz of the record kernel (HMRM_KERNEL=rec; frame.hpp WindowRecord) against the CPU oracle on the maps it is
for: flat or rolling ground with tall single cells at densities 1/4 .. 1/500, ties among the tall cells, maps 1 .. 300 cells
wide (clipped windows), every projection, all three kinds of grid width, cameras inside / above / beside the box.
Test infrastructure, not collected by pytest: python tests/deep_fuzz_records.py <first seed> <scenes> [seconds] [updates].

With `updates` every scene lives through three more hmrm_scene_update calls (other height range, other luminance weights)
and is compared with the oracle after each: the records are built on demand (api.cpp ensure_records), so a stale table
would show here.  Odd seeds run that mode with HMRM_KERNEL unset and eight moving-camera frames per update, so that the
scene's own probe (launch_order.cpp) makes the choice between the kernels -- and makes it again after every update."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
os.environ["HMRM_KERNEL"] = "rec"
os.environ["HMRM_STEP_CAP"] = "300000"
hm = importlib.import_module("heightmap-ray-marcher_amd")
from oracle import oracle_py as oracle

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
budget_s = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
updates_mode = len(sys.argv) > 4 and sys.argv[4] == "updates"
updated = probe_frames = 0
t0 = time.time()
bad = done = leaped = jumps = 0
for k in range(count):
    if time.time() - t0 > budget_s:
        break
    seed = seed0 + k
    rng = np.random.RandomState(seed)
    mw = int(rng.choice([1, 2, 3, 5, 16, 17, 33, 64, 100, 129, 256, 300]))
    mh = int(rng.choice([1, 2, 4, 15, 16, 31, 64, 128, 257]))
    ground = int(rng.randint(0, 120))
    v = np.full((mh, mw), ground, dtype=np.int64)
    style = int(rng.randint(0, 4))
    yy, xx = np.mgrid[0:mh, 0:mw]
    if style == 1:
        v = (ground + 14 * np.sin(xx / rng.uniform(3, 20)) * np.cos(yy / rng.uniform(3, 20))).astype(np.int64)
    elif style == 2:
        v = ground + (xx + yy) % int(rng.randint(2, 9))
    elif style == 3:
        v = ground + rng.randint(0, 6, size=(mh, mw))
    dens = float(rng.choice([1 / 4, 1 / 8, 1 / 30, 1 / 64, 1 / 200, 1 / 500]))
    tall = rng.rand(mh, mw) < dens
    tops = rng.randint(150, 256, size=int(tall.sum())) if k % 2 else np.full(int(tall.sum()), int(rng.choice([200, 255])))
    v[tall] = tops
    v8 = np.clip(v, 0, 255).astype(np.uint8)
    rgb = np.ascontiguousarray(np.repeat(v8[:, :, None], 3, axis=2))
    cmap = rng.randint(0, 256, size=(mh, mw, 4)).astype(np.uint8)
    gw = float(rng.choice([1.0, 1.0, 0.5, 2.0, 0.05, 0.3, 1.7, 37.0, 0.01]))
    lo = float(rng.choice([0.0, 0.0, -1.5, 2.0, -100.0]))
    hi = lo + float(rng.uniform(0.5, 0.4 * max(mw, mh, 8))) * gw
    params = hm.SceneParams.make(lo, hi, grid_width=gw)
    ex, ey = mw * gw, mh * gw
    proj = int(rng.choice([1, 2, 3]))
    mode = int(rng.randint(0, 5))
    ang = rng.uniform(0, 2 * np.pi)
    dist = rng.uniform(0.0, 1.8) * max(ex, ey)
    zc = 2 * lo + (hi - lo) * float(rng.uniform(0.05, 1.6))   # between the ground and the tops as often as above them
    pos = [ex / 2 + dist * np.cos(ang), -ey / 2 + dist * np.sin(ang), zc]
    hang = float(np.arctan2(-ey / 2 - pos[1], ex / 2 - pos[0]) + rng.uniform(-0.6, 0.6))
    vang = float(hm.degrees_to_rads(rng.uniform(60, 160)))
    if mode == 0:    # axis-parallel, level: rays run along rows / columns of cells, grazing cell boundaries
        hang = float(rng.choice([0.0, np.pi / 2, np.pi, -np.pi / 2]))
        vang = float(np.pi / 2)
    elif mode == 1:  # towards the origin corner: rays leave through the low edges (coordinates in (-1, 0))
        pos = [ex * rng.uniform(0.3, 1.5), -ey * rng.uniform(0.3, 1.5), zc]
        hang = float(np.arctan2(0.0 - pos[1], 0.0 - pos[0]) + rng.uniform(-0.3, 0.3))
    sd = float(rng.choice([0.05, 0.1, 0.25, 0.5, 1.0, 0.37, 3.0]) * gw)
    cam = hm.Camera.make(width=int(rng.randint(8, 96)), height=int(rng.randint(8, 64)), projection=proj,
                         hfov=float(hm.degrees_to_rads(rng.uniform(20, 179))), hang=hang, vang=vang, pos=tuple(pos),
                         ortho_width=float(rng.uniform(0.05, 4.0) * gw), step_dist=sd,
                         bg=tuple(int(b) for b in rng.randint(0, 256, size=3)))
    heights = oracle.update_heightmap(rgb, params)
    cfg = oracle.make_cfg(cam, params, mw, mh, step_cap=300000)
    ofb, total, capped, osteps, oentry = oracle.render(cfg, heights, cmap, per_pixel=True)
    scene = hm.Scene(rgb, cmap, params)
    fb, st, steps, entry = scene.render_stats(cam, per_pixel=True, allow_capped=True)
    ok = np.array_equal(fb, ofb) and st.capped == capped
    if capped == 0:
        ok = ok and np.array_equal(steps.astype(np.int64), osteps) and st.steps == total
        ok = ok and np.array_equal(scene.render(cam), ofb)
    if not ok:
        bad += 1
        diff = np.argwhere(np.any(fb != ofb, axis=2))
        print(f"MISMATCH seed {seed} map {(mw, mh)} gw {gw} heights {(lo, hi)} proj {proj} mode {mode} style {style} dens {dens:.4f} sd {sd} "
              f"res {(cam.width, cam.height)} capped {capped} {st.capped} px diff {len(diff)} first {diff[:3].tolist()}", flush=True)
    if updates_mode:
        free_choice = bool(seed & 1)
        camu = cam
        if free_choice:  # (the probe wants a full frame of at least 12 tile rows: api.cpp launch_frame `eligible`)
            del os.environ["HMRM_KERNEL"]
            camu = hm.Camera.make(width=cam.width, height=208, projection=proj, hfov=cam.hfov, hang=hang, vang=vang, pos=tuple(pos),
                                  ortho_width=cam.ortho_width, step_dist=sd, bg=(cam.bg_r, cam.bg_g, cam.bg_b))
        for u in range(3):
            lo2 = lo + float(rng.uniform(-0.5, 0.5)) * gw
            p2 = hm.SceneParams.make(lo2, lo2 + float(rng.uniform(0.5, 0.4 * max(mw, mh, 8))) * gw, grid_width=gw,
                                     lum=tuple(float(x) for x in rng.dirichlet([1.0, 1.0, 1.0])))
            scene.update(p2)
            h2 = oracle.update_heightmap(rgb, p2)
            cfg2 = oracle.make_cfg(camu, p2, mw, mh, step_cap=300000)
            ofb2, total2, capped2, osteps2, _ = oracle.render(cfg2, h2, cmap, per_pixel=True)
            ok2 = capped2 > 0 or np.array_equal(scene.render(camu), ofb2)
            if free_choice and capped2 == 0:  # never-repeating cameras: the sixth full frame carries the scene's probe
                for j in range(8):
                    camj = hm.Camera.make(width=camu.width, height=camu.height, projection=proj, hfov=cam.hfov, hang=hang + 1e-3 * (j + 1),
                                          vang=vang, pos=tuple(pos), ortho_width=cam.ortho_width, step_dist=sd, bg=(1, 2, 3))
                    try:
                        scene.render(camj)
                    except hm.HmrmError as e:  # (a neighbouring pose may hold a ray that never ends: reported, not a mismatch)
                        if e.code != hm.HMRM_E_NOTERM:
                            raise
                    probe_frames += 1
                ok2 = ok2 and np.array_equal(scene.render(camu), ofb2)
            fb2, st2, steps2, _ = scene.render_stats(camu, per_pixel=True, allow_capped=True)
            ok2 = ok2 and np.array_equal(fb2, ofb2) and st2.capped == capped2
            if capped2 == 0:
                ok2 = ok2 and np.array_equal(steps2.astype(np.int64), osteps2) and st2.steps == total2
            updated += 1
            if not ok2:
                bad += 1
                print(f"MISMATCH after update {u} seed {seed} map {(mw, mh)} gw {gw} heights {(p2.min_height, p2.max_height)} proj {proj} "
                      f"free_choice {free_choice} res {(camu.width, camu.height)} capped {capped2}", flush=True)
        os.environ["HMRM_KERNEL"] = "rec"
    leaped += st.leaped_steps
    jumps += st.leaps
    done += 1
    scene.close()
    if done % 200 == 0:
        print(f"... {done} scenes, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"records: scenes {done}, mismatches {bad}, jumps {jumps}, leaped steps {leaped}, {time.time() - t0:.0f} s"
      + (f"; height updates checked {updated}, frames under the scene's own kernel choice {probe_frames}" if updates_mode else ""))
sys.exit(1 if bad else 0)
