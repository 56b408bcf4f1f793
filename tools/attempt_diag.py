"""Why leap attempts fail and at which pyramid level (instrumented kernel, HMRM_DIAG_ITERS=4..20)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hm = importlib.import_module("heightmap-ray-marcher_amd")
for name in sys.argv[1:] or ["C3", "C5"]:
    wl = hm.synth.WORKLOADS[name] if "/" not in name else hm.synth.content_workload(*name.split("/"))  # e.g. C3/white
    rgb, cmap = wl.maps()
    gw = float(os.environ.get("HMRM_DIAG_GW", "1"))  # the same scene scaled to another grid width
    params = wl.scene_params()
    cam = wl.camera()
    if gw != 1.0:
        size = wl.map_size * gw
        params = hm.SceneParams.make(0.0, size / 16.0, grid_width=gw)
        cam.pos[0], cam.pos[1], cam.pos[2] = cam.pos[0] * gw, cam.pos[1] * gw, cam.pos[2] * gw
        cam.step_dist = wl.step_dist * gw
        cam.ortho_width = cam.ortho_width * gw
    if os.environ.get("HMRM_DIAG_CAM") == "down":  # perspective, 60 degrees, looking down at 140: every ray hits
        cam.projection, cam.hfov, cam.vang = 1, hm.degrees_to_rads(60), hm.degrees_to_rads(140.0)
        cam.pos[0], cam.pos[1], cam.pos[2] = 1000.0 * gw, -1000.0 * gw, 1500.0 * gw
    scene = hm.Scene(rgb, cmap, params)
    _, st, *_ = scene.render_stats(cam)
    print(name, "attempts", st.leap_attempts, "leaps", st.leaps, "groups", st.groups, "leaped", st.leaped_steps, "steps", st.steps)
    for mode, labels in ((4, ("below max", "short z-bound", "short lateral", "verify failed")), (5, ("att L0", "att L1", "att L2", "att L3")),
                         (6, ("ok L0", "ok L1", "ok L2", "ok L3")), (7, ("steps L0", "steps L1", "steps L2", "steps L3")),
                         (9, ("shortlat L0", "L1", "L2", "L3+")), (10, ("refused: landing off the map", "outside the window", "below the maximum", "binade tests")),
                         (11, ("below L0", "L1", "L2", "L3")),
                         (12, ("wave iterations", "with attempt block", "with group block", "active lanes")),
                         (13, ("attempt lane slots", "attempt useful", "group lane slots", "group useful")),
                         (14, ("attempt iters share<1/8", "1/8..1/4", "1/4..1/2", ">=1/2")),
                         (15, ("group iters share<1/8", "1/8..1/4", "1/4..1/2", ">=1/2")),
                         (21, ("jumps limited by the window maximum", "jumps limited otherwise", "steps of the former", "steps of the latter")),
                         (22, ("z-limited jumps L0", "L1", "L2", "L3+")),
                         (23, ("attempts L3-4", "L5-6", "whole map", "steps jumped at whole-map level")),
                         (20, ("groups after a jump to a binade's end", "after an attempt without binade room", "without an attempt", "after other attempts"))):
        os.environ["HMRM_DIAG_ITERS"] = str(mode)
        cam.bg_r = mode  # defeat the frame cache
        _, s2, *_ = scene.render_stats(cam)
        print("  ", dict(zip(labels, (s2.leap_attempts, s2.leaps, s2.groups, s2.leaped_steps))))
    os.environ.pop("HMRM_DIAG_ITERS")
    scene.close()
