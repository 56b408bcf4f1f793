"""GPU parity tests: the HIP path through the C ABI against the CPU oracle.

Bar: bit-exact (integer colour lookups AND every fp64 intermediate that is
observable: heights, distance(), per-ray step counts, ray directions).
Sizes the oracle finishes in seconds are compared directly; BASELINE.json's
full sizes are covered by row-subsampled oracle comparison and by
size-independent properties (determinism, strips == full frame, counters add up).
"""
import contextlib
import importlib
import os
import subprocess

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


# HMRM_KERNEL: "leap" = production kernel (speculative groups + exact leaps), "group" = groups
# only, "simple" = the literal one-step loop, "rec" = groups + leaps over window records (nearest sampling; the plain
# groups otherwise).  All must agree with the oracle bit for bit.
KERNEL_VARIANTS = ("leap", "group", "simple", "rec")
KERNEL_NAMES = ("the production kernel", "the plain groups", "the literal loop", "the groups with window records")  # Scene.kernel_choice()


@contextlib.contextmanager
def kernel_variant(name):
    old = os.environ.get("HMRM_KERNEL")
    os.environ["HMRM_KERNEL"] = name
    try:
        yield
    finally:
        if old is None:
            del os.environ["HMRM_KERNEL"]
        else:
            os.environ["HMRM_KERNEL"] = old


@contextlib.contextmanager
def env(**kw):
    """Temporarily set environment knobs (the Python wrappers make a live scene re-read them)."""
    old = {k: os.environ.get(k) for k in kw}
    os.environ.update({k: str(v) for k, v in kw.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def gpu(hmrm):
    assert hmrm.device_count() >= 1, "no GPU visible: these tests must run on the MI355X box"
    hmrm.set_device(0)
    return hmrm


@pytest.mark.parametrize("case", scenes.cases(), ids=scenes.case_ids())
def test_frame_steps_and_distance_bit_exact(gpu, oracle, case):
    name, rgb, cmap, params, cam = scenes.build_case(case)
    scene = gpu.Scene(rgb, cmap, params)
    heights = oracle.update_heightmap(rgb, params)
    assert np.array_equal(_bits(scene.read_heights()), _bits(heights)), "UpdateHeightmap (hmap.cpp:171-191)"
    cfg = oracle.make_cfg(cam, params, rgb.shape[1], rgb.shape[0])
    ofb, total, capped, osteps, oentry = oracle.render(cfg, heights, cmap, per_pixel=True)
    hits = None
    for variant in KERNEL_VARIANTS:
        with kernel_variant(variant):
            fb, st, steps, entry = scene.render_stats(cam, per_pixel=True)
            assert np.array_equal(_bits(entry), _bits(oentry)), f"{variant}: distance() (AABB.cpp:49-77)"
            assert np.array_equal(steps.astype(np.int64), osteps), f"{variant}: per-ray step count (hmap.cpp:1000-1038)"
            assert np.array_equal(fb, ofb), f"{variant}: frame (hmap.cpp:978-1058)"
            assert (st.rays, st.steps, st.capped) == (cam.width * cam.height, total, 0), variant
            assert hits is None or st.hits == hits
            hits = st.hits
            # the un-instrumented kernel writes the same pixels
            assert np.array_equal(scene.render(cam), ofb), variant
    scene.close()


def test_golden_frames(gpu):
    data = np.load(os.path.join(GOLDEN, "frames.npz"))
    for case in scenes.cases():
        name, rgb, cmap, params, cam = scenes.build_case(case)
        scene = gpu.Scene(rgb, cmap, params)
        fb, st, steps, entry = scene.render_stats(cam, per_pixel=True)
        assert np.array_equal(fb, data[name + "/frame"]), name
        assert np.array_equal(steps, data[name + "/steps"]), name
        assert np.array_equal(_bits(entry), data[name + "/entry_bits"]), name
        scene.close()


@pytest.mark.parametrize("case", scenes.cases()[:9], ids=scenes.case_ids()[:9])
def test_ray_generation_bit_exact(gpu, oracle, case):
    """ImagePlane::GetRay on the device (incl. device sqrt and division) vs the oracle."""
    name, rgb, cmap, params, cam = scenes.build_case(case)
    scene = gpu.Scene(rgb, cmap, params)
    cfg = oracle.make_cfg(cam, params, rgb.shape[1], rgb.shape[0])
    rng = np.random.RandomState(3)
    pix = [(0, 0), (cam.width - 1, 0), (0, cam.height - 1), (cam.width - 1, cam.height - 1)]
    pix += [(int(rng.randint(cam.width)), int(rng.randint(cam.height))) for _ in range(24)]
    for px, py in pix:
        pos, d, dist = scene.debug_ray(cam, px, py)
        opos, od, odist = oracle.probe_ray(cfg, px, py)
        assert np.array_equal(_bits(pos), _bits(opos)) and np.array_equal(_bits(d), _bits(od)), (px, py)
        assert _bits(dist) == _bits(odist), (px, py)
    scene.close()


def test_scene_update_recomputes_heights(gpu, oracle):
    rgb, cmap = scenes.small_maps(48, 40, 77, color_heights=True)
    p1 = gpu.SceneParams.make(0.0, 5.0, grid_width=0.5)
    p2 = gpu.SceneParams.make(-1.0, 9.0, lum=(0.2, 0.7, 0.1), grid_width=0.25)
    cam = gpu.Camera.make(width=64, height=48, hfov=gpu.degrees_to_rads(80), hang=gpu.degrees_to_rads(-45),
                          vang=gpu.degrees_to_rads(118), pos=(-4.0, 4.0, 14.0), step_dist=0.125, bg=(3, 2, 1))
    scene = gpu.Scene(rgb, cmap, p1)
    for p in (p1, p2, p1):
        scene.update(p)
        heights = oracle.update_heightmap(rgb, p)
        assert np.array_equal(_bits(scene.read_heights()), _bits(heights))
        ofb, *_ = oracle.render(oracle.make_cfg(cam, p, 48, 40), heights, cmap)
        assert np.array_equal(scene.render(cam), ofb)
    scene.close()


def test_row_strips_and_cyclic_bands_equal_full_frame(gpu):
    """Multi-GPU partitioning on one device: strips / bands reassemble to the frame."""
    import torch
    rgb, cmap = gpu.synth.synth_maps(256)
    wl = gpu.synth.WORKLOADS["C1"]
    params = wl.scene_params()
    scene = gpu.Scene(rgb, cmap, params)
    for proj in (1, 2, 3):
        cam = wl.camera()
        cam.projection = proj
        cam.width, cam.height = 203, 117  # ragged: not a multiple of the 16x16 tile
        cam.ortho_width = 1.5
        full = scene.render(cam)
        # contiguous strips
        parts = []
        for r0, r1 in ((0, 40), (40, 41), (41, 117)):
            buf = torch.zeros((r1 - r0, cam.width, 4), dtype=torch.uint8, device="cuda")
            scene.render_rows_device(cam, buf.data_ptr(), cam.width * 4, r0, r1,
                                     stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            parts.append(buf.cpu().numpy())
        assert np.array_equal(np.concatenate(parts, 0), full)
        # cyclic bands of 16 rows over 3 "ranks"
        band, n = 16, 3
        out = np.zeros_like(full)
        for rank in range(n):
            rows = gpu.band_local_rows(cam.height, band, rank, n)
            buf = torch.zeros((rows, cam.width, 4), dtype=torch.uint8, device="cuda")
            scene.render_rows_device(cam, buf.data_ptr(), cam.width * 4, band_rows=band, band_index=rank,
                                     band_count=n, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            host = buf.cpu().numpy()
            for k in range(rows // band):
                g0 = (rank + k * n) * band
                g1 = min(g0 + band, cam.height)
                if g0 < cam.height:
                    out[g0:g1] = host[k * band:k * band + (g1 - g0)]
        assert np.array_equal(out, full)
    scene.close()


@pytest.mark.parametrize("wl_name", ["C1", "C2"])
def test_baseline_config_full_frame_vs_oracle(gpu, oracle, wl_name):
    """BASELINE configs small enough for the oracle to render completely (C2: ~2 M rays)."""
    wl = gpu.synth.WORKLOADS[wl_name]
    rgb, cmap = gpu.synth.synth_maps(wl.map_size)
    params, cam = wl.scene_params(), wl.camera()
    scene = gpu.Scene(rgb, cmap, params)
    heights = oracle.update_heightmap(rgb, params)
    ofb, total, capped, *_ = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size), heights, cmap)
    fb, st, _, _ = scene.render_stats(cam)
    assert np.array_equal(fb, ofb)
    assert st.steps == total and capped == 0 and st.capped == 0
    scene.close()


@pytest.mark.parametrize("wl_name", ["C3", "C3h", "C5", "REFDEF"])
def test_baseline_config_full_size_subsampled_and_properties(gpu, oracle, wl_name):
    """Full BASELINE size (3840x2160 over 4096^2): every 24th row against the oracle, plus
    size-independent properties: determinism, instrumented == plain kernel, counters add up.
    REFDEF = the reference's own operating point (sample_config.txt:5-7: grid_width 0.01, step_dist 0.05 = 5 cells per
    step) on the C5 frame: a general grid width with multi-cell steps, where leaps barely apply."""
    wl = gpu.synth.WORKLOADS[wl_name]
    rgb, cmap = gpu.synth.synth_maps(wl.map_size)
    params, cam = wl.scene_params(), wl.camera()
    scene = gpu.Scene(rgb, cmap, params)
    fb, st, steps, entry = scene.render_stats(cam, per_pixel=True)
    fb2 = scene.render(cam)
    assert np.array_equal(fb, fb2), "instrumented and plain kernels differ"
    for variant in ("group", "simple", "rec"):
        with kernel_variant(variant):
            fbv, stv, stepsv, _ = scene.render_stats(cam, per_pixel=True)
        assert np.array_equal(fbv, fb), f"kernel variant {variant} differs from the production kernel"
        assert np.array_equal(stepsv, steps) and stv.steps == st.steps, variant
    assert np.array_equal(scene.render(cam), fb2), "render is not deterministic"
    assert int(steps.astype(np.int64).sum()) == st.steps and st.rays == cam.width * cam.height
    assert (fb[:, :, 3] == 255).all()
    heights = oracle.update_heightmap(rgb, params)
    assert np.array_equal(_bits(scene.read_heights()), _bits(heights))
    ofb = np.zeros_like(fb)
    cfg = oracle.make_cfg(cam, params, wl.map_size, wl.map_size)
    stride = 24
    ofb, total, capped, osteps, oentry = oracle.render(cfg, heights, cmap, per_pixel=True, row_stride=stride)
    assert capped == 0
    rows = slice(0, cam.height, stride)
    assert np.array_equal(fb[rows], ofb[rows])
    assert np.array_equal(steps[rows].astype(np.int64), osteps[rows])
    assert np.array_equal(_bits(entry[rows]), _bits(oentry[rows]))
    scene.close()


@pytest.mark.parametrize("kind", ["white", "spikes", "needles", "canyon"])
def test_hostile_content_full_frames_match_oracle(gpu, oracle, kind):
    """Maps built to defeat the exact-leap traversal (synth.CONTENT_KINDS: white noise, a 255-spike per 256^2 block,
    needles on a plateau, a canyon flown at low altitude) at BASELINE C2's size: full frame, per-ray step counts and
    distance() bits against the oracle, and the three kernel variants against each other.  Content changes the
    traversal's speed (profiles/r04_content.txt), never a pixel."""
    wl = gpu.synth.content_workload("C2", kind)
    rgb, cmap = wl.maps()
    params, cam = wl.scene_params(), wl.camera()
    scene = gpu.Scene(rgb, cmap, params)
    heights = oracle.update_heightmap(rgb, params)
    ofb, total, capped, osteps, oentry = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size), heights, cmap, per_pixel=True)
    fb, st, steps, entry = scene.render_stats(cam, per_pixel=True)
    assert capped == 0 and st.capped == 0 and st.steps == total and total > 0
    assert np.array_equal(fb, ofb) and np.array_equal(scene.render(cam), ofb)
    assert np.array_equal(steps.astype(np.int64), osteps) and np.array_equal(_bits(entry), _bits(oentry))
    for variant in ("group", "simple", "rec"):
        with kernel_variant(variant):
            assert np.array_equal(scene.render(cam), ofb), variant
    # the spherical headline camera over the same map (C3's pose scaled to this map), every 8th row
    cam3 = gpu.synth.content_workload("C3", kind).camera()
    s = float(wl.map_size)
    cam3.width, cam3.height = 960, 540
    cam3.pos[0], cam3.pos[1], cam3.pos[2] = -s / 8.0, s / 8.0, s * (1.0 / 32.0 if kind == "canyon" else 0.25)
    ofb3, total3, capped3, *_ = oracle.render(oracle.make_cfg(cam3, params, wl.map_size, wl.map_size), heights, cmap, row_stride=8)
    fb3, st3, *_ = scene.render_stats(cam3)
    rows = slice(0, cam3.height, 8)
    assert capped3 == 0 and np.array_equal(fb3[rows], ofb3[rows])
    scene.close()


def test_kernel_probe_never_changes_a_pixel(gpu, oracle):
    """The scene's one-time kernel probe (api.cpp launch_frame: the launch-order calibration of the first repeatedly
    rendered camera also times the plain-groups kernel and the scene keeps the faster one): every frame of the sequence
    that carries the probe -- before it, the measured launches of either kernel, after the verdict -- equals the oracle's,
    on a map where the groups win (needles under C5's camera) and on one where the leaps win; HMRM_TRY_GROUP=0 keeps
    the production kernel; a height update forgets the verdict."""
    for kind, expect in (("needles", None), ("smooth", 0)):
        wl = gpu.synth.content_workload("C2", kind) if kind != "smooth" else gpu.synth.WORKLOADS["C2"]
        rgb, cmap = wl.maps()
        params, cam = wl.scene_params(), wl.camera()
        heights = oracle.update_heightmap(rgb, params)
        ofb, *_ = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size), heights, cmap)
        scene = gpu.Scene(rgb, cmap, params)
        assert scene.kernel_choice() == 0
        for k in range(16):  # (first launch, 2 x up to 4 measured trials, settled launches)
            assert np.array_equal(scene.render(cam), ofb), (kind, k)
        choice = scene.kernel_choice()
        print(f"{kind}: the probe chose {KERNEL_NAMES[choice]}")
        assert choice in (0, 1, 3) and (expect is None or choice == expect)
        # another camera of the same scene renders with the scene's verdict: still the oracle's pixels
        cam2 = wl.camera(5, 64)
        ofb2, *_ = oracle.render(oracle.make_cfg(cam2, params, wl.map_size, wl.map_size), heights, cmap)
        assert np.array_equal(scene.render(cam2), ofb2)
        # strips too
        import torch
        strip = torch.zeros((cam.height, cam.width, 4), dtype=torch.uint8, device="cuda")
        scene.render_rows_device(cam, strip.data_ptr(), cam.width * 4, 100, 300, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(strip.cpu().numpy()[:200], ofb[100:300])
        scene.update(params)  # (same parameters: the heights are recomputed, the verdict is forgotten)
        assert scene.kernel_choice() == 0 and np.array_equal(scene.render(cam), ofb)
        scene.close()
        with env(HMRM_TRY_GROUP=0):
            scene = gpu.Scene(rgb, cmap, params)
            for k in range(12):
                assert np.array_equal(scene.render(cam), ofb)
            assert scene.kernel_choice() == 0
            scene.close()


def test_shadow_probe_for_cameras_that_never_repeat(gpu, oracle):
    """A moving camera is never calibrated, so the scene's kernel probe cannot ride on a calibration: the sixth full frame
    of such a scene is launched twice into the same buffer (production kernel, then plain groups; both measured) and a
    later launch reads the verdict.  Every frame of the sequence -- before, the doubled one, after -- is the oracle's; a
    ray stopped by the step cap in the doubled frame is counted once; on a map where rays cannot jump the scene may
    switch to the groups."""
    wl = gpu.synth.content_workload("C2", "needles")
    rgb, cmap = wl.maps()
    params = wl.scene_params()
    heights = oracle.update_heightmap(rgb, params)
    scene = gpu.Scene(rgb, cmap, params)
    for k in range(14):
        cam = wl.camera(k, 64)  # (14 different cameras of the orbit: nothing repeats)
        cam.width, cam.height = 640, 360
        ofb, *_ = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size), heights, cmap)
        assert np.array_equal(scene.render(cam), ofb), k
    print("needles under a moving camera: the shadow probe chose", KERNEL_NAMES[scene.kernel_choice()])
    assert scene.kernel_choice() in (0, 1, 3)
    scene.close()
    # capped rays in the doubled frame are reported once
    with env(HMRM_STEP_CAP=60):
        scene = gpu.Scene(rgb, cmap, params)
        for k in range(9):
            cam = wl.camera(k, 64)
            cam.width, cam.height = 640, 360
            ofb, total, capped, *_ = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size, step_cap=60), heights, cmap)
            fb, st, *_ = scene.render_stats(cam, allow_capped=True)  # (the instrumented count of this frame)
            assert st.capped == capped and np.array_equal(fb, ofb)
            try:
                got = scene.render(cam)
                n_capped = 0
            except gpu.HmrmError as e:
                assert e.code == gpu.HMRM_E_NOTERM
                n_capped = int(e.message.split()[0])
                got = None
            assert n_capped == capped, (k, n_capped, capped)
        scene.close()


def test_no_probe_flag_keeps_a_frame_from_being_launched_twice(gpu, oracle):
    """HMRM_NO_PROBE (hmrm_render_device_begin_flags): a caller that counts on every frame's latency keeps the scene's
    shadow probe -- one frame launched twice, production kernel and plain groups -- off its frames.  On a map where the probe, once
    it runs, picks the plain groups (needles) the kernel choice stays "production" through twenty never-repeating flagged frames
    and changes only after unflagged ones; every frame is the oracle's either way."""
    import torch
    wl = gpu.synth.content_workload("C2", "needles")
    rgb, cmap = wl.maps()
    params = wl.scene_params()
    heights = oracle.update_heightmap(rgb, params)
    scene = gpu.Scene(rgb, cmap, params)
    out = torch.zeros((360, 640, 4), dtype=torch.uint8, device="cuda")

    def frame(k, no_probe):
        cam = wl.camera(k, 97)
        cam.width, cam.height = 640, 360
        t = scene.render_device_begin(cam, out.data_ptr(), 640 * 4, no_probe=no_probe)
        scene.render_device_wait(t)
        return cam
    for k in range(20):
        cam = frame(k, True)
        assert scene.kernel_choice() == 0, k
    ofb, *_ = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size), heights, cmap)
    assert np.array_equal(out.cpu().numpy(), ofb)
    for k in range(20, 40):
        cam = frame(k, False)
    ofb, *_ = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size), heights, cmap)
    assert np.array_equal(out.cpu().numpy(), ofb)
    print("needles, unflagged frames: the shadow probe chose", KERNEL_NAMES[scene.kernel_choice()])
    scene.close()


def test_step_cap_is_reported_not_silent(gpu, oracle):
    """A vertical upward ray over a non-hitting cell never leaves the reference's while(true)
    (hmap.cpp:1000-1038).  The kernel stops at the cap, shades a miss and says so."""
    rgb = np.zeros((8, 8, 3), dtype=np.uint8)
    cmap = np.full((8, 8, 4), 255, dtype=np.uint8)
    params = gpu.SceneParams.make(0.0, 4.0, grid_width=1.0)
    # orthographic straight UP from below the box: dir = (0,0,1) after the float round trip
    cam = gpu.Camera.make(width=4, height=4, projection=3, hang=0.0, vang=0.0, pos=(4.0, -4.0, -3.0),
                          ortho_width=0.5, step_dist=0.0, bg=(9, 8, 7))
    os.environ["HMRM_STEP_CAP"] = "1000"
    try:
        scene = gpu.Scene(rgb, cmap, params)
        fb, st, steps, _ = scene.render_stats(cam, per_pixel=True, allow_capped=True)
        assert st.capped == 16 and (steps == 1000).all()
        with pytest.raises(gpu.HmrmError) as e:
            scene.render(cam)
        assert e.value.code == gpu.HMRM_E_NOTERM
        heights = oracle.update_heightmap(rgb, params)
        ofb, total, capped, *_ = oracle.render(oracle.make_cfg(cam, params, 8, 8, step_cap=1000), heights, cmap)
        assert capped == 16 and np.array_equal(fb, ofb)
        scene.close()
    finally:
        del os.environ["HMRM_STEP_CAP"]


def test_cli_end_to_end(gpu, oracle, tmp_path):
    """hmap <config>: reference-format config file -> PNG and PPM with the oracle's pixels."""
    wl = gpu.synth.WORKLOADS["C1"]
    rgb, cmap = gpu.synth.synth_maps(wl.map_size)
    hp, cp = str(tmp_path / "h.ppm"), str(tmp_path / "c.png")
    gpu.write_ppm(hp, rgb)
    gpu.write_png(cp, cmap)
    params, cam = wl.scene_params(), wl.camera()
    heights = oracle.update_heightmap(rgb, params)
    ofb, *_ = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size), heights, cmap)
    exe = os.path.join(os.path.dirname(gpu.LIB_PATH), "hmap")
    for ext in ("png", "ppm"):
        outp = str(tmp_path / f"frame.{ext}")
        cfgp = tmp_path / f"c_{ext}.txt"
        cfgp.write_text(gpu.synth.config_text(wl, hp, cp, outp))
        r = subprocess.run([exe, str(cfgp)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert f"Saved screenshot at {outp}" in r.stdout and r.stdout.startswith("resolution 320 240\n")
        img, n = gpu.image_load(outp, 4)
        assert np.array_equal(img, ofb)
        if ext == "png":
            assert open(outp, "rb").read() == gpu.png_encode(ofb)


def test_cli_devices_key(gpu, oracle, tmp_path):
    """`devices 3` in the CLI: one frame as three scenes' bands (hmrm_render_multi) and a recording with frame
    k on scene k mod 3 -- on this box all three on the one GPU (HMRM_OVERSUBSCRIBE_DEVICES=1)."""
    wl = gpu.synth.WORKLOADS["C1"]
    rgb, cmap = gpu.synth.synth_maps(wl.map_size)
    hp, cp = str(tmp_path / "h.ppm"), str(tmp_path / "c.png")
    gpu.write_ppm(hp, rgb)
    gpu.write_png(cp, cmap)
    params, cam = wl.scene_params(), wl.camera()
    heights = oracle.update_heightmap(rgb, params)
    ofb, *_ = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size), heights, cmap)
    exe = os.path.join(os.path.dirname(gpu.LIB_PATH), "hmap")
    envp = dict(os.environ, HMRM_OVERSUBSCRIBE_DEVICES="1")
    outp = str(tmp_path / "frame.png")
    cfgp = tmp_path / "one.txt"
    cfgp.write_text(gpu.synth.config_text(wl, hp, cp, outp) + "devices 3\n")
    r = subprocess.run([exe, str(cfgp)], capture_output=True, text=True, timeout=300, env=envp)
    assert r.returncode == 0 and "on 3 devices" in r.stdout and "devices 3" in r.stdout, r.stdout + r.stderr
    assert open(outp, "rb").read() == gpu.png_encode(ofb)
    rec = tmp_path / "rec"
    cfg2 = tmp_path / "rec.txt"
    cfg2.write_text(gpu.synth.config_text(wl, hp, cp, str(rec)) + "devices 3 record orbit recording_frame_count 7\n")
    r = subprocess.run([exe, str(cfg2)], capture_output=True, text=True, timeout=300, env=envp)
    assert r.returncode == 0 and r.stdout.count("Saved screenshot at ") == 7, r.stdout + r.stderr
    files = sorted(rec.iterdir())
    assert len(files) == 7
    # frame 0 of the orbit through the configured pose is that pose
    f0 = [p for p in files if p.name.endswith("_0.png")][0]
    img, _ = gpu.image_load(str(f0), 4)
    assert img.shape == ofb.shape


def test_recording_orbit_writes_reference_named_pngs(gpu, oracle, tmp_path):
    """`record orbit`: frames <dir>/hmap_<id>_<n>.png (hmap.cpp:1132-1134), each equal to the oracle's frame."""
    rgb, cmap = scenes.small_maps(64, 64, 41)
    params = gpu.SceneParams.make(0.0, 8.0, grid_width=1.0)
    base = gpu.Camera.make(width=80, height=45, projection=1, hfov=gpu.degrees_to_rads(80), hang=0.0,
                           vang=gpu.degrees_to_rads(112), pos=(-20.0, 20.0, 30.0), step_dist=0.5, bg=(4, 5, 6))
    scene = gpu.Scene(rgb, cmap, params)
    heights = oracle.update_heightmap(rgb, params)
    frames, cx, cy, radius, hang0 = 5, 32.0, -32.0, 70.0, gpu.degrees_to_rads(-45.0)
    out = tmp_path / "rec"
    out.mkdir()
    gpu.record_orbit(scene, base, cx, cy, radius, hang0, frames, str(out), 1234, encoder_threads=3)
    assert sorted(p.name for p in out.iterdir()) == [f"hmap_1234_{k}.png" for k in range(frames)]
    for k in range(frames):
        cam = gpu.orbit_camera(base, cx, cy, radius, hang0, k, frames)
        ofb, *_ = oracle.render(oracle.make_cfg(cam, params, 64, 64), heights, cmap)
        img, n = gpu.image_load(str(out / f"hmap_1234_{k}.png"), 4)
        assert n == 4 and np.array_equal(img, ofb), k
        assert (out / f"hmap_1234_{k}.png").read_bytes() == gpu.png_encode(ofb)
    scene.close()
    # CLI: `record orbit` + recording_frame_count
    hp, cp = str(tmp_path / "h.ppm"), str(tmp_path / "c.png")
    gpu.write_ppm(hp, rgb)
    gpu.write_png(cp, cmap)
    cfgp = tmp_path / "rec.txt"
    cfgp.write_text(f"resolution 40 30 pos -20 20 30 vang 112 max_height 8 grid_width 1 step_dist 0.5 cycle 1\n"
                    f"heightmap {hp}\ncolormap {cp}\nrecord orbit recording_frame_count 3 output {tmp_path / 'cli'}\n")
    exe = os.path.join(os.path.dirname(gpu.LIB_PATH), "hmap")
    r = subprocess.run([exe, str(cfgp)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.rstrip().endswith("Done recording.") and r.stdout.count("Saved screenshot at ") == 3
    assert len(list((tmp_path / "cli").iterdir())) == 3


def test_c4_orthographic_8192_subsampled(gpu, oracle):
    """BASELINE config C4 on one GPU: 7680x4320 orthographic over an 8192^2 map (every 96th row vs oracle)."""
    wl = gpu.synth.WORKLOADS["C4"]
    rgb, cmap = gpu.synth.synth_maps(wl.map_size)
    params, cam = wl.scene_params(), wl.camera()
    scene = gpu.Scene(rgb, cmap, params)
    fb, st, steps, entry = scene.render_stats(cam, per_pixel=True)
    assert np.array_equal(scene.render(cam), fb)
    assert int(steps.astype(np.int64).sum()) == st.steps and st.capped == 0
    heights = oracle.update_heightmap(rgb, params)
    stride = 96
    ofb, total, capped, osteps, oentry = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size),
                                                       heights, cmap, per_pixel=True, row_stride=stride)
    rows = slice(0, cam.height, stride)
    assert capped == 0 and np.array_equal(fb[rows], ofb[rows])
    assert np.array_equal(steps[rows].astype(np.int64), osteps[rows])
    assert np.array_equal(_bits(entry[rows]), _bits(oentry[rows]))
    scene.close()


def test_exact_tie_steps_and_many_binades(gpu, oracle):
    """Orthographic rays along +x with step_dist = 0.25 + 2^-43: in the binade [1024, 2048)
    (ulp 2^-42) every addition x + s is an exact rounding tie, in [512, 1024) it is exact, in
    [2048, 4096) it rounds normally -- the leap arithmetic must reproduce the sequential sums
    across all of them (and across ~12 smaller binades near x = 0)."""
    rng = np.random.RandomState(9)
    mw, mh = 4096, 16
    rgb = np.repeat(rng.randint(0, 40, size=(mh, mw, 1)).astype(np.uint8), 3, axis=2)
    rgb[:, 3900:3910] = 255  # a wall near the far end stops every ray
    cmap = rng.randint(0, 256, size=(mh, mw, 4)).astype(np.uint8)
    cmap[:, :, 3] = 255
    params = gpu.SceneParams.make(0.0, 8.0, grid_width=1.0)
    sd = 0.25 + 2.0 ** -43
    cam = gpu.Camera.make(width=16, height=12, projection=3, hang=0.0, vang=gpu.degrees_to_rads(90),
                          pos=(-3.0, -8.0, 4.0), ortho_width=0.45, step_dist=sd, bg=(0, 0, 0))
    scene = gpu.Scene(rgb, cmap, params)
    heights = oracle.update_heightmap(rgb, params)
    ofb, total, capped, osteps, oentry = oracle.render(oracle.make_cfg(cam, params, mw, mh), heights, cmap, per_pixel=True)
    assert capped == 0 and osteps.max() > 10000
    for variant in KERNEL_VARIANTS:
        with kernel_variant(variant):
            fb, st, steps, entry = scene.render_stats(cam, per_pixel=True)
        assert np.array_equal(fb, ofb) and np.array_equal(steps.astype(np.int64), osteps), variant
        if variant == "leap":
            assert st.leaped_steps > 0.9 * st.steps, "the leap path was not exercised"
    scene.close()


def test_jumps_cross_binade_boundaries_themselves(gpu, oracle, monkeypatch):
    """csrc/render_fast.hip (HMRM_CROSS): a jump ends with one real step, so a jump that a binade's end cut short
    carries its coordinate into the next binade itself.  The instrumented kernel counts what came before each group of
    real steps (diagnostic mode 20): none may follow such a jump any more -- before, half of all groups did.  Pixels and
    step counts against the oracle as everywhere; a general grid width, whose binade boundaries fall inside windows."""
    rng = np.random.RandomState(21)
    mw = mh = 512
    rgb = np.repeat(rng.randint(0, 30, size=(mh, mw, 1)).astype(np.uint8), 3, axis=2)
    cmap = rng.randint(0, 256, size=(mh, mw, 4)).astype(np.uint8)
    cmap[:, :, 3] = 255
    for gw in (1.0, 0.3):
        params = gpu.SceneParams.make(0.0, 12.0 * gw, grid_width=gw)
        cam = gpu.Camera.make(width=96, height=64, projection=1, hfov=gpu.degrees_to_rads(70), hang=gpu.degrees_to_rads(40),
                              vang=gpu.degrees_to_rads(100), pos=(30.0 * gw, -40.0 * gw, 60.0 * gw), step_dist=0.11 * gw, bg=(1, 2, 3))
        scene = gpu.Scene(rgb, cmap, params)
        heights = oracle.update_heightmap(rgb, params)
        ofb, total, capped, osteps, _ = oracle.render(oracle.make_cfg(cam, params, mw, mh), heights, cmap, per_pixel=True)
        fb, st, steps, _ = scene.render_stats(cam, per_pixel=True)
        assert capped == 0 and np.array_equal(fb, ofb) and np.array_equal(steps.astype(np.int64), osteps)
        assert st.leaps > 1000 and st.leaped_steps > 0.8 * st.steps
        monkeypatch.setenv("HMRM_DIAG_ITERS", "20")
        cam.bg_r = 7  # (another frame record: the diagnostic mode is not part of the cache key)
        _, d, *_ = scene.render_stats(cam)
        monkeypatch.delenv("HMRM_DIAG_ITERS")
        after_binade_jump, no_binade_room, paused, other = d.leap_attempts, d.leaps, d.groups, d.leaped_steps
        assert after_binade_jump == 0, (gw, after_binade_jump, no_binade_room, paused, other)
        assert paused + other > 0
        scene.close()


@pytest.mark.parametrize("gw,sd", [(0.3, 0.15), (0.05, 0.02), (3.0, 0.7)])
def test_general_grid_width_full_frame(gpu, oracle, gw, sd):
    """grid_width that is not a power of two: cell = trunc(x * fl(1/gw)) with the true division only
    near integer boundaries must equal trunc(x / gw) everywhere (1024^2 map, 960x540, all rays)."""
    wl = gpu.synth.WORKLOADS["C2"]
    rgb, cmap = gpu.synth.synth_maps(wl.map_size)
    s = wl.map_size * gw
    params = gpu.SceneParams.make(0.0, s / 16.0, grid_width=gw)
    cam = gpu.Camera.make(width=960, height=540, projection=1, hfov=gpu.degrees_to_rads(90),
                          hang=gpu.degrees_to_rads(-45), vang=gpu.degrees_to_rads(115),
                          pos=(-s / 8.0, s / 8.0, s / 4.0), step_dist=sd, bg=(0, 0, 0))
    scene = gpu.Scene(rgb, cmap, params)
    heights = oracle.update_heightmap(rgb, params)
    ofb, total, capped, osteps, _ = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size), heights, cmap,
                                                   per_pixel=True)
    fb, st, steps, _ = scene.render_stats(cam, per_pixel=True)
    assert capped == 0 and np.array_equal(fb, ofb) and np.array_equal(steps.astype(np.int64), osteps)
    assert st.steps == total and st.leaped_steps > 0
    scene.close()


def test_grid_width_with_integer_reciprocal_at_binade_boundaries(gpu, oracle):
    """grid_width 0.05 (the reference's default, hmap.cpp:65) and 0.01 (its sample config): fl(1 / gw) is the integer 20 /
    100, so every power of two of the world coordinates is also a CELL boundary, and the step of a jump that carries a
    coordinate into its next binade lands right behind one.  Rays whose crossing step ends within 2^-20 cells of it (a
    handful per 4K frame; round 4 found pixels (1758, 737) and (2263, 819) of the C3 camera at gw 0.05 that way) have a
    landing cell the reciprocal cannot name: the kernel accepts such a jump only if both candidate cells lie inside the
    window, and must still produce the reference's pixels and step counts.  Rows holding those rays, the orthographic
    camera looking along an axis (coordinates that never move), and the time such a frame takes against the same
    cells at grid_width 0.07 (the refusals marched 175 groups where one jump does: 1.35 x the frame time)."""
    rgb, cmap = gpu.synth.synth_maps(4096)
    times = {}
    for gw in (0.05, 0.01, 0.07):
        wl = gpu.synth.grid_workload("C3", gw)
        params, cam = wl.scene_params(), wl.camera()
        scene = gpu.Scene(rgb, cmap, params)
        fb, st, steps, entry = scene.render_stats(cam, per_pixel=True)
        assert np.array_equal(scene.render(cam), fb)
        heights = oracle.update_heightmap(rgb, params)
        cfg = oracle.make_cfg(cam, params, 4096, 4096)
        for r in (737, 819, 880, 881, 964, 1090, 1216, 1500):
            ofb, total, capped, osteps, oentry = oracle.render(cfg, heights, cmap, per_pixel=True, rows=(r, r + 1))
            assert capped == 0 and np.array_equal(fb[r], ofb[r]), (gw, r)
            assert np.array_equal(steps[r].astype(np.int64), osteps[r]) and np.array_equal(_bits(entry[r]), _bits(oentry[r])), (gw, r)
        for _ in range(12):
            scene.bench_kernel_ms(cam, 1)
        times[gw] = min(scene.bench_kernel_ms(cam, 10) for _ in range(3))
        scene.close()
    print("kernel ms by grid width:", {k: round(v, 4) for k, v in times.items()})
    assert times[0.05] < 1.15 * times[0.07] and times[0.01] < 1.15 * times[0.07], times
    # coordinates that never move: orthographic rays along -y / +x over a 0.05 grid, columns a quarter cell apart from a
    # corner that is a multiple of the cell: every 4th column of rays runs ALONG a cell boundary (to the last bits)
    rgb2, cmap2 = scenes.small_maps(96, 96, 53)
    params = gpu.SceneParams.make(0.0, 0.8, grid_width=0.05)
    heights = oracle.update_heightmap(rgb2, params)
    scene = gpu.Scene(rgb2, cmap2, params)
    for hang_deg in (-90.0, 0.0):
        cam = gpu.Camera.make(width=193, height=97, projection=3, hang=gpu.degrees_to_rads(hang_deg), vang=gpu.degrees_to_rads(120),
                              pos=(2.4 if hang_deg else -1.0, 1.0 if hang_deg else -2.4, 2.0), ortho_width=0.0125 * 192 / 193, step_dist=0.0125, bg=(3, 4, 5))
        ofb, total, capped, osteps, _ = oracle.render(oracle.make_cfg(cam, params, 96, 96), heights, cmap2, per_pixel=True)
        fb, st, steps, _ = scene.render_stats(cam, per_pixel=True)
        assert capped == 0 and np.array_equal(fb, ofb) and np.array_equal(steps.astype(np.int64), osteps), hang_deg
        assert np.array_equal(scene.render(cam), ofb)
    scene.close()


def test_degenerate_parameters_agree_with_oracle(gpu, oracle):
    """Negative / huge step_dist, camera below the map, inverted height range: nothing special-cased
    in the reference, so whatever its loop does the kernels must do too (within the step cap)."""
    rgb, cmap = scenes.small_maps(48, 48, 91)
    base = dict(width=40, height=30, projection=1, hfov=gpu.degrees_to_rads(85), hang=gpu.degrees_to_rads(-45),
                vang=gpu.degrees_to_rads(118), pos=(-8.0, 8.0, 20.0), bg=(1, 2, 3))
    trials = [
        (gpu.SceneParams.make(0.0, 6.0, grid_width=1.0), dict(step_dist=-0.5)),
        (gpu.SceneParams.make(0.0, 6.0, grid_width=1.0), dict(step_dist=1e6)),
        (gpu.SceneParams.make(0.0, 6.0, grid_width=1.0), dict(step_dist=1e-3)),
        (gpu.SceneParams.make(6.0, 0.0, grid_width=1.0), dict(step_dist=0.5)),           # max < min
        (gpu.SceneParams.make(0.0, 6.0, grid_width=1.0), dict(step_dist=0.5, pos=(10.0, -10.0, -5.0), vang=gpu.degrees_to_rads(60))),
        (gpu.SceneParams.make(0.0, 6.0, grid_width=-1.0), dict(step_dist=0.5)),          # box folds over
    ]
    os.environ["HMRM_STEP_CAP"] = "200000"
    try:
        for params, over in trials:
            kw = dict(base)
            kw.update(over)
            cam = gpu.Camera.make(**kw)
            scene = gpu.Scene(rgb, cmap, params)
            heights = oracle.update_heightmap(rgb, params)
            ofb, total, capped, osteps, _ = oracle.render(oracle.make_cfg(cam, params, 48, 48, step_cap=200000),
                                                           heights, cmap, per_pixel=True)
            for variant in KERNEL_VARIANTS:
                with kernel_variant(variant):
                    fb, st, steps, _ = scene.render_stats(cam, per_pixel=True, allow_capped=True)
                assert np.array_equal(fb, ofb), (over, variant)
                assert st.capped == capped and st.steps == total, (over, variant)
            scene.close()
    finally:
        del os.environ["HMRM_STEP_CAP"]


def test_random_cameras_and_parameters_fuzz(gpu, oracle):
    """Seeded fuzz: random projection / pose / fov / step / grid width / height range on random
    maps, production kernel vs oracle (frame + per-ray step counts).  The leap path has many
    data-dependent corners (window edges, binade crossings, z-limited jumps); this sweeps them."""
    rng = np.random.RandomState(20261003)
    os.environ["HMRM_STEP_CAP"] = "400000"
    checked = leaped = 0
    try:
        for trial in range(48):
            mw, mh = int(rng.choice([48, 96, 160, 257])), int(rng.choice([48, 96, 131]))
            rgb, cmap = scenes.small_maps(mw, mh, 1000 + trial, color_heights=bool(trial % 3 == 0))
            gw = float(rng.choice([1.0, 0.5, 0.05, 0.3, 2.0, 1.7]))
            lo = float(rng.choice([0.0, 0.0, -1.5, 2.0]))
            hi = lo + float(rng.uniform(0.5, 0.3 * mw)) * gw
            params = gpu.SceneParams.make(lo, hi, grid_width=gw)
            ext_x, ext_y = mw * gw, mh * gw
            proj = int(rng.choice([1, 2, 3]))
            # camera somewhere around / above the map, looking roughly at its centre
            ang = rng.uniform(0, 2 * np.pi)
            dist = rng.uniform(0.2, 1.6) * max(ext_x, ext_y)
            pos = (ext_x / 2 + dist * np.cos(ang), -ext_y / 2 + dist * np.sin(ang), hi + rng.uniform(-0.5, 2.0) * (hi - lo + gw))
            hang = float(np.arctan2(-ext_y / 2 - pos[1], ext_x / 2 - pos[0]) + rng.uniform(-0.4, 0.4))
            vang = float(gpu.degrees_to_rads(rng.uniform(60, 170)))
            cam = gpu.Camera.make(width=int(rng.randint(17, 90)), height=int(rng.randint(9, 70)), projection=proj,
                                  hfov=float(gpu.degrees_to_rads(rng.uniform(30, 175))), hang=hang, vang=vang, pos=pos,
                                  ortho_width=float(rng.uniform(0.2, 3.0) * gw),
                                  step_dist=float(rng.choice([0.05, 0.1, 0.25, 0.5, 1.0, 0.37]) * gw),
                                  bg=tuple(int(v) for v in rng.randint(0, 256, size=3)))
            heights = oracle.update_heightmap(rgb, params)
            ofb, total, capped, osteps, oentry = oracle.render(oracle.make_cfg(cam, params, mw, mh, step_cap=400000),
                                                               heights, cmap, per_pixel=True)
            scene = gpu.Scene(rgb, cmap, params)
            fb, st, steps, entry = scene.render_stats(cam, per_pixel=True, allow_capped=True)
            label = (trial, proj, gw, cam.step_dist)
            assert np.array_equal(_bits(entry), _bits(oentry)), label
            assert np.array_equal(fb, ofb), label
            if capped == 0:
                assert np.array_equal(steps.astype(np.int64), osteps) and st.steps == total, label
            assert st.capped == capped, label
            try:
                plain = scene.render(cam)
            except gpu.HmrmError as e:
                assert e.code == gpu.HMRM_E_NOTERM and capped > 0
            else:
                assert np.array_equal(plain, ofb), label
            leaped += st.leaped_steps
            checked += 1
            scene.close()
    finally:
        del os.environ["HMRM_STEP_CAP"]
    assert checked == 48 and leaped > 0


def test_window_records_are_the_ninth_highest_and_the_cells_above_it(gpu):
    """k_build_records against numpy on the device's own threshold table: every 16 x 16 window (one every 4 cells, clipped
    at the map's edge) holds max2 = its ninth-highest threshold rounded UP to float (-inf with fewer than nine cells) and
    exactly the cells strictly above that value -- at most eight by construction; ties at the ninth value are not recorded."""
    rng = np.random.RandomState(7)
    for mw, mh, kind in ((64, 48, "needles"), (37, 21, "noise"), (130, 9, "ties"), (3, 70, "noise"), (1, 1, "flat"), (257, 33, "rolling")):
        if kind == "needles":
            v = np.where(rng.rand(mh, mw) < 1 / 20, 255, 40)
        elif kind == "noise":
            v = rng.randint(0, 256, size=(mh, mw))
        elif kind == "ties":
            v = np.where(rng.rand(mh, mw) < 1 / 6, 200, np.where(rng.rand(mh, mw) < 0.5, 90, 30))
        elif kind == "flat":
            v = np.full((mh, mw), 77)
        else:
            yy, xx = np.mgrid[0:mh, 0:mw]
            v = (100 + 60 * np.sin(xx / 11.0) * np.cos(yy / 5.0)).astype(np.int64)
        v8 = v.astype(np.uint8)
        rgb = np.ascontiguousarray(np.repeat(v8[:, :, None], 3, axis=2))
        scene = gpu.Scene(rgb, np.zeros((mh, mw, 4), dtype=np.uint8), gpu.SceneParams.make(-1.5, 23.0, grid_width=0.7))
        recs, thr = scene.read_records()
        scene.close()
        assert recs.shape == ((mh + 3) // 4, (mw + 3) // 4)
        for iy in range(recs.shape[0]):
            for ix in range(recs.shape[1]):
                win = thr[iy * 4:iy * 4 + 16, ix * 4:ix * 4 + 16]
                order = np.sort(win.ravel())[::-1]
                ninth = order[8] if order.size > 8 else -np.inf
                want2 = np.float32(ninth)
                if float(want2) < ninth:
                    want2 = np.nextafter(want2, np.float32(np.inf))
                r = recs[iy, ix]
                assert r["max2"] == want2 or (np.isneginf(want2) and np.isneginf(r["max2"])), (kind, ix, iy)
                got = {(int(x), int(y)) for x, y in zip(r["xs"], r["ys"]) if x != 255 or y != 255}
                ys, xs = np.nonzero(win > ninth)
                assert got == {(int(x), int(y)) for x, y in zip(xs, ys)}, (kind, ix, iy)
                assert all((x == 255) == (y == 255) for x, y in zip(r["xs"], r["ys"]))


def test_window_records_follow_a_height_update(gpu, oracle):
    """The records are built when the record kernel is first about to run after a height update (api.cpp ensure_records), not
    by the update: frames before and after hmrm_scene_update -- other height range, other luminance weights -- equal the
    oracle's under HMRM_KERNEL=rec, and the table read back afterwards describes the NEW thresholds."""
    rng = np.random.RandomState(11)
    mw, mh = 96, 80
    v8 = np.where(rng.rand(mh, mw) < 1 / 40, 250, 30 + (rng.rand(mh, mw) * 8).astype(np.int64)).astype(np.uint8)
    rgb = np.ascontiguousarray(np.stack([v8, np.roll(v8, 3, axis=1), np.roll(v8, 5, axis=0)], axis=2))
    cmap = rng.randint(0, 256, size=(mh, mw, 4)).astype(np.uint8)
    cam = gpu.Camera.make(width=120, height=68, projection=2, hfov=gpu.degrees_to_rads(150), hang=gpu.degrees_to_rads(-40.0),
                          vang=gpu.degrees_to_rads(100), pos=(-10.0, 12.0, 9.0), step_dist=0.25, bg=(9, 8, 7))
    p1 = gpu.SceneParams.make(0.0, 14.0, grid_width=1.0, lum=(1.0, 0.0, 0.0))
    p2 = gpu.SceneParams.make(-2.0, 9.0, grid_width=1.0, lum=(0.0, 0.5, 0.5))
    with kernel_variant("rec"):
        scene = gpu.Scene(rgb, cmap, p1)
        for params in (p1, p2, p1):
            scene.update(params)
            heights = oracle.update_heightmap(rgb, params)
            ofb, total, capped, osteps, _ = oracle.render(oracle.make_cfg(cam, params, mw, mh), heights, cmap, per_pixel=True)
            fb, st, steps, _ = scene.render_stats(cam, per_pixel=True)
            assert capped == 0 and np.array_equal(fb, ofb) and np.array_equal(steps.astype(np.int64), osteps), params.max_height
            assert np.array_equal(scene.render(cam), ofb) and st.leaps > 0
            recs, thr = scene.read_records()
            win = thr[0:16, 0:16]
            ninth = np.sort(win.ravel())[::-1][8]
            assert recs[0, 0]["max2"] >= ninth and recs[0, 0]["max2"] <= np.nextafter(np.float32(ninth), np.float32(np.inf))
        scene.close()


def test_record_kernel_on_sparse_tall_cells_fuzz(gpu, oracle):
    """The record kernel (HMRM_KERNEL=rec, frame.hpp WindowRecord: a 16-cell window's maximum without its 8 highest
    cells, and where those stand) on the maps it is for -- flat or gently rolling ground with tall single cells at a
    density of 1/8 .. 1/300, map sizes that clip the last windows, every projection, grid widths of all three kinds
    -- frame, per-ray step counts and distance() bits against the oracle.  A path that grazes a recorded cell must be
    refused, one that misses it by a hair must land exactly where the marched ray stands; a ray that leaves through the
    map's low edge passes coordinates in (-1, 0), which the reference's truncation still counts as cell 0."""
    rng = np.random.RandomState(20261005)
    os.environ["HMRM_STEP_CAP"] = "400000"
    checked = leaped = 0
    try:
        with kernel_variant("rec"):
            for trial in range(60):
                mw, mh = int(rng.choice([1, 3, 37, 64, 130, 257])), int(rng.choice([2, 33, 64, 131]))
                ground = rng.randint(20, 60)
                v = np.full((mh, mw), ground, dtype=np.uint8)
                if trial % 3 == 1:  # rolling ground
                    yy, xx = np.mgrid[0:mh, 0:mw]
                    v = (ground + 12 * np.sin(xx / 9.0) * np.cos(yy / 7.0)).astype(np.uint8)
                dens = float(rng.choice([1 / 8, 1 / 30, 1 / 64, 1 / 300]))
                tall = rng.rand(mh, mw) < dens
                v[tall] = rng.randint(180, 256, size=int(tall.sum())).astype(np.uint8) if trial % 2 else 255
                rgb = np.ascontiguousarray(np.repeat(v[:, :, None], 3, axis=2))
                cmap = rng.randint(0, 256, size=(mh, mw, 4)).astype(np.uint8)
                gw = float(rng.choice([1.0, 1.0, 0.5, 0.05, 0.3, 1.7]))
                lo = float(rng.choice([0.0, -1.5, 2.0]))
                hi = lo + float(rng.uniform(2.0, 0.25 * max(mw, mh, 16))) * gw
                params = gpu.SceneParams.make(lo, hi, grid_width=gw)
                ext_x, ext_y = mw * gw, mh * gw
                proj = int(rng.choice([1, 2, 3]))
                ang = rng.uniform(0, 2 * np.pi)
                dist = rng.uniform(0.0, 1.2) * max(ext_x, ext_y)
                # cameras between the ground and the tops of the tall cells as often as above them
                zc = lo + (hi - lo) * float(rng.uniform(0.15, 1.4))
                pos = (ext_x / 2 + dist * np.cos(ang), -ext_y / 2 + dist * np.sin(ang), 2 * lo + zc)
                hang = float(np.arctan2(-ext_y / 2 - pos[1], ext_x / 2 - pos[0]) + rng.uniform(-0.5, 0.5))
                cam = gpu.Camera.make(width=int(rng.randint(24, 100)), height=int(rng.randint(16, 72)), projection=proj,
                                      hfov=float(gpu.degrees_to_rads(rng.uniform(40, 175))), hang=hang,
                                      vang=float(gpu.degrees_to_rads(rng.uniform(75, 150))), pos=pos,
                                      ortho_width=float(rng.uniform(0.2, 3.0) * gw),
                                      step_dist=float(rng.choice([0.05, 0.1, 0.25, 0.5, 1.0, 0.37]) * gw),
                                      bg=tuple(int(b) for b in rng.randint(0, 256, size=3)))
                heights = oracle.update_heightmap(rgb, params)
                ofb, total, capped, osteps, oentry = oracle.render(oracle.make_cfg(cam, params, mw, mh, step_cap=400000),
                                                                   heights, cmap, per_pixel=True)
                scene = gpu.Scene(rgb, cmap, params)
                fb, st, steps, entry = scene.render_stats(cam, per_pixel=True, allow_capped=True)
                label = (trial, proj, gw, cam.step_dist, dens)
                assert np.array_equal(_bits(entry), _bits(oentry)), label
                assert np.array_equal(fb, ofb), label
                if capped == 0:
                    assert np.array_equal(steps.astype(np.int64), osteps) and st.steps == total, label
                    assert np.array_equal(scene.render(cam), ofb), label
                assert st.capped == capped, label
                leaped += st.leaped_steps
                checked += 1
                scene.close()
    finally:
        del os.environ["HMRM_STEP_CAP"]
    assert checked == 60 and leaped > 0


def test_async_ring_frames_in_flight(gpu, oracle):
    """hmrm_render_begin/_wait/_release: several frames in flight (kernel k+1 beside the copy of frame
    k), each equal to the oracle's frame for its camera, slots reusable after release."""
    rgb, cmap = scenes.small_maps(96, 96, 21)
    params = gpu.SceneParams.make(0.0, 9.0, grid_width=1.0)
    scene = gpu.Scene(rgb, cmap, params)
    heights = oracle.update_heightmap(rgb, params)
    base = gpu.Camera.make(width=160, height=90, projection=2, hfov=gpu.degrees_to_rads(150), hang=0.0,
                           vang=gpu.degrees_to_rads(112), pos=(-30.0, 30.0, 40.0), step_dist=0.25, bg=(4, 5, 6))
    cams = [gpu.orbit_camera(base, 48.0, -48.0, 90.0, gpu.degrees_to_rads(-45.0), k, 7) for k in range(7)]
    want = [oracle.render(oracle.make_cfg(c, params, 96, 96), heights, cmap)[0] for c in cams]
    for lap in range(2):
        tickets = [scene.render_begin(c) for c in cams[:4]]           # four in flight
        assert len(set(tickets)) == 4
        for k in (0, 1):
            assert np.array_equal(scene.render_wait(tickets[k], (90, 160)), want[k])
            scene.render_release(tickets[k])
        tickets += [scene.render_begin(c) for c in cams[4:]]           # reuses the released slots
        for k in range(2, 7):
            view = scene.render_wait(tickets[k], (90, 160), copy=False)
            assert np.array_equal(view, want[k])
            scene.render_release(tickets[k])
    with pytest.raises(gpu.HmrmError):
        scene.render_wait(tickets[0], (90, 160))                       # released: no such frame in flight
    scene.close()


def test_device_tickets_and_launch_lanes(gpu, oracle):
    """hmrm_render_device_begin / _wait: frames into the caller's device memory through the scene's three launch lanes.
    Several cameras in flight at once (each lane has its own tables and counters), every frame equal to the oracle's;
    a capped ray is reported by the wait of its lane; and -- the point of the lanes, VERDICT r03 #8 -- a sequence of
    small frames (BASELINE C2: 1080p over 1024^2) with three tickets in flight takes less GPU time per frame than the
    same frames back to back on one stream, without the caller managing a stream (profiles/r04_lanes.txt)."""
    import time
    import torch
    rgb, cmap = scenes.small_maps(96, 96, 41)
    params = gpu.SceneParams.make(0.0, 10.0, grid_width=1.0)
    heights = oracle.update_heightmap(rgb, params)
    scene = gpu.Scene(rgb, cmap, params)
    cams = [gpu.Camera.make(width=150 + 7 * k, height=90 + 3 * k, projection=1 + k % 3, hfov=gpu.degrees_to_rads(140 if k % 3 == 1 else 80),
                            hang=gpu.degrees_to_rads(-45 + 5 * k), vang=gpu.degrees_to_rads(112), pos=(-30.0, 30.0, 40.0),
                            ortho_width=0.8, step_dist=0.25, bg=(k, 2, 3)) for k in range(7)]
    bufs = [torch.zeros((c.height, c.width + 5, 4), dtype=torch.uint8, device="cuda") for c in cams]  # (padded rows: a stride)
    tickets = [scene.render_device_begin(c, b.data_ptr(), (c.width + 5) * 4) for c, b in zip(cams, bufs)]
    assert len(set(tickets)) == len(tickets)
    for t in reversed(tickets):
        scene.render_device_wait(t)
    for c, b in zip(cams, bufs):
        ofb, *_ = oracle.render(oracle.make_cfg(c, params, 96, 96), heights, cmap)
        got = b.cpu().numpy()
        assert np.array_equal(got[:, :c.width], ofb), c.projection
        assert (got[:, c.width:] == 0).all()
    with pytest.raises(gpu.HmrmError):
        scene.render_device_wait(tickets[0])  # (already waited for)
    with env(HMRM_STEP_CAP=40):
        slow = gpu.Camera.make(width=64, height=40, projection=1, hang=gpu.degrees_to_rads(-45), vang=gpu.degrees_to_rads(112),
                               pos=(-30.0, 30.0, 40.0), step_dist=0.05)
        out = torch.zeros((40, 64, 4), dtype=torch.uint8, device="cuda")
        t = scene.render_device_begin(slow, out.data_ptr(), 64 * 4)
        with pytest.raises(gpu.HmrmError) as e:
            scene.render_device_wait(t)
        assert e.value.code == gpu.HMRM_E_NOTERM
    scene.close()
    # the rate on C2
    wl = gpu.synth.WORKLOADS["C2"]
    scene = gpu.Scene(*wl.maps(), wl.scene_params())
    cam = wl.camera()
    W, H = cam.width, cam.height
    fb = scene.render(cam)
    out = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(3)]
    for _ in range(12):
        scene.bench_kernel_ms(cam, 1)

    def one_stream(n):
        s0 = torch.cuda.current_stream().cuda_stream
        for _ in range(n):
            scene.render_rows_device(cam, out[0].data_ptr(), W * 4, 0, H, stream=s0)

    def lanes(n):
        inflight = []
        for i in range(n):
            if len(inflight) == 3:
                scene.render_device_wait(inflight.pop(0))
            inflight.append(scene.render_device_begin(cam, out[i % 3].data_ptr(), W * 4))
        for t in inflight:
            scene.render_device_wait(t)

    def ms(fn, n=300):
        fn(30)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n
    a = min(ms(one_stream) for _ in range(3))
    b = min(ms(lanes) for _ in range(3))
    for o in out:
        assert np.array_equal(o.cpu().numpy(), fb)
    print(f"C2: one stream {a:.4f} ms per frame, three tickets in flight {b:.4f} ({b / a:.3f} x)")
    assert b < 0.92 * a, (a, b)
    scene.close()


def test_scene_update_with_tickets_in_flight(gpu, oracle):
    """hmrm_scene_update while frames are in flight on the scene's launch lanes: the update drains the
    lanes and the ring's copy stream before it rewrites the threshold table and the pyramid, so every ticket begun
    before it finishes with the OLD heights -- never a torn mix of old thresholds and new window maxima -- and every
    frame begun after it sees the new ones.  Slow frames on purpose (the literal loop over a 512^2 map) so that the
    update really arrives while they run."""
    import torch
    rgb, cmap = gpu.synth.synth_maps(512)
    p_old = gpu.SceneParams.make(0.0, 40.0, grid_width=1.0)
    p_new = gpu.SceneParams.make(5.0, 90.0, lum=(0.2, 0.7, 0.1), grid_width=1.0)
    cam = gpu.Camera.make(width=480, height=270, projection=1, hfov=gpu.degrees_to_rads(90), hang=gpu.degrees_to_rads(-45),
                          vang=gpu.degrees_to_rads(115), pos=(-64.0, 64.0, 128.0), step_dist=0.25, bg=(1, 2, 3))
    want = {}
    for key, p in (("old", p_old), ("new", p_new)):
        heights = oracle.update_heightmap(rgb, p)
        want[key] = oracle.render(oracle.make_cfg(cam, p, 512, 512), heights, cmap)[0]
    assert not np.array_equal(want["old"], want["new"])
    for variant in ("simple", "leap"):
        with kernel_variant(variant):
            scene = gpu.Scene(rgb, cmap, p_old)
            for lap in range(3):
                scene.update(p_old)
                dev = [torch.zeros((270, 480, 4), dtype=torch.uint8, device="cuda") for _ in range(6)]
                tickets = [scene.render_device_begin(cam, b.data_ptr(), 480 * 4) for b in dev]
                ring = [scene.render_begin(cam) for _ in range(3)]
                scene.update(p_new)  # (frames of both kinds still in flight)
                after = torch.zeros((270, 480, 4), dtype=torch.uint8, device="cuda")
                t_after = scene.render_device_begin(cam, after.data_ptr(), 480 * 4)
                for t in tickets:
                    scene.render_device_wait(t)
                for b in dev:
                    assert np.array_equal(b.cpu().numpy(), want["old"]), (variant, lap)
                for t in ring:
                    assert np.array_equal(scene.render_wait(t, (270, 480)), want["old"]), (variant, lap)
                    scene.render_release(t)
                scene.render_device_wait(t_after)
                assert np.array_equal(after.cpu().numpy(), want["new"]), (variant, lap)
                assert np.array_equal(scene.render(cam), want["new"])
            scene.close()


def test_recording_sharded_over_scenes(gpu, oracle, tmp_path):
    """hmrm_record_orbit_multi (BASELINE config C5: frame k on GPU k mod N): with one scene per
    "device" (here two or three scenes on the one GPU of the box) the files are byte for byte those of
    the single-scene recording and the oracle's frames."""
    rgb, cmap = scenes.small_maps(64, 64, 43)
    params = gpu.SceneParams.make(0.0, 8.0, grid_width=1.0)
    base = gpu.Camera.make(width=96, height=54, projection=1, hfov=gpu.degrees_to_rads(80), hang=0.0,
                           vang=gpu.degrees_to_rads(112), pos=(-20.0, 20.0, 30.0), step_dist=0.5, bg=(4, 5, 6))
    heights = oracle.update_heightmap(rgb, params)
    frames, cx, cy, radius, hang0 = 11, 32.0, -32.0, 70.0, gpu.degrees_to_rads(-45.0)
    one = gpu.Scene(rgb, cmap, params)
    d1 = tmp_path / "one"
    d1.mkdir()
    gpu.record_orbit(one, base, cx, cy, radius, hang0, frames, str(d1), 7, encoder_threads=2)
    for n_scenes, threads in ((2, 3), (3, 1)):
        many = [gpu.Scene(rgb, cmap, params) for _ in range(n_scenes)]
        dn = tmp_path / f"many{n_scenes}"
        dn.mkdir()
        gpu.record_orbit_multi(many, base, cx, cy, radius, hang0, frames, str(dn), 7, encoder_threads=threads)
        assert sorted(p.name for p in dn.iterdir()) == sorted(f"hmap_7_{k}.png" for k in range(frames))
        for k in range(frames):
            assert (dn / f"hmap_7_{k}.png").read_bytes() == (d1 / f"hmap_7_{k}.png").read_bytes(), (n_scenes, k)
        for sc in many:
            sc.close()
    for k in (0, 5, 10):
        cam = gpu.orbit_camera(base, cx, cy, radius, hang0, k, frames)
        ofb, *_ = oracle.render(oracle.make_cfg(cam, params, 64, 64), heights, cmap)
        assert (d1 / f"hmap_7_{k}.png").read_bytes() == gpu.png_encode(ofb), k
    one.close()


def test_one_frame_over_several_scenes(gpu, oracle):
    """hmrm_render_multi (BASELINE config C4's sharding in the C ABI): 2, 3 and 5 scenes, each rendering
    its cyclic 16-row bands and copying them to their rows of the host frame, give the single-scene
    frame and the oracle's -- ragged height, all projections, and a capped scene reports its rays."""
    rgb, cmap = scenes.small_maps(128, 128, 29)
    params = gpu.SceneParams.make(0.0, 12.0, grid_width=1.0)
    heights = oracle.update_heightmap(rgb, params)
    many = [gpu.Scene(rgb, cmap, params) for _ in range(5)]
    for proj in (1, 2, 3):
        cam = gpu.Camera.make(width=211, height=117, projection=proj, hfov=gpu.degrees_to_rads(150 if proj == 2 else 85),
                              hang=gpu.degrees_to_rads(-45), vang=gpu.degrees_to_rads(112), pos=(-40.0, 40.0, 50.0),
                              ortho_width=0.9, step_dist=0.25, bg=(5, 6, 7))
        ofb, *_ = oracle.render(oracle.make_cfg(cam, params, 128, 128), heights, cmap)
        assert np.array_equal(many[0].render(cam), ofb)
        for n in (2, 3, 5):
            assert np.array_equal(gpu.render_multi(many[:n], cam), ofb), (proj, n)
    with env(HMRM_STEP_CAP=50):
        cam = gpu.Camera.make(width=64, height=40, projection=1, hang=gpu.degrees_to_rads(-45), vang=gpu.degrees_to_rads(112),
                              pos=(-40.0, 40.0, 50.0), step_dist=0.05)
        with pytest.raises(gpu.HmrmError) as e:
            gpu.render_multi(many[:3], cam)
        assert e.value.code == gpu.HMRM_E_NOTERM
    # a PINNED destination frame takes the direct path (bands copied straight into the caller's rows, with a row
    # stride wider than the frame); a pageable one goes through each scene's pinned staging strip: same pixels
    import ctypes
    import torch
    cam = gpu.Camera.make(width=211, height=117, projection=1, hfov=gpu.degrees_to_rads(85), hang=gpu.degrees_to_rads(-45),
                          vang=gpu.degrees_to_rads(112), pos=(-40.0, 40.0, 50.0), step_dist=0.25, bg=(5, 6, 7))
    ofb, *_ = oracle.render(oracle.make_cfg(cam, params, 128, 128), heights, cmap)
    stride = cam.width * 4 + 64
    for pinned in (True, False):
        host = torch.full((cam.height, stride), 9, dtype=torch.uint8)
        if pinned:
            host = host.pin_memory()
        for sc in many[:3]:
            sc._sync_env()  # (the step cap of the block above is gone: the raw C call below does not re-read the knobs)
        arr = (ctypes.c_void_p * 3)(*[sc._h for sc in many[:3]])
        rc = gpu.lib.lib.hmrm_render_multi(arr, 3, ctypes.byref(cam), ctypes.c_void_p(host.data_ptr()), stride)
        assert rc == 0, gpu.last_error()
        got = host.numpy()
        assert np.array_equal(got[:, :cam.width * 4].reshape(cam.height, cam.width, 4), ofb), pinned
        assert (got[:, cam.width * 4:] == 9).all(), "bytes between the rows were written"
    for sc in many:
        sc.close()


def test_strip_pipeline_renders_whole_frames_on_the_gpu(gpu, oracle):
    """strips.StripPipeline with the HIP renderer behind it, one rank: a sequence of orbit frames, each rendered
    in `chunks` interleaved band sets into double-buffered device strips (hmrm_render_rows_device in cyclic-band mode with
    band_count = chunks) and reassembled, equals the oracle's frame for its camera -- two frames in flight, buffers reused."""
    import torch
    strips = importlib.import_module("heightmap-ray-marcher_amd.strips")
    rgb, cmap = scenes.small_maps(96, 80, 53)
    params = gpu.SceneParams.make(0.0, 9.0, grid_width=1.0)
    heights = oracle.update_heightmap(rgb, params)
    scene = gpu.Scene(rgb, cmap, params)
    base = gpu.Camera.make(width=150, height=101, projection=1, hfov=gpu.degrees_to_rads(80), hang=0.0, vang=gpu.degrees_to_rads(114),
                           pos=(-30.0, 30.0, 40.0), step_dist=0.25, bg=(4, 5, 6))
    cams = [gpu.orbit_camera(base, 48.0, -40.0, 90.0, gpu.degrees_to_rads(-45.0), k, 6) for k in range(6)]
    want = [oracle.render(oracle.make_cfg(c, params, 96, 80), heights, cmap)[0] for c in cams]
    stream = torch.cuda.current_stream().cuda_stream
    for chunks, depth in ((1, 2), (3, 2), (2, 3)):
        plan = strips.BandPlan(height=base.height, width=base.width, band_rows=16, world=1)
        pipe = strips.StripPipeline(plan, 0, None, torch, "cuda", depth=depth, chunks=chunks)

        def render_rows_of(k):
            def render_rows(chunk_t, band_rows, band_index, band_count):
                chunk_t.zero_()
                scene.render_rows_device(cams[k], chunk_t.data_ptr(), base.width * 4, band_rows=band_rows, band_index=band_index,
                                         band_count=band_count, stream=stream)
            return render_rows
        got = {}
        pipe.run(range(6), render_rows_of, on_frame=lambda k, f: got.__setitem__(k, f.cpu().numpy()))
        for k in range(6):
            assert np.array_equal(got[k], want[k]), (chunks, depth, k)
    scene.close()


def test_two_streams_two_spherical_cameras(gpu, oracle):
    """hmrm_render_rows_device from two HIP streams with two different spherical cameras, alternating
    without a host sync in between: each stream has its own tables / counters (api.cpp StreamCtx), so
    neither launch can read the other camera's sin/cos tables."""
    import torch
    rgb, cmap = scenes.small_maps(128, 128, 61)
    params = gpu.SceneParams.make(0.0, 12.0, grid_width=1.0)
    scene = gpu.Scene(rgb, cmap, params)
    heights = oracle.update_heightmap(rgb, params)
    W, H = 320, 180
    cams = [gpu.Camera.make(width=W, height=H, projection=2, hfov=gpu.degrees_to_rads(fov), hang=gpu.degrees_to_rads(hang),
                            vang=gpu.degrees_to_rads(110), pos=pos, step_dist=0.25, bg=(7, 8, 9))
            for fov, hang, pos in ((170, -45, (-40.0, 40.0, 50.0)), (120, 135, (170.0, -170.0, 35.0)))]
    want = [oracle.render(oracle.make_cfg(c, params, 128, 128), heights, cmap)[0] for c in cams]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = [[torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(6)] for _ in range(2)]
    torch.cuda.synchronize()  # (the fills ran on torch's current stream, the renders use their own)
    for rep in range(6):
        for i in (0, 1):
            # stream i alternates between the two cameras as well: slot reuse within one stream
            cam = cams[(i + rep) % 2]
            scene.render_rows_device(cam, bufs[i][rep].data_ptr(), W * 4, 0, H, stream=streams[i].cuda_stream)
    torch.cuda.synchronize()
    for rep in range(6):
        for i in (0, 1):
            assert np.array_equal(bufs[i][rep].cpu().numpy(), want[(i + rep) % 2]), (rep, i)
    assert scene.take_capped(streams[0].cuda_stream) == 0 and scene.take_capped(streams[1].cuda_stream) == 0
    scene.close()


def test_moving_spherical_camera_shares_table_halves(gpu, oracle):
    """A moving spherical camera on one stream, no host sync between frames: the library copies the row tables
    (sin / cos of va) or the column tables (of ha) from a cached record when only the other half changed
    (api.cpp prepare_frame), fills the rest on its host pool and uploads with a kernel of the launch stream.
    Orbit steps (hang and pos change), tilts (vang), both, a translation, a bigger frame in between (the
    per-stream table arena is re-allocated) and more cameras than the cache has slots: every frame equals the
    oracle's (Spherical.cpp:17-31)."""
    import torch
    rgb, cmap = scenes.small_maps(96, 96, 62)
    params = gpu.SceneParams.make(0.0, 10.0, grid_width=1.0)
    scene = gpu.Scene(rgb, cmap, params)
    heights = oracle.update_heightmap(rgb, params)

    def cam(hang, vang, pos=(-30.0, 30.0, 40.0), W=256, H=144, fov=170.0):
        return gpu.Camera.make(width=W, height=H, projection=2, hfov=gpu.degrees_to_rads(fov), hang=gpu.degrees_to_rads(hang),
                               vang=gpu.degrees_to_rads(vang), pos=pos, step_dist=0.25, bg=(5, 6, 7))
    seq = [cam(-45, 110), cam(-40, 110), cam(-45, 120), cam(-40, 120), cam(-40, 120, pos=(-20.0, 35.0, 44.0)),
           cam(-45, 110, W=1400, H=900), cam(-45, 110), cam(-40, 120, fov=120.0), cam(-40, 110, fov=120.0)]
    seq += [cam(-45 + 0.5 * k, 110) for k in range(70)]   # (more than the 64 cached records of a stream)
    seq += [cam(-45, 100 + 0.25 * k) for k in range(10)] + [cam(-45, 110), cam(-40, 110)]
    st = torch.cuda.Stream()
    outs = [torch.zeros((c.height, c.width, 4), dtype=torch.uint8, device="cuda") for c in seq]
    torch.cuda.synchronize()
    for c, o in zip(seq, outs):
        scene.render_rows_device(c, o.data_ptr(), c.width * 4, 0, c.height, stream=st.cuda_stream)
    torch.cuda.synchronize()
    for i, (c, o) in enumerate(zip(seq, outs)):
        want = oracle.render(oracle.make_cfg(c, params, 96, 96), heights, cmap)[0]
        assert np.array_equal(o.cpu().numpy(), want), i
    assert scene.take_capped(st.cuda_stream) == 0
    scene.close()


def test_two_host_threads_share_one_scene(gpu, oracle):
    """SURVEY 8(b) "Threading": two host threads drive one scene at the same time, each on its own HIP
    stream with its own cameras (spherical and perspective, so per-frame tables are in play); every frame
    equals the oracle's.  The scene serialises only the host-side set-up, not the launches."""
    import threading
    import torch
    rgb, cmap = scenes.small_maps(128, 128, 67)
    params = gpu.SceneParams.make(0.0, 12.0, grid_width=1.0)
    scene = gpu.Scene(rgb, cmap, params)
    heights = oracle.update_heightmap(rgb, params)
    W, H = 256, 144
    base = gpu.Camera.make(width=W, height=H, projection=2, hfov=gpu.degrees_to_rads(160), hang=0.0,
                           vang=gpu.degrees_to_rads(110), pos=(-40.0, 40.0, 50.0), step_dist=0.25, bg=(1, 2, 3))
    cams = []
    for k in range(6):
        c = gpu.orbit_camera(base, 64.0, -64.0, 120.0, gpu.degrees_to_rads(-45.0), k, 6)
        c.projection = 2 if k % 2 == 0 else 1
        c.hfov = gpu.degrees_to_rads(160 if k % 2 == 0 else 80)
        cams.append(c)
    want = [oracle.render(oracle.make_cfg(c, params, 128, 128), heights, cmap)[0] for c in cams]
    errors = []

    def worker(tid):
        try:
            torch.cuda.set_device(0)
            gpu.set_device(0)
            stream = torch.cuda.Stream()
            buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()  # (the fill ran on torch's current stream, the renders use `stream`)
            for rep in range(8):
                for k in range(tid, 6, 2):
                    scene.render_rows_device(cams[k], buf.data_ptr(), W * 4, 0, H, stream=stream.cuda_stream)
                    stream.synchronize()
                    if not np.array_equal(buf.cpu().numpy(), want[k]):
                        errors.append((tid, rep, k))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    scene.close()


def test_rows_device_reports_capped_rays(gpu):
    """The asynchronous strip entry cannot return HMRM_E_NOTERM itself; hmrm_scene_take_capped does,
    once, for the launches of that stream (the cap must never be silent)."""
    import torch
    rgb = np.zeros((8, 8, 3), dtype=np.uint8)
    cmap = np.full((8, 8, 4), 255, dtype=np.uint8)
    params = gpu.SceneParams.make(0.0, 4.0, grid_width=1.0)
    cam = gpu.Camera.make(width=4, height=4, projection=3, hang=0.0, vang=0.0, pos=(4.0, -4.0, -3.0),
                          ortho_width=0.5, step_dist=0.0, bg=(9, 8, 7))  # straight up, step 0: never ends
    with env(HMRM_STEP_CAP=500):
        scene = gpu.Scene(rgb, cmap, params)
        buf = torch.zeros((4, 4, 4), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for variant in KERNEL_VARIANTS:
            with kernel_variant(variant):
                scene.render_rows_device(cam, buf.data_ptr(), 16, 0, 4, stream=st)
                scene.render_rows_device(cam, buf.data_ptr(), 16, 0, 2, stream=st)
                with pytest.raises(gpu.HmrmError) as e:
                    scene.take_capped(st)
                assert e.value.code == gpu.HMRM_E_NOTERM and "24 ray(s)" in e.value.message, variant
                assert scene.take_capped(st) == 0           # reported once
        scene.close()


def test_bench_two_ranks_sharing_this_gpu(gpu):
    """The N > 1 line of bench.py with TWO ranks for real -- its own launcher, torch.distributed rendezvous, the orbit leg
    sharded frame k -> rank k mod 2, the one-GPU reference leg where rank 1 only stands in the barriers, every pixel check --
    on this one-GPU box: `--share-gpu` lets both ranks render on the same device and runs the collectives over gloo on host
    copies.  (No 8-GPU node has been available in any round; this is the nearest thing to a run of that code path.)"""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env2 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HMRM_FORCE_DIST")}
    env2["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--workload", "C2", "--steps", "6",
                        "--warmup", "2", "--no-c4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env2)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["share_gpu"] is True and line["scaling"] == "weak"
    assert line["rccl_ranks"]["world_size"] == 2 and line["rccl_ranks"]["backend"] == "gloo"
    # (no bound on the efficiency: two processes time-slicing one GPU say nothing about scaling, only that the legs ran)
    assert line["value"] > 0 and line["one_gpu_same_leg"]["value"] > 0 and line["efficiency_vs_one_gpu"] > 0
    assert "C2" in line["config"]["workload"] and "frame k on GPU k mod 2" in line["config"]["parallelism"]
    # the strips mode: one frame in cyclic bands over the two ranks, gathered to rank 0 and checked there
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--workload", "C2", "--mode", "strips",
                        "--steps", "4", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env2)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["scaling"] == "strong" and "cyclic 16-row bands over 2 GPUs" in line["config"]["parallelism"]


def test_strips_over_rccl_two_gpus(gpu):
    """bench.py --mode strips on two GPUs over RCCL (C2: cyclic 16-row bands, gather to rank 0, frame
    checked against the single-GPU frame inside bench.py).  Skips on a one-GPU box."""
    if gpu.device_count() < 2:
        pytest.skip("needs two GPUs")
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("strips", "frames"):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(root, "bench.py"),
                            "--gpus", "2", "--steps", "4", "--warmup", "2", "--workload", "C2", "--mode", mode,
                            "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["value"] > 0


def test_float_heights_mode_bit_exact_vs_its_oracle_definition(gpu, oracle):
    """`heights f32` (HMRM_NEAREST_F32; north_star "float heights"): the reference's loop with fp64
    positions and thresholds (float)(heightmap_buf[i] + min_height).  Its definition is the oracle's
    (sampling 2) and the GPU matches it bit for bit -- frame, per-ray steps, distance() -- on every
    edge scene and on random ones, through leaps and float-rounded window maxima."""
    todo = [scenes.build_case(c) for c in scenes.cases()]
    rng = np.random.RandomState(808)
    for trial in range(16):
        mw, mh = int(rng.choice([96, 257, 600])), int(rng.choice([131, 300]))
        rgb, cmap = scenes.small_maps(mw, mh, 7000 + trial, color_heights=bool(trial % 2))
        gw = float(rng.choice([1.0, 0.5, 0.3]))
        hi = float(rng.uniform(3.0, 0.2 * mw)) * gw
        params = gpu.SceneParams.make(float(rng.choice([0.0, -1.5])), hi, grid_width=gw)
        ext = max(mw, mh) * gw
        ang = rng.uniform(0, 2 * np.pi)
        pos = (mw * gw / 2 + 0.9 * ext * np.cos(ang), -mh * gw / 2 + 0.9 * ext * np.sin(ang), hi * rng.uniform(1.0, 2.5))
        cam = gpu.Camera.make(width=int(rng.randint(40, 160)), height=int(rng.randint(30, 100)), projection=int(rng.choice([1, 2, 3])),
                              hfov=float(gpu.degrees_to_rads(rng.uniform(50, 170))),
                              hang=float(np.arctan2(-mh * gw / 2 - pos[1], mw * gw / 2 - pos[0])),
                              vang=float(gpu.degrees_to_rads(rng.uniform(95, 140))), pos=pos,
                              ortho_width=float(rng.uniform(0.5, 3.0) * gw), step_dist=float(rng.choice([0.1, 0.25, 0.5]) * gw))
        todo.append((f"fuzz{trial}", rgb, cmap, params, cam))
    leaped = 0
    with env(HMRM_STEP_CAP=300000):
        for name, rgb, cmap, params, cam in todo:
            cam.sampling = gpu.NEAREST_F32
            scene = gpu.Scene(rgb, cmap, params)
            heights = oracle.update_heightmap(rgb, params)
            cfg = oracle.make_cfg(cam, params, rgb.shape[1], rgb.shape[0], step_cap=300000)
            ofb, total, capped, osteps, oentry = oracle.render(cfg, heights, cmap, per_pixel=True)
            for variant in ("leap", "group", "rec"):
                with kernel_variant(variant):
                    fb, st, steps, entry = scene.render_stats(cam, per_pixel=True, allow_capped=True)
                ok = osteps >= 0
                assert np.array_equal(fb, ofb), (name, variant)
                assert np.array_equal(steps.astype(np.int64)[ok], osteps[ok]) and st.steps == total, (name, variant)
                assert np.array_equal(_bits(entry), _bits(oentry)) and st.capped == capped, (name, variant)
                leaped += st.leaped_steps if variant == "leap" else 0
            if capped == 0:
                assert np.array_equal(scene.render(cam), ofb), name
            scene.close()
    assert leaped > 0


@pytest.mark.parametrize("wl_name", ["C2", "C3"])
def test_float_heights_mode_tolerance_against_f64(gpu, wl_name):
    """north_star's bar for float heights ("within 1 ULP on the hit-point t"), made concrete: the
    slab-entry t is bit-identical (positions stay fp64); a threshold moves by at most half a float ulp
    (relative 2^-24), so a ray's hit step changes only where z passed within that of the threshold:
    a handful of rays per frame take a different number of steps (measured on MI355X: C3 12 of
    1 056 334 entering rays) and the frames agree to > 60 dB PSNR.  Tolerance written here: at most
    1e-4 of the entering rays differ in step count."""
    wl = gpu.synth.WORKLOADS[wl_name]
    rgb, cmap = gpu.synth.synth_maps(wl.map_size)
    params, cam = wl.scene_params(), wl.camera()
    scene = gpu.Scene(rgb, cmap, params)
    fb64, st64, steps64, entry64 = scene.render_stats(cam, per_pixel=True)
    cam32 = wl.camera()
    cam32.sampling = gpu.NEAREST_F32
    fb32, st32, steps32, entry32 = scene.render_stats(cam32, per_pixel=True)
    assert np.array_equal(_bits(entry32), _bits(entry64)), "slab-entry t must not depend on the height type"
    entering = int((steps64 > 0).sum())
    differ = int((steps32 != steps64).sum())
    assert differ <= max(2, int(1e-4 * entering)), (differ, entering)
    mse = float(((fb32.astype(np.float64) - fb64.astype(np.float64)) ** 2).mean())
    psnr = float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)
    assert psnr > 60.0, psnr
    assert st32.capped == 0 and abs(int(st32.steps) - int(st64.steps)) <= 4096 * max(1, differ)
    assert np.array_equal(scene.render(cam32), fb32)
    scene.close()


@pytest.mark.parametrize("frame", [16, 32, 48])
def test_c5_orbit_frames_full_size(gpu, oracle, frame):
    """BASELINE config C5 beyond frame 0: orbit frames 16, 32 and 48 of 64 at 3840x2160 over the 4096^2
    map (the camera looks at the map from the other three sides), every 48th row against the oracle."""
    wl = gpu.synth.WORKLOADS["C5"]
    rgb, cmap = gpu.synth.synth_maps(wl.map_size)
    params, cam = wl.scene_params(), wl.camera(frame, 64)
    scene = gpu.Scene(rgb, cmap, params)
    fb, st, steps, entry = scene.render_stats(cam, per_pixel=True)
    assert np.array_equal(scene.render(cam), fb) and st.capped == 0 and st.hits > 100000
    heights = oracle.update_heightmap(rgb, params)
    stride = 48
    ofb, _, capped, osteps, oentry = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size), heights, cmap,
                                                   per_pixel=True, row_stride=stride)
    rows = slice(0, cam.height, stride)
    assert capped == 0 and np.array_equal(fb[rows], ofb[rows])
    assert np.array_equal(steps[rows].astype(np.int64), osteps[rows]) and np.array_equal(_bits(entry[rows]), _bits(oentry[rows]))
    scene.close()


def test_c4_eight_rank_band_emulation_full_size(gpu, oracle):
    """BASELINE config C4's sharding at its full size on one device: the 7680x4320 orthographic frame
    over the 8192^2 map as eight ranks' cyclic 16-row bands (hmrm_render_rows_device, one strip per
    "rank"), reassembled as the gather would; equals the single-launch frame and, on every 96th row,
    the oracle."""
    import torch
    strips = importlib.import_module("heightmap-ray-marcher_amd.strips")
    wl = gpu.synth.WORKLOADS["C4"]
    rgb, cmap = gpu.synth.synth_maps(wl.map_size)
    params, cam = wl.scene_params(), wl.camera()
    scene = gpu.Scene(rgb, cmap, params)
    full = scene.render(cam)
    world, band = 8, 16
    plan = strips.BandPlan(height=cam.height, width=cam.width, band_rows=band, world=world)
    st = torch.cuda.current_stream().cuda_stream
    block = torch.zeros((world, plan.strip_rows, cam.width, 4), dtype=torch.uint8, device="cuda")
    for rank in range(world):
        assert gpu.band_local_rows(cam.height, band, rank, world) <= plan.strip_rows
        scene.render_rows_device(cam, block[rank].data_ptr(), cam.width * 4, band_rows=band, band_index=rank,
                                 band_count=world, stream=st)
    frame = strips.reassemble_torch(plan, block).cpu().numpy()
    assert scene.take_capped(st) == 0
    assert np.array_equal(frame, full)
    heights = oracle.update_heightmap(rgb, params)
    stride = 96
    ofb, _, capped, *_ = oracle.render(oracle.make_cfg(cam, params, wl.map_size, wl.map_size), heights, cmap, row_stride=stride)
    rows = slice(0, cam.height, stride)
    assert capped == 0 and np.array_equal(frame[rows], ofb[rows])
    scene.close()


# The four slices share ONE time budget (conftest.FUZZ_BUDGET_S, env HMRM_FUZZ_BUDGET_S, default 270 s) in the proportions
# 140 : 140 : 90 : 50 : 40 (round 3's four fixed slices were 150 / 150 / 100 / 60 s; round 4 added the cell-boundary fuzzer).
_FUZZ_SHARES = (("deep_fuzz.py", 140, []), ("deep_fuzz_big.py", 140, ["4096"]), ("deep_fuzz_edges.py", 90, []),
                ("deep_fuzz_binades.py", 50, []), ("deep_fuzz_cells.py", 40, []))


def _fuzz_cases():
    import conftest
    total = sum(w for _, w, _ in _FUZZ_SHARES)
    return [(script, ["20260000", "1000000", str(max(10, int(conftest.FUZZ_BUDGET_S * w / total)))] + extra)
            for script, w, extra in _FUZZ_SHARES]


@pytest.mark.parametrize("script,args", _fuzz_cases(), ids=[c[0] for c in _FUZZ_SHARES])
def test_deep_fuzz_slice(gpu, script, args):
    """A seeded time-boxed slice of each long fuzzer (tests/deep_fuzz*.py: random maps from 1x1 up, random cameras
    over the 4096^2 map, cameras whose rays graze the box's edges and corners, and long low maps crossed end to end by
    shallow rays through a dozen binades, and rays along / onto cell boundaries under general grid widths incl. the ones
    whose reciprocal is an integer; all projections and sampling modes, GPU vs oracle on frames, per-ray step
    counts, distance() bits and cap counts) inside the suite the driver runs."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, script)] + args, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "mismatches 0" in r.stdout, tail
    import re
    m = re.search(r"^scenes (\d+), mismatches 0|: cameras (\d+), mismatches 0|: scenes (\d+), mismatches 0", r.stdout, flags=re.M)
    assert m and int(m.group(1) or m.group(2) or m.group(3)) >= int(args[2]), tail  # (at least one scene per second of the slice)
    print(tail.strip().splitlines()[-1])  # (visible with -rP / in the junit output: how far the slice got)


def test_renders_repeat_bit_for_bit(gpu):
    """A time-boxed slice of tests/stress_repeat.py: thousands of small random scenes created, rendered seven times each with the
    instrumented and the plain kernel, and destroyed; every render must equal its scene's first one bit for bit (frames, per-ray
    step counts, distance() bits).  Looks for races -- copies, tables, stale buffers --, not for arithmetic (builder run, round 5:
    79 976 scenes, 959 712 renders, none differed)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "stress_repeat.py"), "6000000", "15", "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "nondeterministic 0" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_rcp_f64_accuracy_bound(gpu):
    """v_rcp_f64 on THIS device, measured: slab_classify's margins (device_common.hpp kRcpRelErr = 2^-24,
    kSlabMargin = 16x that) hold only while the hardware reciprocal stays within that relative error.  The leading
    32 mantissa bits exhaustively, hashed inputs over the whole exponent range, and the product the shortcut forms
    (n * rcp(d) against n / d); the full sweep is tools/rcp_accuracy.py -> profiles/r03_rcp_accuracy.txt."""
    bound = 2.0 ** -24
    worst, n = 0.0, 0
    lines = []
    for args in ((0, 1 << 32, 2, 0, 0), (0, 1 << 32, 1, 0, 0), (1, 1 << 31, 11, -40, 0), (1, 1 << 31, 12, -1, 14),
                 (1, 1 << 31, 13, -1000, 1000), (2, 1 << 31, 14, -10, 14), (2, 1 << 31, 15, -60, 60)):
        m, hist = gpu.rcp_error(*args)
        assert int(hist.sum()) == args[1]
        lines.append(f"mode {args[0]} n {args[1]} seed {args[2]} exponents [{args[3]}, {args[4]}]: max rel err {m:.6e} = 2^{np.log2(m):.3f}")
        worst, n = max(worst, m), n + args[1]
    lines.append(f"total {n} inputs, worst {worst:.6e} = 2^{np.log2(worst):.3f}; bound used by the kernel 2^-24")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "rcp_accuracy_in_test.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    assert n >= (1 << 32) and 0.0 < worst <= bound, lines


def test_rays_grazing_box_edges_and_corners(gpu, oracle):
    """Adversarial for the one-division shortcut through distance() (AABB.cpp:49-77): frames whose neighbouring
    rays pass an edge or a corner of the box at relative offsets 2^-14 .. 2^-52, on both sides, all three
    projections (tests/deep_fuzz_edges.py; tests/test_slab_classify_model.py shows that 60 % of these rays fall
    inside the shortcut's margin).  distance() bits, frames and per-ray step counts against the oracle."""
    import deep_fuzz_edges as edges
    cache = {}
    strad = {1: 0, 2: 0, 3: 0}
    fine = 0
    for seed in range(900):
        ok, s, cam, lg = edges.run_case(cache, oracle, scenes, seed)
        assert ok, (seed, cam.projection, lg)
        strad[cam.projection] += int(s)
        fine += int(s and lg < -24)
    for sc, *_ in cache.values():
        sc.close()
    # the frames do straddle silhouette edges, in every projection, also below the reciprocal's resolution
    assert min(strad.values()) >= 60 and fine >= 100, (strad, fine)


def test_progressive_cycle_refresh(gpu, oracle):
    """`cycle n` (hmap.cpp:976-983): each call rewrites pixels p = cycle (mod n); n calls give the frame."""
    rgb, cmap = scenes.small_maps(64, 64, 55)
    params = gpu.SceneParams.make(0.0, 8.0, grid_width=1.0)
    cam = gpu.Camera.make(width=53, height=37, projection=1, hfov=gpu.degrees_to_rads(90), hang=gpu.degrees_to_rads(-45),
                          vang=gpu.degrees_to_rads(118), pos=(-10.0, 10.0, 24.0), step_dist=0.5, bg=(12, 34, 56))
    scene = gpu.Scene(rgb, cmap, params)
    full = scene.render(cam)
    period = 47  # the reference's default cycle_period (hmap.cpp:71)
    fb = np.full_like(full, 7)
    scene.render_cycle(cam, fb, 3, period)
    p = np.arange(cam.width * cam.height).reshape(cam.height, cam.width)
    sel = (p % period) == 3
    assert np.array_equal(fb[sel], full[sel]) and (fb[~sel] == 7).all()
    for c in range(period):
        scene.render_cycle(cam, fb, c, period)
    assert np.array_equal(fb, full)
    with pytest.raises(gpu.HmrmError):
        scene.render_cycle(cam, fb, 47, 47)
    scene.close()


def test_render_argument_validation(gpu):
    import ctypes as C
    from importlib import import_module
    lib = import_module("heightmap-ray-marcher_amd.lib").lib
    rgb, cmap = scenes.small_maps(32, 32, 3)
    scene = gpu.Scene(rgb, cmap, gpu.SceneParams.make(0.0, 4.0, grid_width=1.0))
    cam = gpu.Camera.make(width=16, height=16, pos=(-4.0, 4.0, 8.0), vang=gpu.degrees_to_rads(115))
    fb = np.zeros((16, 16, 4), dtype=np.uint8)
    assert lib.hmrm_render(scene._h, C.byref(cam), fb.ctypes.data, 16 * 4 - 1) == gpu.HMRM_E_ARG   # stride too small
    assert lib.hmrm_render(scene._h, C.byref(cam), None, 64) == gpu.HMRM_E_ARG
    assert lib.hmrm_render(None, C.byref(cam), fb.ctypes.data, 64) == gpu.HMRM_E_ARG
    for args in ((5, 4, 0, 0, 1), (0, 17, 0, 0, 1), (-1, 4, 0, 0, 1), (0, 0, 16, 2, 2), (0, 0, 16, -1, 2)):
        with pytest.raises(gpu.HmrmError) as e:
            scene.render_rows_device(cam, 4096, 64, *args)
        assert e.value.code == gpu.HMRM_E_ARG
    with pytest.raises(gpu.HmrmError):
        scene.debug_ray(cam, 16, 0)
    with pytest.raises(ValueError):
        gpu.Scene(rgb, cmap[:16], gpu.SceneParams.make())     # hmap.cpp:503-515 dimension check
    # padded stride: rows land stride_bytes apart, padding untouched
    wide = np.full((16, 24, 4), 9, dtype=np.uint8)
    assert lib.hmrm_render(scene._h, C.byref(cam), wide.ctypes.data, 24 * 4) == gpu.HMRM_OK
    assert np.array_equal(wide[:, :16], scene.render(cam)) and (wide[:, 16:] == 9).all()
    scene.close()


# ---- bilinear quality mode (additive): defined by the oracle, not the reference ----

def _bilinear(cam):
    import copy
    c = copy.copy(cam)
    c.sampling = 1
    return c


@pytest.mark.parametrize("case", scenes.cases(), ids=scenes.case_ids())
def test_bilinear_mode_bit_exact(gpu, oracle, case):
    """`sampling bilinear`: interpolated height test + interpolated colour; every other rule of the
    loop (entry nudge, range test, sky, alpha-0 on the nearest cell) unchanged.  Leaps run on the
    3x3-dilated pyramid; frame and per-ray step counts must equal the oracle's."""
    name, rgb, cmap, params, cam = scenes.build_case(case)
    cam = _bilinear(cam)
    scene = gpu.Scene(rgb, cmap, params)
    heights = oracle.update_heightmap(rgb, params)
    cfg = oracle.make_cfg(cam, params, rgb.shape[1], rgb.shape[0])
    assert cfg.sampling == 1
    ofb, total, capped, osteps, oentry = oracle.render(cfg, heights, cmap, per_pixel=True)
    for variant in ("leap", "group", "simple", "rec"):  # "simple" and "rec" have no bilinear loop: served by "group"
        with kernel_variant(variant):
            fb, st, steps, entry = scene.render_stats(cam, per_pixel=True)
            assert np.array_equal(_bits(entry), _bits(oentry)), variant
            assert np.array_equal(steps.astype(np.int64), osteps), variant
            assert np.array_equal(fb, ofb), variant
            assert (st.rays, st.steps, st.capped) == (cam.width * cam.height, total, 0), variant
            assert np.array_equal(scene.render(cam), ofb), variant
    scene.close()


def test_bilinear_fuzz_update_and_cache(gpu, oracle):
    """Seeded fuzz of the bilinear mode (all three grid-width paths, all projections), plus: a height
    update between two bilinear frames of the same camera rebuilds the dilated pyramid, and
    nearest/bilinear frames of one scene do not leak state into each other."""
    rng = np.random.RandomState(77)
    os.environ["HMRM_STEP_CAP"] = "400000"
    leaped = differs = 0
    try:
        for trial in range(30):
            mw, mh = int(rng.choice([48, 96, 160, 257])), int(rng.choice([48, 96, 131]))
            rgb, cmap = scenes.small_maps(mw, mh, 3000 + trial, color_heights=bool(trial % 3 == 0))
            gw = float(rng.choice([1.0, 0.5, 0.05, 0.3, 1.7]))
            lo = float(rng.choice([0.0, -1.5, 2.0]))
            hi = lo + float(rng.uniform(0.5, 0.3 * mw)) * gw
            params = gpu.SceneParams.make(lo, hi, grid_width=gw)
            ext_x, ext_y = mw * gw, mh * gw
            ang = rng.uniform(0, 2 * np.pi)
            dist = rng.uniform(0.2, 1.6) * max(ext_x, ext_y)
            pos = (ext_x / 2 + dist * np.cos(ang), -ext_y / 2 + dist * np.sin(ang), hi + rng.uniform(-0.5, 2.0) * (hi - lo + gw))
            hang = float(np.arctan2(-ext_y / 2 - pos[1], ext_x / 2 - pos[0]) + rng.uniform(-0.4, 0.4))
            cam = gpu.Camera.make(width=int(rng.randint(17, 90)), height=int(rng.randint(9, 70)),
                                  projection=int(rng.choice([1, 2, 3])),
                                  hfov=float(gpu.degrees_to_rads(rng.uniform(30, 175))), hang=hang,
                                  vang=float(gpu.degrees_to_rads(rng.uniform(60, 170))), pos=pos,
                                  ortho_width=float(rng.uniform(0.2, 3.0) * gw),
                                  step_dist=float(rng.choice([0.05, 0.1, 0.25, 0.5, 0.37]) * gw),
                                  bg=tuple(int(v) for v in rng.randint(0, 256, size=3)), sampling=gpu.BILINEAR)
            near_cam = gpu.Camera.make(width=cam.width, height=cam.height, projection=cam.projection, hfov=cam.hfov,
                                       hang=cam.hang, vang=cam.vang, pos=tuple(cam.pos), ortho_width=cam.ortho_width,
                                       step_dist=cam.step_dist, bg=(cam.bg_r, cam.bg_g, cam.bg_b))
            heights = oracle.update_heightmap(rgb, params)
            ofb, total, capped, osteps, _ = oracle.render(oracle.make_cfg(cam, params, mw, mh, step_cap=400000),
                                                          heights, cmap, per_pixel=True)
            onear, *_ = oracle.render(oracle.make_cfg(near_cam, params, mw, mh, step_cap=400000), heights, cmap)
            scene = gpu.Scene(rgb, cmap, params)
            fb, st, steps, _ = scene.render_stats(cam, per_pixel=True, allow_capped=True)
            assert np.array_equal(fb, ofb), trial
            if capped == 0:
                assert np.array_equal(steps.astype(np.int64), osteps) and st.steps == total, trial
            assert st.capped == capped, trial
            if capped == 0:
                assert np.array_equal(scene.render(near_cam), onear), trial
                assert np.array_equal(scene.render(cam), ofb), trial
            leaped += st.leaped_steps
            differs += int(not np.array_equal(ofb, onear))
            if trial % 5 == 0 and capped == 0:
                # new heights, same camera: the cached frame and the dilated pyramid must both refresh
                params2 = gpu.SceneParams.make(lo, lo + 0.5 * (hi - lo), grid_width=gw)
                scene.update(params2)
                h2 = oracle.update_heightmap(rgb, params2)
                cfg2 = oracle.make_cfg(cam, params2, mw, mh, step_cap=400000)
                o2, _, cap2, *_ = oracle.render(cfg2, h2, cmap)
                if cap2 == 0:
                    assert np.array_equal(scene.render(cam), o2), trial
            scene.close()
    finally:
        del os.environ["HMRM_STEP_CAP"]
    assert leaped > 0 and differs > 20


def test_bilinear_full_size_c3_subsampled(gpu, oracle):
    """BASELINE's C3 frame in bilinear mode: every 48th row against the oracle, determinism, and the
    instrumented kernel writes the same frame as the plain one."""
    wl = gpu.synth.WORKLOADS["C3"]
    rgb, cmap = gpu.synth.synth_maps(wl.map_size)
    params, cam = wl.scene_params(), wl.camera()
    cam.sampling = gpu.BILINEAR
    scene = gpu.Scene(rgb, cmap, params)
    fb = scene.render(cam)
    assert np.array_equal(fb, scene.render(cam))
    fbs, st, steps, _ = scene.render_stats(cam, per_pixel=True)
    assert np.array_equal(fbs, fb) and st.capped == 0 and st.leaped_steps > 0
    heights = oracle.update_heightmap(rgb, params)
    cfg = oracle.make_cfg(cam, params, wl.map_size, wl.map_size)
    stride = 48
    ofb, total, capped, osteps, _ = oracle.render(cfg, heights, cmap, per_pixel=True, row_stride=stride)
    rows = slice(0, cam.height, stride)
    assert capped == 0
    assert np.array_equal(fb[rows], ofb[rows])
    assert np.array_equal(steps[rows].astype(np.int64), osteps[rows])
    scene.close()


def test_launch_order_never_changes_a_pixel(gpu, oracle):
    """The cost-rotated launch order (api.cpp choose_tile_rot) is scheduling only: HMRM_TILE_ORDER=0
    (row-major) and the default order give identical frames, counters and strips; checked where the
    rotation is non-trivial (horizon in mid-frame) and on a row strip that starts past it."""
    import torch
    rgb, cmap = scenes.small_maps(160, 131, 4242)
    params = gpu.SceneParams.make(0.0, 12.0, grid_width=1.0)
    heights = oracle.update_heightmap(rgb, params)
    for proj, vang in ((2, 95.0), (1, 80.0), (3, 120.0)):
        cam = gpu.Camera.make(width=203, height=157, projection=proj, hfov=gpu.degrees_to_rads(150 if proj == 2 else 80),
                              hang=gpu.degrees_to_rads(-45), vang=gpu.degrees_to_rads(vang), pos=(-20.0, 20.0, 30.0),
                              ortho_width=1.1, step_dist=0.25, bg=(1, 2, 3))
        ofb, total, capped, *_ = oracle.render(oracle.make_cfg(cam, params, 160, 131), heights, cmap)
        assert capped == 0
        frames = {}
        for order in ("0", "1"):
            os.environ["HMRM_TILE_ORDER"] = order
            try:
                scene = gpu.Scene(rgb, cmap, params)
                fb, st, *_ = scene.render_stats(cam)
                buf = torch.zeros((131 - 48, cam.width, 4), dtype=torch.uint8, device="cuda")
                scene.render_rows_device(cam, buf.data_ptr(), cam.width * 4, 48, 131,
                                         stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                strip = buf.cpu().numpy()
                frames[order] = (fb, st.steps, scene.render(cam), strip)
                scene.close()
            finally:
                del os.environ["HMRM_TILE_ORDER"]
        for order in ("0", "1"):
            fb, steps, plain, strip = frames[order]
            assert np.array_equal(fb, ofb) and np.array_equal(plain, ofb) and steps == total, (proj, order)
            assert np.array_equal(strip, ofb[48:131]), (proj, order)


def test_calibrated_and_pieced_launch_orders_never_change_a_pixel(gpu, oracle, capfd):
    """The launch order beyond the rotation (api.cpp: up to three tile-row pieces started first, the rest wrapping
    around; calibrated per cached camera from a measured launch, then verified by a second measured launch) is
    scheduling only.  A camera rendered again and again walks through every calibration state -- plain, measured,
    candidate measured, settled -- and every frame, on the scene's stream and on a caller's, equals the oracle's;
    so do frames under explicit pieces (HMRM_TILE_SEGMENTS, a tool knob), valid ones in any order and invalid ones
    (overlapping, not contiguous, out of range), which fall back to the rotation."""
    import torch
    rgb, cmap = scenes.small_maps(200, 180, 4243)
    params = gpu.SceneParams.make(0.0, 14.0, grid_width=1.0)
    heights = oracle.update_heightmap(rgb, params)
    st = torch.cuda.Stream()
    with env(HMRM_ORDER_VERBOSE=1):
        scene = gpu.Scene(rgb, cmap, params)
        for proj, vang in ((2, 100.0), (1, 95.0), (3, 125.0)):
            cam = gpu.Camera.make(width=251, height=420, projection=proj, hfov=gpu.degrees_to_rads(160 if proj == 2 else 85),
                                  hang=gpu.degrees_to_rads(-45), vang=gpu.degrees_to_rads(vang), pos=(-25.0, 25.0, 34.0),
                                  ortho_width=0.6, step_dist=0.25, bg=(4, 5, 6))
            ofb, *_ = oracle.render(oracle.make_cfg(cam, params, 200, 180), heights, cmap)
            buf = torch.zeros((cam.height, cam.width, 4), dtype=torch.uint8, device="cuda")
            for rep in range(12):
                assert np.array_equal(scene.render(cam), ofb), (proj, rep)
                scene.render_rows_device(cam, buf.data_ptr(), cam.width * 4, 0, cam.height, stream=st.cuda_stream)
                torch.cuda.synchronize()
                assert np.array_equal(buf.cpu().numpy(), ofb), (proj, rep)
            for segs in ("9:5,20:6,14:6", "20:6,9:5,14:6", "0:27", "3:4", "26:1,25:1,24:1", "9:5,12:6", "9:5,20:6", "5:40", "9:5,14:6,20:6"):
                with env(HMRM_TILE_SEGMENTS=segs):
                    assert np.array_equal(scene.render(cam), ofb), (proj, segs)
        scene.close()
    err = capfd.readouterr().err
    assert err.count("hmrm order: trial 0 of") >= 3 and err.count("hmrm order: settled") >= 3, err   # every camera was calibrated


def test_very_tall_frame_uses_the_third_grid_dimension(gpu, oracle):
    """More than 32768 tile rows (16 pixels each): the launch folds the rows into grid y and z."""
    import conftest
    conftest.skip_if_over_budget(8, "very tall frame")
    rgb, cmap = scenes.small_maps(40, 33, 77)
    params = gpu.SceneParams.make(0.0, 6.0, grid_width=1.0)
    cam = gpu.Camera.make(width=3, height=32768 * 16 + 37, projection=2, hfov=gpu.degrees_to_rads(20),
                          hang=gpu.degrees_to_rads(-45), vang=gpu.degrees_to_rads(110), pos=(-6.0, 6.0, 9.0),
                          step_dist=0.5, bg=(4, 5, 6))
    heights = oracle.update_heightmap(rgb, params)
    ofb, total, capped, *_ = oracle.render(oracle.make_cfg(cam, params, 40, 33), heights, cmap)
    assert capped == 0
    scene = gpu.Scene(rgb, cmap, params)
    fb, st, *_ = scene.render_stats(cam)
    assert st.rays == cam.width * cam.height and st.steps == total
    assert np.array_equal(fb, ofb) and np.array_equal(scene.render(cam), ofb)
    scene.close()


def test_pyramid_planes_beyond_4_gib_of_offsets(gpu, oracle):
    """VERDICT r04 #7 / ADVICE r04: a legal, very oblong map near the 2^29-cell limit (16385 x 32766) has pyramid planes 2^28 floats
    apart -- look-up offsets beyond 32 bits of bytes.  Round 4 rendered such scenes with the literal loop (85 x slower, and its
    shadow probe did not); the production kernel now forms the offset in 64 bits.  Rendering that map takes 537 M cells of
    host arrays, so the same plane pitch is FORCED on small maps (HMRM_DEBUG_PLANE_SHIFT=28: an 8 GiB pyramid, window indices up
    to 7 << 28): frames, per-ray step counts and distance() bits against the oracle, leaps really taken at every level."""
    with env(HMRM_DEBUG_PLANE_SHIFT=28):
        for case in scenes.cases()[:6]:
            name, rgb, cmap, params, cam = scenes.build_case(case)
            scene = gpu.Scene(rgb, cmap, params)
            heights = oracle.update_heightmap(rgb, params)
            ofb, total, capped, osteps, oentry = oracle.render(oracle.make_cfg(cam, params, rgb.shape[1], rgb.shape[0]), heights, cmap, per_pixel=True)
            fb, st, steps, entry = scene.render_stats(cam, per_pixel=True)
            assert np.array_equal(fb, ofb) and np.array_equal(steps.astype(np.int64), osteps) and np.array_equal(_bits(entry), _bits(oentry)), name
            assert np.array_equal(scene.render(cam), ofb), name
            scene.close()
        wl = gpu.synth.WORKLOADS["C2"]
        rgb, cmap = wl.maps()
        scene = gpu.Scene(rgb, cmap, wl.scene_params())
        cam = wl.camera()
        fb, st, steps, _ = scene.render_stats(cam, per_pixel=True)
        assert st.leaps > 1000000 and st.leaped_steps > 0.9 * st.steps and scene.kernel_choice() == 0
        assert np.array_equal(scene.render(cam), fb)
        scene.close()
    scene = gpu.Scene(rgb, cmap, wl.scene_params())  # (the ordinary layout: same frame, same counts)
    fb2, st2, steps2, _ = scene.render_stats(cam, per_pixel=True)
    assert np.array_equal(fb2, fb) and np.array_equal(steps2, steps) and (st2.leap_attempts, st2.leaps) == (st.leap_attempts, st.leaps)
    rows = list(range(0, cam.height, 24))
    heights = oracle.update_heightmap(rgb, wl.scene_params())
    cfg = oracle.make_cfg(cam, wl.scene_params(), wl.map_size, wl.map_size)
    for r in rows[:12]:
        ofb = np.zeros_like(fb)
        oracle.render(cfg, heights, cmap, rows=(r, r + 1), framebuf=ofb)
        assert np.array_equal(ofb[r], fb[r]), r
    scene.close()


def test_map_with_a_side_of_2_pow_24_cells(gpu, oracle):
    """The production kernel indexes cells with 24-bit multiplies; a strip map 2^24 cells long (the longest side the
    reference's image loader accepts) is routed through the literal kernel and still matches the oracle; the
    additive sampling modes refuse it by name."""
    import conftest
    conftest.skip_if_over_budget(10, "map with a side of 2^24 cells")
    w, h = 1 << 24, 1
    rng = np.random.RandomState(5)
    rgb = rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
    cmap = rng.randint(0, 256, size=(h, w, 4), dtype=np.uint8)
    cmap[..., 3] = 255
    params = gpu.SceneParams.make(0.0, 6.0, grid_width=1.0)
    cam = gpu.Camera.make(width=24, height=16, projection=1, hfov=gpu.degrees_to_rads(20), hang=0.0,
                          vang=gpu.degrees_to_rads(172), pos=(16000000.25, -0.5, 14.0), step_dist=0.25, bg=(9, 8, 7))
    heights = oracle.update_heightmap(rgb, params)
    ofb, total, capped, *_ = oracle.render(oracle.make_cfg(cam, params, w, h), heights, cmap)
    assert capped == 0 and total > 0
    scene = gpu.Scene(rgb, cmap, params)
    fb, st, *_ = scene.render_stats(cam)
    assert np.array_equal(fb, ofb) and st.steps == total and st.hits > 0
    assert st.leap_attempts == 0  # (the literal loop ran)
    with pytest.raises(gpu.HmrmError) as e:
        scene.render(_bilinear(cam))
    assert e.value.code == gpu.HMRM_E_ARG and "2^24" in str(e.value)
    scene.close()


def test_many_streams_round_robin_and_recycled_stream_state(gpu, oracle):
    """Frames in flight (bench.py's default): consecutive frames go round-robin to a dozen HIP streams, and then to
    forty -- more than the scene keeps launch state for, so states are recycled between launches.  Every frame of
    every stream equals the oracle's."""
    import torch
    rgb, cmap = scenes.small_maps(96, 96, 67)
    params = gpu.SceneParams.make(0.0, 10.0, grid_width=1.0)
    scene = gpu.Scene(rgb, cmap, params)
    heights = oracle.update_heightmap(rgb, params)
    W, H = 160, 90
    cams = [gpu.Camera.make(width=W, height=H, projection=2, hfov=gpu.degrees_to_rads(150), hang=gpu.degrees_to_rads(h),
                            vang=gpu.degrees_to_rads(112), pos=(-20.0, 20.0, 30.0), step_dist=0.25, bg=(1, 2, 3))
            for h in (-45, -30, -60)]
    want = [oracle.render(oracle.make_cfg(c, params, 96, 96), heights, cmap)[0] for c in cams]
    for nstreams in (12, 40):
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2 * nstreams)]
        torch.cuda.synchronize()
        for k in range(2 * nstreams):
            scene.render_rows_device(cams[k % 3], bufs[k].data_ptr(), W * 4, 0, H, stream=streams[k % nstreams].cuda_stream)
        torch.cuda.synchronize()
        for k in range(2 * nstreams):
            assert np.array_equal(bufs[k].cpu().numpy(), want[k % 3]), (nstreams, k)
    scene.close()
