"""Host-side per-frame set-up (csrc/camera.cpp) against the oracle, CPU only:
camera basis, image plane and box corners of main/hmap.cpp:661-672,:952-974."""
import numpy as np
import pytest

import scenes


def _bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("case", scenes.cases(), ids=scenes.case_ids())
def test_frame_record_matches_oracle(hmrm, oracle, case):
    name, mw, mh, seed, params, cam = case
    rec = hmrm.debug_frame(cam, params, mw, mh)
    cfg = oracle.make_cfg(cam, params, mw, mh)
    o = oracle.frame_record(cfg)
    assert np.array_equal(_bits(rec["c0"]), _bits(o[15:18]))
    assert np.array_equal(_bits(rec["c1"]), _bits(o[18:21]))
    assert _bits(rec["nudge"]) == _bits(o[21])
    assert np.array_equal(_bits(rec["cam"]), _bits(o[0:3]))
    if cam.projection in (1, 3):
        assert np.array_equal(_bits(rec["upper_left"]), _bits(o[3:6]))
        assert np.array_equal(_bits(rec["plane_right"]), _bits(o[6:9]))
        assert np.array_equal(_bits(rec["plane_down"]), _bits(o[9:12]))
    if cam.projection == 3:
        assert np.array_equal(_bits(rec["look"]), _bits(o[12:15]))
        # float round trip (Orthographic.cpp:3): every component is exactly a float
        assert np.array_equal(rec["look"], rec["look"].astype(np.float32).astype(np.float64))
    if cam.projection == 2:
        # separable tables reproduce Spherical::GetRay (Spherical.cpp:18-25) for every pixel
        W, H = cam.width, cam.height
        with np.errstate(all="ignore"):
            dx = rec["row_sin_va"][:, None] * rec["col_cos_ha"][None, :]
            dy = rec["row_sin_va"][:, None] * rec["col_sin_ha"][None, :]
            dz = np.repeat(rec["row_cos_va"][:, None], W, 1)
        for py in range(0, H, max(1, H // 7)):
            for px in range(0, W, max(1, W // 7)):
                pos, d, dist = oracle.probe_ray(cfg, px, py)
                got = np.array([dx[py, px], dy[py, px], dz[py, px]])
                assert np.array_equal(_bits(got), _bits(d)), (px, py)


def test_grid_pow2_flag(hmrm):
    cam = hmrm.Camera.make(width=4, height=4)
    for gw, exp in ((1.0, 1), (0.5, 1), (2.0 ** -20, 1), (1024.0, 1), (0.05, 0), (3.0, 0), (0.75, 0)):
        rec = hmrm.debug_frame(cam, hmrm.SceneParams.make(grid_width=gw), 8, 8)
        assert rec["grid_pow2"] == exp, gw
        if exp:
            assert rec["inv_grid_width"] == 1.0 / gw


def test_spherical_tables_of_a_4k_camera_filled_on_the_host_pool(hmrm, oracle):
    """A 3840x2160 spherical camera (BASELINE config C3's) has 12 000 table entries: the library fills them in
    pieces on its host threads (csrc/host_pool.cpp); every entry must be the glibc value the reference's
    per-pixel sin / cos calls produce (Spherical.cpp:18-25), whichever thread wrote it -- checked for every row
    and column against the oracle's rays along the frame's diagonal, its first row and its first column, for
    several cameras in a row (the pool is reused) and from two caller threads at once."""
    import threading
    wl = hmrm.synth.WORKLOADS["C3"]
    params = wl.scene_params()

    def check(k):
        cam = wl.camera(k, 64)
        rec = hmrm.debug_frame(cam, params, wl.map_size, wl.map_size)
        cfg = oracle.make_cfg(cam, params, wl.map_size, wl.map_size)
        W, H = cam.width, cam.height
        for px in range(0, W, 7):
            _, d, _ = oracle.probe_ray(cfg, px, 1080)
            got = [rec["row_sin_va"][1080] * rec["col_cos_ha"][px], rec["row_sin_va"][1080] * rec["col_sin_ha"][px]]
            assert np.array_equal(_bits(got), _bits(d[:2])), (k, px)
        for py in range(0, H, 5):
            _, d, _ = oracle.probe_ray(cfg, 17, py)
            got = [rec["row_sin_va"][py] * rec["col_cos_ha"][17], rec["row_sin_va"][py] * rec["col_sin_ha"][17], rec["row_cos_va"][py]]
            assert np.array_equal(_bits(got), _bits(d)), (k, py)
        for px in (0, 1023, 1024, 1499, 1500, 2999, 3000, 3839):  # piece boundaries of 2, 3 and 4 threads and the ends
            _, d, _ = oracle.probe_ray(cfg, px, 0)
            got = [rec["row_sin_va"][0] * rec["col_cos_ha"][px], rec["row_sin_va"][0] * rec["col_sin_ha"][px]]
            assert np.array_equal(_bits(got), _bits(d[:2])), (k, px)

    for k in range(4):
        check(k)
    errors = []

    def worker(ks):
        try:
            for k in ks:
                check(k)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    ts = [threading.Thread(target=worker, args=(range(a, a + 3),)) for a in (10, 20)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
