"""CPU-only checks of the oracle itself (oracle/hmrm_oracle.c).

The reference ships no tests or golden data for the render path and cannot be
built here (SDL2/glm absent), so the oracle is "parity unpinned" against the
reference.  What CAN be checked, and is:
  * the C oracle agrees bit for bit with a second, independent numpy restatement;
  * the -O0 (reference optimisation level) and -O2 builds agree;
  * hand-derivable properties of the reference's loop hold (camera inside the
    box never hits, sky formula, alpha-0 gives bg, min_height is added twice);
  * committed golden frames (tests/golden) still reproduce.
"""
import json
import os

import numpy as np
import pytest

import np_marcher
import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _oracle_frame(oracle, case, opt="O2", per_pixel=True):
    name, rgb, cmap, params, cam = scenes.build_case(case)
    heights = oracle.update_heightmap(rgb, params, opt=opt)
    cfg = oracle.make_cfg(cam, params, rgb.shape[1], rgb.shape[0])
    return oracle.render(cfg, heights, cmap, per_pixel=per_pixel, opt=opt), heights


@pytest.mark.parametrize("case", scenes.cases(), ids=scenes.case_ids())
def test_oracle_matches_numpy_restatement(oracle, hmrm, case):
    name, rgb, cmap, params, cam = scenes.build_case(case)
    (fb, total, capped, steps, entry), heights = _oracle_frame(oracle, case)
    assert capped == 0
    # UpdateHeightmap (hmap.cpp:171-191) in numpy
    r, g, b = (rgb[:, :, i].astype(np.float64) for i in range(3))
    value = np.clip((params.lum_r * r + params.lum_g * g) + params.lum_b * b, 0.0, 255.0)
    np_heights = (value / 255.0) * (params.max_height - params.min_height) + params.min_height
    assert np.array_equal(heights.view(np.uint64), np_heights.view(np.uint64))
    nfb, nsteps, ndist = np_marcher.render(cam, params, heights, cmap)
    assert np.array_equal(entry.view(np.uint64), ndist.view(np.uint64)), "distance() differs"
    assert np.array_equal(steps, nsteps), "per-ray step counts differ"
    assert np.array_equal(fb, nfb), "frames differ"
    assert total == int(nsteps.sum())


@pytest.mark.parametrize("case", scenes.cases(), ids=scenes.case_ids())
def test_oracle_O0_equals_O2(oracle, case):
    (fb2, t2, c2, s2, e2), h2 = _oracle_frame(oracle, case, "O2")
    (fb0, t0, c0, s0, e0), h0 = _oracle_frame(oracle, case, "O0")
    assert np.array_equal(h0.view(np.uint64), h2.view(np.uint64))
    assert np.array_equal(fb0, fb2) and np.array_equal(s0, s2) and t0 == t2
    assert np.array_equal(e0.view(np.uint64), e2.view(np.uint64))


def test_camera_inside_box_never_hits(oracle, hmrm):
    # AABB.cpp:38-40: d < 0 -> no intersection -> sky or background only
    for case in scenes.cases():
        if not case[0].endswith("_inside"):
            continue
        (fb, total, capped, steps, entry), _ = _oracle_frame(oracle, case)
        assert total == 0 and (steps == 0).all()
        assert ((entry < 0) | np.isinf(entry)).all()


def test_sky_formula(oracle, hmrm):
    # hmap.cpp:1041-1057 on a spherical camera looking up: dir.z = cos(va) per row
    import math
    rgb, cmap = scenes.small_maps(16, 16, 1)
    params = hmrm.SceneParams.make(0.0, 1.0, grid_width=1.0)
    cam = hmrm.Camera.make(width=8, height=16, projection=2, hfov=hmrm.degrees_to_rads(60), hang=0.3,
                           vang=hmrm.degrees_to_rads(40), pos=(100.0, 100.0, 50.0), step_dist=0.5, bg=(10, 250, 3))
    heights = oracle.update_heightmap(rgb, params)
    fb, total, *_ = oracle.render(oracle.make_cfg(cam, params, 16, 16), heights, cmap)
    assert total == 0
    ar = 8 / 16
    vfov = cam.hfov / ar
    ul_vang = cam.vang - vfov / 2.0
    for row in range(16):
        va = ul_vang + (row / 15) * vfov
        z = math.cos(va)
        if z > 0.0:
            exp = [math.floor(min(max(220.0 * (z * z) + 10, 0.0), 255.0)),
                   math.floor(min(max(240.0 * (z * z) + 250, 0.0), 255.0)),
                   math.floor(min(max(255.0 * z + 3, 0.0), 255.0)), 255]
        else:
            exp = [10, 250, 3, 255]
        assert fb[row, 0].tolist() == exp, row


def test_alpha_zero_draws_background_and_min_height_added_twice(oracle, hmrm):
    # top-down orthographic over a flat map: every ray hits the cell below it.
    rgb = np.full((8, 8, 3), 255, dtype=np.uint8)          # value 255 -> height == max_height
    cmap = np.zeros((8, 8, 4), dtype=np.uint8)
    cmap[:, :, 0] = np.arange(8)[None, :] * 10
    cmap[:, :, 3] = 255
    cmap[2, 3, 3] = 0                                       # hmap.cpp:1020
    params = hmrm.SceneParams.make(2.0, 5.0, grid_width=1.0)
    cam = hmrm.Camera.make(width=8, height=8, projection=3, hang=0.0, vang=hmrm.degrees_to_rads(180),
                           pos=(4.0, -4.0, 100.0), ortho_width=0.97, step_dist=0.25, bg=(1, 2, 3))
    heights = oracle.update_heightmap(rgb, params)
    assert (heights == 5.0).all()
    fb, total, capped, steps, entry = oracle.render(oracle.make_cfg(cam, params, 8, 8), heights, cmap, per_pixel=True)
    # rays enter the box top at z = max_height = 5 (minus the nudge) and compare z < 5 + 2:
    # hit on the very first sample although the box is only 3 high (min_height counted twice, hmap.cpp:1016)
    inside = steps > 0
    assert inside.any() and (steps[inside] == 1).all()
    assert (fb[~inside][:, :3] == (1, 2, 3)).all()
    hole = (fb[:, :, 0] == 1) & (fb[:, :, 1] == 2) & (fb[:, :, 2] == 3) & inside
    assert hole.any(), "the alpha-0 texel must render as bg_color"
    assert (fb[:, :, 3] == 255).all()


def test_golden_frames_reproduce(oracle):
    """tests/golden/frames.npz was produced by tests/golden/make_golden.py from THIS oracle
    (no reference output exists to pin against); it guards against drift."""
    path = os.path.join(GOLDEN, "frames.npz")
    data = np.load(path)
    for case in scenes.cases():
        (fb, total, capped, steps, entry), _ = _oracle_frame(oracle, case)
        name = case[0]
        assert np.array_equal(fb, data[name + "/frame"]), name
        assert np.array_equal(steps.astype(np.uint32), data[name + "/steps"]), name
        assert np.array_equal(entry.view(np.uint64), data[name + "/entry_bits"]), name


# ---- bilinear quality mode (additive; the oracle is its definition) ----

def test_bilinear_mode_definition(oracle, hmrm):
    """Top-down orthographic rays make the mode checkable by hand: each ray hits at a known (x, y),
    so its colour must be the bilinear mix of the four surrounding texels (cell centres at
    integer + 0.5, edges clamped), rounded with floor(f + 0.5); the step at which it hits is the
    first sample with z below the interpolated height (+ min_height a second time, hmap.cpp:1016)."""
    import math
    rng = np.random.RandomState(5)
    mw = mh = 12
    rgb = np.repeat(rng.randint(0, 256, size=(mh, mw, 1)), 3, axis=2).astype(np.uint8)
    cmap = rng.randint(0, 256, size=(mh, mw, 4)).astype(np.uint8)
    cmap[:, :, 3] = 255
    cmap[4, 7, 3] = 0
    params = hmrm.SceneParams.make(0.0, 6.0, grid_width=1.0)
    kw = dict(width=31, height=29, projection=3, hang=0.0, vang=hmrm.degrees_to_rads(180), pos=(6.0, -6.0, 50.0),
              ortho_width=0.41, step_dist=0.125, bg=(9, 8, 7))
    cam = hmrm.Camera.make(sampling=hmrm.BILINEAR, **kw)
    heights = oracle.update_heightmap(rgb, params)
    fb, total, capped, steps, entry = oracle.render(oracle.make_cfg(cam, params, mw, mh), heights, cmap, per_pixel=True)
    checked = 0
    for py in range(cam.height):
        for px in range(cam.width):
            ray = oracle.probe_ray(oracle.make_cfg(cam, params, mw, mh), px, py)
            pos, dirv, d = ray
            if not (d >= 0 and math.isfinite(d)):
                continue
            assert abs(dirv[2] + 1.0) < 1e-6, "camera must look straight down"
            qx, qy = pos[0], -pos[1]
            if not (0.0 < qx < mw and 0.0 < qy < mh):
                continue
            u, v = qx - 0.5, qy - 0.5
            fu, fv = math.floor(u), math.floor(v)
            tx, ty = u - fu, v - fv
            i0, i1 = min(max(int(fu), 0), mw - 1), min(max(int(fu) + 1, 0), mw - 1)
            j0, j1 = min(max(int(fv), 0), mh - 1), min(max(int(fv) + 1, 0), mh - 1)

            def mix(a00, a10, a01, a11):
                a = a00 + tx * (a10 - a00)
                c = a01 + tx * (a11 - a01)
                return a + ty * (c - a)
            if cmap[int(qy), int(qx), 3] == 0:
                exp = [9, 8, 7, 255]
            else:
                exp = [int(math.floor(min(max(mix(*(float(cmap[j, i, k]) for j, i in ((j0, i0), (j0, i1), (j1, i0), (j1, i1)))) + 0.5, 0.0), 255.0)))
                       for k in range(3)] + [255]
            assert fb[py, px].tolist() == exp, (px, py)
            t = mix(*(heights[j, i] + params.min_height for j, i in ((j0, i0), (j0, i1), (j1, i0), (j1, i1))))
            # straight down: z after k steps (the first sample is at the nudged entry point)
            z0 = 6.0 - 0.01
            k_hit = max(1, int(math.ceil((z0 - t) / 0.125 + 1e-9)) + 1) if z0 >= t else 1
            assert abs(int(steps[py, px]) - k_hit) <= 1, (px, py, steps[py, px], k_hit)
            checked += 1
    assert checked > 300


def test_bilinear_on_flat_map_equals_nearest_and_config_flag_defaults_off(oracle, hmrm):
    rgb = np.full((16, 16, 3), 90, dtype=np.uint8)
    cmap = np.full((16, 16, 4), 200, dtype=np.uint8)
    params = hmrm.SceneParams.make(0.0, 3.0, grid_width=0.5)
    heights = oracle.update_heightmap(rgb, params)
    frames = []
    for sampling in (hmrm.NEAREST, hmrm.BILINEAR):
        cam = hmrm.Camera.make(width=40, height=30, projection=1, hang=hmrm.degrees_to_rads(-40),
                               vang=hmrm.degrees_to_rads(115), pos=(-3.0, 3.0, 4.0), step_dist=0.05, sampling=sampling)
        cfg = oracle.make_cfg(cam, params, 16, 16)
        assert cfg.sampling == sampling
        frames.append(oracle.render(cfg, heights, cmap, per_pixel=True))
    assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][3], frames[1][3])
    assert hmrm.Camera.make().sampling == hmrm.NEAREST


def test_glm_sensitivity_switch(oracle, hmrm):
    """The oracle's glm formulas are restated, not pinned.  The switch that swaps in the
    alternatives (tests/glm_sensitivity.py, profiles/r02_glm_sensitivity.txt) changes last bits of ray
    directions but, on this frame, no pixel; and it is off by default."""
    import scenes
    rgb, cmap = scenes.small_maps(64, 64, 3)
    params = hmrm.SceneParams.make(0.0, 8.0, grid_width=1.0)
    cam = hmrm.Camera.make(width=96, height=54, projection=1, hfov=hmrm.degrees_to_rads(90), hang=hmrm.degrees_to_rads(-45),
                           vang=hmrm.degrees_to_rads(118), pos=(-10.0, 10.0, 24.0), step_dist=0.5)
    heights = oracle.update_heightmap(rgb, params)
    cfg = oracle.make_cfg(cam, params, 64, 64)
    fb0, total0, _, steps0, entry0 = oracle.render(cfg, heights, cmap, per_pixel=True)
    try:
        changed_bits = 0
        for v in (1, 2, 3):
            oracle.set_glm_variant(v)
            fb, total, _, steps, entry = oracle.render(cfg, heights, cmap, per_pixel=True)
            changed_bits += int((entry.view(np.uint64) != entry0.view(np.uint64)).sum())
            assert int((fb != fb0).any(axis=2).sum()) <= 2 and abs(total - total0) <= 8
        assert changed_bits > 0, "the switch must reach the ray generator"
    finally:
        oracle.set_glm_variant(0)
    fb, total, *_ = oracle.render(cfg, heights, cmap, per_pixel=True)
    assert np.array_equal(fb, fb0) and total == total0
