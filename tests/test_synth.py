"""The synthetic inputs are pinned: integer arithmetic on seeded hashes, the same bytes on every machine and whatever
the number of threads that fill the rows (bench numbers and the committed PMC profiles refer to these maps)."""
import hashlib

import numpy as np
import pytest


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


PINNED_512 = {"smooth": "da5333e5d9df68bf", "white": "c6510e35544481a9", "spikes": "7afb165e4bb48d7a",
              "needles": "b3fe196695359d19", "canyon": "1a16ce8a345aea8b"}


@pytest.mark.parametrize("kind", sorted(PINNED_512))
def test_content_maps_are_pinned(hmrm, kind):
    v = hmrm.synth.content_heights_u8(512, kind)
    assert v.shape == (512, 512) and v.dtype == np.uint8
    assert _sha(v) == PINNED_512[kind]
    rgb, cmap = hmrm.synth.content_maps(512, kind)
    assert np.array_equal(rgb[:, :, 0], v) and np.array_equal(rgb[:, :, 1], v) and (cmap[:, :, 3] == 255).all()


def test_bench_maps_hash_is_the_one_in_the_committed_bench_lines(hmrm):
    rgb, cmap = hmrm.synth.synth_maps(1024)
    assert hmrm.synth.maps_sha256(rgb, cmap)[:16] == "34d26a2efe18439e"


def test_threaded_fill_equals_single_block(hmrm):
    s = hmrm.synth
    total = sum(65535 >> o for o in range(6))
    one = s._noise_rows(300, s.SEED, 6, total, 0, 300)
    assert np.array_equal(s.value_noise_u8(300), one)


def test_hostile_content_has_the_intended_structure(hmrm):
    s = hmrm.synth
    sp, sm = s.content_heights_u8(1024, "spikes"), s.content_heights_u8(1024, "smooth")
    assert (sp != sm).sum() == 16 and (sp == 255).sum() == 16  # one spike per 256 x 256 block
    for by in range(4):
        for bx in range(4):
            assert (sp[by * 256:(by + 1) * 256, bx * 256:(bx + 1) * 256] == 255).sum() == 1
    nd = s.content_heights_u8(1024, "needles")
    assert set(np.unique(nd)) == {40, 255} and 0.01 < (nd == 255).mean() < 0.02
    cy = s.content_heights_u8(1024, "canyon")
    ys, xs = np.mgrid[0:1024, 0:1024]
    assert (cy[np.abs(xs - ys) >= 48] >= 180).all() and (cy[np.abs(xs - ys) < 48] <= 31).all()
    w = s.content_workload("C3", "canyon")
    assert w.camera().pos[2] < w.scene_params().max_height * 180 / 255  # below the raised terrain, outside the box
    assert w.camera().pos[0] < 0 and w.camera().pos[1] > 0


def test_grid_workloads_see_the_same_cells(hmrm):
    s = hmrm.synth
    a, b = s.WORKLOADS["C5"], s.grid_workload("C5", 0.05)
    ca, cb = a.camera(), b.camera()
    assert abs(cb.pos[0] / 0.05 - ca.pos[0]) < 1e-9 and abs(cb.step_dist / 0.05 - ca.step_dist) < 1e-12
    assert abs(b.scene_params().max_height / 0.05 - a.scene_params().max_height) < 1e-9
    r = s.WORKLOADS["REFDEF"]
    assert r.scene_params().grid_width == 0.01 and abs(r.camera().step_dist - 0.05) < 1e-15  # sample_config.txt:5-7
