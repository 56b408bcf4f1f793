"""Config grammar = ConsumeConfigStream (main/hmap.cpp:309-520), CPU only."""
import math
import os
import subprocess

import numpy as np
import pytest


@pytest.fixture()
def maps(hmrm, tmp_path):
    rng = np.random.RandomState(0)
    h = rng.randint(0, 256, size=(6, 8, 3)).astype(np.uint8)
    c = rng.randint(0, 256, size=(6, 8, 4)).astype(np.uint8)
    hp, cp = str(tmp_path / "h.ppm"), str(tmp_path / "c.png")
    hmrm.write_ppm(hp, h)
    hmrm.write_png(cp, c)
    return hp, cp, h, c


def test_defaults_match_reference_globals(hmrm, maps):
    hp, cp, h, c = maps
    cfg = hmrm.Config().consume_string(f"heightmap {hp}\ncolormap {cp}\n")
    cam, sp = cfg.camera(), cfg.scene_params()
    # main/hmap.cpp:31-112
    assert (cam.width, cam.height) == (800, 600)
    assert cam.hfov == math.pi / 2.0 and cam.hang == -math.pi / 4.0 and cam.vang == math.pi / 2.0
    assert list(cam.pos) == [-5.0, 5.0, 0.0]
    assert (sp.min_height, sp.max_height) == (0.0, 10.0)
    assert (sp.lum_r, sp.lum_g, sp.lum_b) == (0.299, 0.587, 0.114)
    assert sp.grid_width == 0.05 and cam.step_dist == 5.0 * 0.05 and cam.ortho_width == 2.0 * 0.05
    assert (cam.bg_r, cam.bg_g, cam.bg_b) == (0, 0, 0)
    assert cfg.cycle == 47 and cfg.recording_frame_count == 200
    assert cam.projection == hmrm.PERSPECTIVE
    assert np.array_equal(cfg.height_rgb(), h)
    assert np.array_equal(cfg.color_rgba(), c)
    assert cfg.take_heightmap_dirty() and not cfg.take_heightmap_dirty()
    assert cfg.log == f"heightmap {hp}\ncolormap {cp}\n"


def test_every_key_and_echo(hmrm, maps):
    hp, cp, *_ = maps
    text = f"""
    resolution 320 240 hfov 60 hang 30 vang 100
    pos 1 2 3 pos_x 1.5 pos_y -2.5 pos_z 7
    min_height -1 max_height 12.5 lum 0.1 0.2 0.3 lum_norm 1 2 1 lum_r 0.5 lum_g 0.25 lum_b 0.125
    grid_width 0.5 ortho_width 0.75 step_dist 0.125 bg_color 300 20 10 cycle 1
    mouse_sens 0.5 scroll_sens 2 move 0.01 recording_frame_count 64
    heightmap {hp} colormap {cp} projection spherical output out.ppm
    """
    cfg = hmrm.Config().consume_string(text)
    cam, sp = cfg.camera(), cfg.scene_params()
    assert (cam.width, cam.height) == (320, 240)
    # DegreesToRads, hmap.cpp:131-133: (deg / 180.0) * M_PI
    assert cam.hfov == (60 / 180.0) * math.pi and cam.hang == (30 / 180.0) * math.pi and cam.vang == (100 / 180.0) * math.pi
    assert list(cam.pos) == [1.5, -2.5, 7.0]
    assert (sp.min_height, sp.max_height) == (-1.0, 12.5)
    assert (sp.lum_r, sp.lum_g, sp.lum_b) == (0.5, 0.25, 0.125)
    assert (sp.grid_width, cam.ortho_width, cam.step_dist) == (0.5, 0.75, 0.125)
    assert (cam.bg_r, cam.bg_g, cam.bg_b) == (300 & 255, 20, 10)  # (Uint8)r, hmap.cpp:459
    assert cfg.cycle == 1 and cfg.recording_frame_count == 64
    assert cam.projection == hmrm.SPHERICAL and cfg.output_path == "out.ppm"
    log = cfg.log.splitlines()
    assert log[0] == "resolution 320 240" and log[1] == "hfov 60" and log[2] == "hang 30" and log[3] == "vang 100"
    assert "pos 1 2 3" in log and "pos_x 1.5" in log and "lum 0.25 0.5 0.25" in log  # lum_norm, hmap.cpp:416-425
    assert "bg_color 44 20 10" in log and "cycle 1" in log and "move 0.01" in log
    assert cfg.warnings == ""


def test_sample_config_quirk_cycle_bits(hmrm, maps):
    """sample_config.txt:9 has `cycle_bits 6`, which the parser does not know: two
    warnings ("cycle_bits", "6") and cycle stays 47 (hmap.cpp:465-469,486-488)."""
    hp, cp, *_ = maps
    text = ("resolution 800 450\nhfov 90\nmin_height 0.0\nmax_height 10.0\ngrid_width 0.01\northo_width 0.1\n"
            f"step_dist 0.05\nbg_color 0 0 0\ncycle_bits 6\nmouse_sens 0.00003\nheightmap {hp}\ncolormap {cp}\n")
    cfg = hmrm.Config().consume_string(text)
    assert cfg.warnings == "WARNING: Unknown identifier: cycle_bits\nWARNING: Unknown identifier: 6\n"
    assert cfg.cycle == 47
    assert cfg.camera().width == 800 and cfg.camera().height == 450
    assert cfg.scene_params().grid_width == 0.01 and cfg.camera().step_dist == 0.05
    assert "mouse_sens 3e-05" in cfg.log


def test_print_dumps_all_options(hmrm, maps):
    hp, cp, *_ = maps
    cfg = hmrm.Config().consume_string(f"heightmap {hp} colormap {cp} print")
    tail = cfg.log.split("print\n", 1)[1].splitlines()
    assert tail == [f"heightmap {hp}", f"colormap {cp}", "resolution 800 600", "hfov 90", "hang -45", "vang 90",
                    "pos -5 5 0", "min_height 0", "max_height 10", "lum 0.299 0.587 0.114", "grid_width 0.05",
                    "ortho_width 0.1", "step_dist 0.25", "bg_color 0 0 0", "cycle 47", "mouse_sens 1",
                    "scroll_sens 1", "move 0.05", "recording_frame_count 200"]


def test_validation_errors(hmrm, maps, tmp_path):
    hp, cp, h, c = maps
    with pytest.raises(hmrm.HmrmError) as e:
        hmrm.Config().consume_string(f"colormap {cp}")
    assert e.value.code == hmrm.HMRM_E_CONFIG and e.value.message == "Must specify heightmap in config"
    with pytest.raises(hmrm.HmrmError) as e:
        hmrm.Config().consume_string(f"heightmap {hp}")
    assert e.value.message == "Must specify colormap in config"
    other = str(tmp_path / "c2.png")
    hmrm.write_png(other, np.zeros((5, 8, 4), dtype=np.uint8))
    with pytest.raises(hmrm.HmrmError) as e:
        hmrm.Config().consume_string(f"heightmap {hp} colormap {other}")
    assert e.value.message == "heightmap dimensions (8x6) must match colormap dimensions (8x5)"
    with pytest.raises(hmrm.HmrmError) as e:
        hmrm.Config().consume_string("heightmap /nonexistent/h.png")
    assert e.value.code == hmrm.HMRM_E_IMAGE
    assert e.value.message.startswith("Failed to load image for heightmap from /nonexistent/h.png")
    with pytest.raises(hmrm.HmrmError) as e:
        hmrm.Config().consume_file(str(tmp_path / "nope.txt"))
    assert e.value.code == hmrm.HMRM_E_IO and e.value.message.startswith("Failed to open input file: ")


def test_malformed_number_ends_the_stream_like_iostream(hmrm, maps):
    """`input >> double` on a non-number sets failbit: the token loop ends there
    (hmap.cpp:313), later keys are never seen, validation still runs."""
    hp, cp, *_ = maps
    cfg = hmrm.Config().consume_string(f"heightmap {hp} colormap {cp} step_dist abc resolution 10 10")
    assert cfg.camera().width == 800
    cfg2 = hmrm.Config()
    with pytest.raises(hmrm.HmrmError):
        cfg2.consume_string(f"hfov oops heightmap {hp} colormap {cp}")


def test_a_key_at_the_very_end_of_a_stream_reads_zeros_not_garbage(hmrm, maps):
    """`input >> x` at end of stream extracts nothing and leaves x alone: the reference's `bg_color`, `lum` and the angle
    keys then use an uninitialised local (hmap.cpp:367-384, :417-427, :456-461).  The library's locals start at zero, so the
    outcome is the one of a malformed number: defined, and the same on every run."""
    hp, cp, *_ = maps
    for _ in range(3):
        cam = hmrm.Config().consume_string(f"heightmap {hp} colormap {cp} bg_color 7 8 9 bg_color").camera()
        assert (cam.bg_r, cam.bg_g, cam.bg_b) == (0, 0, 0)
        cam = hmrm.Config().consume_string(f"heightmap {hp} colormap {cp} hang 30 hang").camera()
        assert cam.hang == 0.0
        cfg = hmrm.Config().consume_string(f"heightmap {hp} colormap {cp} lum 1 2")
        p = cfg.scene_params()
        assert (p.lum_r, p.lum_g, p.lum_b) == (1.0, 2.0, 0.0)


def test_later_keys_win_and_streams_accumulate(hmrm, maps):
    hp, cp, *_ = maps
    cfg = hmrm.Config().consume_string(f"heightmap {hp} colormap {cp} step_dist 1 step_dist 2")
    assert cfg.camera().step_dist == 2.0
    cfg.take_heightmap_dirty()
    cfg.consume_string("max_height 3")  # the runtime console re-feeds the same parser (hmap.cpp:794-803)
    assert cfg.scene_params().max_height == 3.0 and cfg.take_heightmap_dirty()
    cfg.consume_string("hang 10")
    assert not cfg.take_heightmap_dirty()


def test_cli_usage_and_errors(hmrm, tmp_path):
    exe = os.path.join(os.path.dirname(hmrm.LIB_PATH), "hmap")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr == "USAGE: hmap.exe path/to/config.txt\n"  # hmap.cpp:527-530
    r = subprocess.run([exe, str(tmp_path / "missing.txt")], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("Failed to open input file: ")
    cfg = tmp_path / "c.txt"
    cfg.write_text("resolution 4 4\n")
    r = subprocess.run([exe, str(cfg)], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout == "resolution 4 4\n" and r.stderr == "Must specify heightmap in config\n"


def test_record_key_and_orbit_camera(hmrm, maps):
    """Additive `record orbit` key + the orbit sweep of BASELINE config C5 (hmrm_orbit_camera)."""
    hp, cp, *_ = maps
    cfg = hmrm.Config().consume_string(f"heightmap {hp} colormap {cp} record orbit recording_frame_count 64")
    assert cfg.record_mode == 1 and cfg.recording_frame_count == 64 and "record orbit" in cfg.log
    assert hmrm.Config().consume_string(f"heightmap {hp} colormap {cp}").record_mode == 0
    # additive `devices n`: GPUs the recording is sharded over (frame k on device k mod n); default 1, 0 = all
    assert hmrm.Config().consume_string(f"heightmap {hp} colormap {cp}").devices == 1
    cfg8 = hmrm.Config().consume_string(f"heightmap {hp} colormap {cp} devices 8 record orbit")
    assert cfg8.devices == 8 and "devices 8" in cfg8.log
    assert hmrm.Config().consume_string(f"heightmap {hp} colormap {cp} devices 0").devices == 0
    # additive `heights f32`: float hit thresholds (camera.sampling = HMRM_NEAREST_F32); f64 is the default
    c32 = hmrm.Config().consume_string(f"heightmap {hp} colormap {cp} heights f32")
    assert c32.camera().sampling == hmrm.NEAREST_F32 and "heights f32" in c32.log
    assert hmrm.Config().consume_string(f"heightmap {hp} colormap {cp} heights f32 heights f64").camera().sampling == hmrm.NEAREST
    assert "Unknown heights type" in hmrm.Config().consume_string(f"heightmap {hp} colormap {cp} heights f16").warnings
    wl = hmrm.synth.WORKLOADS["C5"]
    s = float(wl.map_size)
    static = wl.camera()
    for k in (0, 1, 16, 63):
        cam = wl.camera(k, 64)
        # on the circle of radius 0.9*S around the map centre (S/2, -S/2), same height
        dx, dy = cam.pos[0] - s / 2.0, cam.pos[1] + s / 2.0
        assert abs(math.hypot(dx, dy) - 0.9 * s) < 1e-9 * s and cam.pos[2] == static.pos[2]
        # looking at the centre: (cos hang, sin hang) points from the camera to the centre
        assert abs(math.cos(cam.hang) * 0.9 * s + dx) < 1e-9 * s and abs(math.sin(cam.hang) * 0.9 * s + dy) < 1e-9 * s
        assert cam.hang == hmrm.degrees_to_rads(-45.0) + (2.0 * math.pi * k) / 64.0
        assert (cam.width, cam.height, cam.vang, cam.step_dist) == (static.width, static.height, static.vang, static.step_dist)
    # frame 0 is the static pose up to the rounding of 0.9*S*cos(45 deg) vs S/8 + S/2
    c0 = wl.camera(0, 64)
    assert abs(c0.pos[0] - static.pos[0]) < 0.02 * s and abs(c0.pos[1] - static.pos[1]) < 0.02 * s


def test_sampling_key(hmrm, maps):
    """Additive `sampling nearest|bilinear` key (quality mode; the reference always takes the nearest cell)."""
    hp, cp, *_ = maps
    base = f"heightmap {hp} colormap {cp} "
    assert hmrm.Config().consume_string(base).camera().sampling == hmrm.NEAREST
    cfg = hmrm.Config().consume_string(base + "sampling bilinear")
    assert cfg.camera().sampling == hmrm.BILINEAR and "sampling bilinear\n" in cfg.log
    cfg = hmrm.Config().consume_string(base + "sampling bilinear sampling nearest")
    assert cfg.camera().sampling == hmrm.NEAREST
    cfg = hmrm.Config().consume_string(base + "sampling cubic")
    assert cfg.camera().sampling == hmrm.NEAREST and "WARNING: Unknown sampling: cubic" in cfg.warnings
