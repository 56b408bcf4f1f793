"""ASan/UBSan fuzz of the host-only logic of the render path, CPU only (tests/fuzz_host_logic.cpp): the config grammar on
hostile streams, frame records and row-cost estimates of degenerate cameras, launch orders made from untrusted measurement
records (every order must hand each tile row to exactly one grid row) and the calibration's state machine under random
event sequences.  The files are compiled with g++ as they are -- none of them contains device code."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "heightmap-ray-marcher_amd", "csrc")


def test_host_logic_survives_hostile_input_under_sanitizers(tmp_path):
    exe = str(tmp_path / "fuzz_host_logic")
    sources = ["config.cpp", "camera.cpp", "row_cost.cpp", "launch_order.cpp", "image_io.cpp", "jpeg_decode.cpp",
               "bmp_tga_decode.cpp", "legacy_formats.cpp"]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fwrapv", "-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=undefined,float-cast-overflow",
           "-I" + SRC, os.path.join(ROOT, "tests", "fuzz_host_logic.cpp")] + [os.path.join(SRC, s) for s in sources] + ["-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("no sanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stderr
    maps = tmp_path / "maps"
    maps.mkdir()
    r = subprocess.run([exe, "6000", str(maps)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.startswith("host logic ok: 6000 config streams")
    accepted = int(r.stdout.split("(")[1].split()[0])
    assert accepted > 50  # (streams that name both maps and survive to the end: the grammar's deep rows were reached)
