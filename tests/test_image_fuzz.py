"""ASan/UBSan mutation fuzz of the native image decoders (PNG, JPEG, BMP, TGA, PNM, GIF, PSD, PIC, HDR) and the PNG
encoder round trip, CPU only.  Maps come from users' files: a malformed one may be refused, it
must not corrupt memory."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "heightmap-ray-marcher_amd", "csrc")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_decoders_survive_mutated_files(tmp_path):
    exe = str(tmp_path / "fuzz_image_io")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fwrapv", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + SRC, os.path.join(ROOT, "tests", "fuzz_image_io.cpp"), os.path.join(SRC, "image_io.cpp"),
           os.path.join(SRC, "jpeg_decode.cpp"), os.path.join(SRC, "bmp_tga_decode.cpp"), os.path.join(SRC, "legacy_formats.cpp"),
           "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("no sanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stderr
    files = []
    for f in ("png_decode", "jpeg_decode", "bmp_tga_decode", "legacy_decode"):
        d = np.load(os.path.join(GOLDEN, f + ".npz"))
        for k in d.files:
            if k.endswith("/bytes"):
                p = tmp_path / f"{f}_{k.split('/')[0]}.bin"
                p.write_bytes(d[k].tobytes())
                files.append(str(p))
    r = subprocess.run([exe, "40"] + files, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "png roundtrips ok" in r.stdout
