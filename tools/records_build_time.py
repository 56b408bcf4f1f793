"""Scene set-up kernels under rocprofv3 (how long the window records take to build beside the pyramid):
  rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/records_build_time.py [map size]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hmrm = importlib.import_module("heightmap-ray-marcher_amd")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for kind in ("smooth", "needles", "white"):
    v = hmrm.synth.content_heights_u8(size, kind)
    import numpy as np
    rgb = np.ascontiguousarray(np.repeat(v[:, :, None], 3, axis=2))
    cmap = np.zeros((size, size, 4), dtype=np.uint8)
    scene = hmrm.Scene(rgb, cmap, hmrm.synth.WORKLOADS["C3"].scene_params())
    for _ in range(3):
        scene.update(hmrm.synth.WORKLOADS["C3"].scene_params())
        scene.read_records()  # (the records are built on demand: api.cpp ensure_records)
    scene.close()
print("done")
