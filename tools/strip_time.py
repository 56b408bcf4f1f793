"""Kernel time of row strips of the C3 frame: how long the heaviest rows take on their own (critical path)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hm = importlib.import_module("heightmap-ray-marcher_amd")
import torch
wl = hm.synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
rgb, cmap = hm.synth.synth_maps(wl.map_size)
scene = hm.Scene(rgb, cmap, wl.scene_params())
cam = wl.camera()
buf = torch.zeros((cam.height, cam.width, 4), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
def t(r0, r1, n=50):
    for _ in range(5):
        scene.render_rows_device(cam, buf.data_ptr(), cam.width * 4, r0, r1, stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        scene.render_rows_device(cam, buf.data_ptr(), cam.width * 4, r0, r1, stream=stream)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
H = cam.height
for r0, r1 in ((0, H), (0, 768), (768, 784), (784, 800), (768, 800), (760, 816), (800, H), (816, H), (1200, H)):
    print("rows %4d..%4d: %.4f ms" % (r0, r1, t(r0, r1)), flush=True)
