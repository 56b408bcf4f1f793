"""ctypes access to oracle/_ref/libstb_ref.so = the reference's vendored
stb_image v2.27 + stb_image_write v1.16 compiled from /root/reference/vendor by
oracle/Makefile.  Test infrastructure only.  Present in the build container and
(as a prebuilt, git-ignored .so) on the GPU box; tests that need it skip otherwise
and fall back to the committed golden vectors that were generated with it."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libstb_ref.so")


class StbRef:
    def __init__(self, lib):
        self.lib = lib
        lib.stbi_load_from_memory.restype = C.POINTER(C.c_uint8)
        lib.stbi_load_from_memory.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                              C.POINTER(C.c_int), C.c_int]
        lib.stbi_image_free.argtypes = [C.c_void_p]
        lib.stbi_write_png_to_mem.restype = C.POINTER(C.c_uint8)
        lib.stbi_write_png_to_mem.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        lib.stbi_failure_reason.restype = C.c_char_p

    def load(self, data: bytes, req_comp: int):
        """-> (array HxWxC, channels_in_file) or (None, reason)."""
        x, y, n = C.c_int(), C.c_int(), C.c_int()
        p = self.lib.stbi_load_from_memory(data, len(data), C.byref(x), C.byref(y), C.byref(n), req_comp)
        if not p:
            return None, (self.lib.stbi_failure_reason() or b"").decode()
        comp = req_comp if req_comp else n.value
        arr = np.ctypeslib.as_array(p, shape=(y.value, x.value, comp)).copy()
        self.lib.stbi_image_free(p)
        return arr, n.value

    def write_png(self, img: np.ndarray) -> bytes:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, comp = img.shape
        n = C.c_int()
        p = self.lib.stbi_write_png_to_mem(img.ctypes.data, w * comp, w, h, comp, C.byref(n))
        data = bytes(np.ctypeslib.as_array(p, shape=(n.value,)))
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        libc.free(p)
        return data


def load():
    if not os.path.exists(_PATH):
        return None
    return StbRef(C.CDLL(_PATH))
