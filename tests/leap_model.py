"""Pure-Python model of the exact-leap arithmetic of csrc/render_fast.hip
(axis_refresh / axis_landing_ok), used by CPU tests to check the claim the GPU
kernel relies on:

    inside one binade, the reference's sequential accumulation p_{k+1} = fl(p_k + s)
    (main/hmap.cpp:1037) satisfies p_n = p_0 + n*delta exactly.

Python floats are IEEE binary64 with round-to-nearest-even, like the device.
"""
import struct


def hi32(v: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", v))[0] >> 32


def lo32(v: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", v))[0] & 0xFFFFFFFF


def f64_from_hi(hi: int) -> float:
    return struct.unpack("<d", struct.pack("<Q", (hi & 0xFFFFFFFF) << 32))[0]


class Axis:
    __slots__ = ("delta", "lim", "rdel", "key", "left")

    def __init__(self):
        self.delta, self.lim, self.rdel, self.key, self.left = 0.0, 0.0, 0.0, 0xFFFFFFFE, -1


def cvt_i32_sat(v: float) -> int:
    """v_cvt_i32_f64: truncation toward zero, saturating; NaN -> 0."""
    if v != v:
        return 0
    return max(-2 ** 31, min(2 ** 31 - 1, int(v)))


def axis_refresh(a: Axis, p: float, s: float, rcp_err: float = 0.0, short: int = 2) -> None:
    p1 = p + s
    p2 = p1 + s
    hp, hp1, hp2 = hi32(p), hi32(p1), hi32(p2)
    e = (hp >> 20) & 0x7FF
    d = p1 - p
    ok = ((hp ^ hp1) >> 20) == 0 and ((hp ^ hp2) >> 20) == 0 and 128 <= e <= 1900 and (p2 - p1) == d
    a.key = (hp >> 20) if ok else 0xFFFFFFFF
    a.delta = d
    if d == 0.0:
        a.lim, a.rdel = p + 1.0, 2.0 ** 40
    else:
        lo = f64_from_hi(hp & 0x7FF00000)
        away = ((hi32(d) ^ hp) >> 31) == 0
        lim_abs = lo + lo if away else lo
        a.lim = -lim_abs if (hp >> 31) else lim_abs
        a.rdel = 1.0 / d  # the device uses an approximate reciprocal; only an estimate anyway
    # kStepsLeft (leap_common.hpp): further steps certain to stay strictly inside the binade -- the estimate, which
    # the device forms with a reciprocal good to 2^-24 (rcp_err models that), shortened by 2^-22 of itself and by `short`
    # steps (2; 0 in HMRM_CROSS builds), and verified at its far end
    if not ok:
        a.left = -1
    elif d == 0.0:
        a.left = 1 << 30
    else:
        k = max(0, min(1 << 30, cvt_i32_sat((a.lim - p) * (a.rdel * (1.0 + rcp_err)) * (1.0 - 2.0 ** -22)) - short))
        pk = p + float(k) * d
        inside = (hi32(pk) >> 20) == (hp >> 20) and ((hi32(pk) & 0xFFFFF) != 0 or lo32(pk) != 0)
        a.left = k if inside else 0


def axis_landing_ok(a: Axis, pn: float) -> bool:
    if (hi32(pn) >> 20) != a.key:
        return False
    return a.delta == 0.0 or (hi32(pn) & 0xFFFFF) != 0 or lo32(pn) != 0


def sequential(p: float, s: float, n: int) -> float:
    for _ in range(n):
        p = p + s
    return p
