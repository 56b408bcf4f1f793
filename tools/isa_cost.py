#!/usr/bin/env python3
"""VALU pipe cost of one k_render_fast instantiation, instruction by instruction (csrc/_build/render_fast.s, `make asm`).

SQ_ACTIVE_INST_VALU counts one quad-cycle per VALU instruction whatever it is (profiles/r02_valu_calibration.txt), so
rocprofv3's VALUBusy prices every instruction at 4 cycles.  tools/valu_calib.hip measured what the classes really hold
the pipe for at 8 waves per SIMD: ~4.15 cycles for anything 64-bit (fp64 arithmetic / compare / min / max / convert,
64-bit integer), for compares, VOP3 selects and the 32-bit multiply, ~2.25 for plain 32-bit VOP1/VOP2 work and 16.2 for
fp64 rcp / sqrt.  This prices each instruction of the kernel with that table and prints, per basic block and for the
whole loop, instructions, cycles and cycles per instruction.  The loop's average is what turns the instruction count
SQ_INSTS_VALU into busy cycles (tools/pmc_summary.py's `weighted` figure prices only the classes the counters separate).
usage: isa_cost.py [instantiation substring]"""
import os, re, sys
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "heightmap-ray-marcher_amd", "csrc", "_build", "render_fast.s")
COST = {"trans64": 16.2, "wide": 4.15, "narrow": 2.3}
if os.environ.get("ISA_COST_TABLE"):  # e.g. "vop3_32=2.25" after a calibration of the 32-bit VOP3 forms
    for kv in os.environ["ISA_COST_TABLE"].split(","):
        k, v = kv.split("=")
        COST[k] = float(v)
COST.setdefault("vop3_32", 4.17)  # three-operand / VOP3-only 32-bit integer forms (v_add3_u32, v_lshl_add_u32, v_mad_u32_u24, v_and_or_b32: measured)


def classify(ins):
    op = ins.split()[0]
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
        return "trans64"
    if "_f64" in op or op.endswith(("_b64", "_u64", "_i64")) or "_b64_" in op or "_u64_" in op or "_i64_" in op:
        return "wide"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "wide"
    if op.startswith("v_cndmask_b32_e64") or op in ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32"):
        return "wide"
    if re.match(r"v_(min|max)_[iu]32", op):  # (measured: 4.15, although VOP2)
        return "wide"
    if op.startswith(("v_add3", "v_lshl_add_u32", "v_lshl_or", "v_and_or", "v_or3", "v_mad_u32_u24", "v_mad_i32_i24", "v_bfe",
                      "v_bfi", "v_xad", "v_add_lshl", "v_min3", "v_max3", "v_med3", "v_alignbit", "v_perm")) or op.endswith("_e64"):
        return "vop3_32"
    return "narrow"


def blocks_of(want, asm_path=path):
    """[(label, [instructions], in_loop)] of the instantiation whose mangled name contains `want`."""
    s = open(asm_path).read()
    parts = re.split(r'\t\.type\t(_ZN4hmrm13k_render_fastI[^,]+),@function\n', s)
    body = next((parts[i + 1].split('s_endpgm')[0] for i in range(1, len(parts), 2) if want in parts[i]), None)
    if body is None:
        return None
    blocks, cur = [], ["entry", [], False]
    for l in body.split('\n'):
        t = l.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            blocks.append(cur)
            cur = [m.group(1), [], "in Loop" in t or "Loop Header" in t]
            continue
        if not t or t.startswith((';', '.')):
            continue
        cur[1].append(t.split(';')[0].strip())
    blocks.append(cur)
    return blocks


def loop_cost(want, asm_path=path):
    """(VALU instructions, pipe cycles) of the march loop's blocks of one instantiation, or None."""
    blocks = blocks_of(want, asm_path)
    if blocks is None:
        return None
    v = [x for _, ins, loop in blocks if loop for x in ins if x.startswith("v_")]
    return len(v), sum(COST[classify(x)] for x in v)


def mangled(template_args):
    """'2, false, 0, 1, 0' (as rocprofv3 prints the kernel; the fourth argument -- 0 plain groups, 1 leaps, 2 records -- was a
    bool until round 5's record kernel) -> the Itanium-mangled argument list."""
    p, stats, gwm, leap, samp = [a.strip() for a in template_args.split(",")]
    b = lambda x: "1" if x == "true" else "0"
    leap = {"true": "1", "false": "0"}.get(leap, leap)
    return f"ILi{p}ELb{b(stats)}ELi{gwm}ELi{leap}ELi{samp}E"


if __name__ == "__main__":
    want = sys.argv[1] if len(sys.argv) > 1 else "ILi2ELb0ELi0ELi1ELi0E"
    blocks = blocks_of(want)
    if blocks is None:
        sys.exit("no such instantiation")
    tot = {"all": [0, 0.0], "loop": [0, 0.0]}
    hist = {}
    for name, ins, loop in blocks:
        v = [x for x in ins if x.startswith("v_")]
        cyc = sum(COST[classify(x)] for x in v)
        for x in v:
            hist[classify(x)] = hist.get(classify(x), 0) + (1 if loop else 0)
        if v:
            print(f"{name:12s} {'loop ' if loop else '     '} valu {len(v):4d}  cycles {cyc:7.1f}  {cyc / len(v):.2f} per instruction")
        for k in (["all", "loop"] if loop else ["all"]):
            tot[k][0] += len(v)
            tot[k][1] += cyc
    for k, (n, c) in tot.items():
        print(f"{k:5s}: {n} VALU instructions, {c:.0f} cycles, {c / max(n, 1):.2f} per instruction  (VALUBusy prices 4.00)")
    print("loop instructions by class:", hist, " costs:", COST)
