#!/usr/bin/env python3
"""Table of tools/valu_calib.hip's results: usage  calib_summary.py <gpurun_out/calib> <out.txt>

Per instruction class: SIMD-cycles per wave-instruction with 8 waves and with 1 wave per SIMD (clock from
GRBM_GUI_ACTIVE / 8 XCDs / duration of the same dispatch), what SQ_ACTIVE_INST_VALU adds per instruction,
and which SQ_INSTS_VALU_<class> counters the instruction increments."""
import collections, csv, glob, os, re, sys

root, out_path = sys.argv[1], sys.argv[2]


def load(d):
    fs = glob.glob(os.path.join(root, d, "*", "*counter_collection.csv"))
    if not fs:
        return {}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        per[int(r["Dispatch_Id"])]["_name"] = r["Kernel_Name"]
    kt = glob.glob(os.path.join(root, d, "*", "*kernel_trace.csv"))[0]
    for r in csv.DictReader(open(kt)):
        i = int(r["Dispatch_Id"])
        if i in per:
            per[i]["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    # the timed launch of each mode is the second (longer) dispatch of its kernel
    best = {}
    for i, c in per.items():
        m = int(re.search(r"k_calib<(\d+)>", c["_name"]).group(1))
        if m not in best or c["_ns"] > best[m]["_ns"]:
            best[m] = c
    return best


names = {}
for line in open(os.path.join(root, "plain_8waves.txt")):
    m = re.match(r"(.+?)\s+blocks\s+\d+", line)
    if m:
        names[len(names)] = m.group(1).strip()
p8, p1, cls = load("pmc8"), load("pmc1"), load("cls8")
lines = ["instruction                      cyc/inst(8 waves/SIMD)  cyc/inst(1 wave/SIMD)  ACTIVE_INST_VALU/inst  clock GHz  class counters (per instruction)"]
for m in sorted(names):
    def cyc(c):
        if not c:
            return float("nan"), float("nan")
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        return 1024.0 * cycles / c["SQ_INSTS_VALU"], cycles / c["_ns"]
    c8, ghz = cyc(p8.get(m))
    c1, _ = cyc(p1.get(m))
    act = p8[m]["SQ_ACTIVE_INST_VALU"] / p8[m]["SQ_INSTS_VALU"] if m in p8 else float("nan")
    cc = ""
    if m in cls:
        tot = cls[m]["SQ_INSTS_VALU"]
        cc = " ".join(f"{k[14:]}={v / tot:.2f}" for k, v in sorted(cls[m].items())
                      if k.startswith("SQ_INSTS_VALU_") and v / tot > 0.004)
    lines.append(f"{names[m]:32s} {c8:10.2f} {c1:22.2f} {act:22.2f} {ghz:10.2f}  {cc}")
lines.append("")
lines.append("(cycles per instruction are over ALL VALU instructions of the dispatch: modes with helper instructions, "
             "e.g. 'v_cvt_i32_f64 (+v_xor)', average the two)")
open(out_path, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
