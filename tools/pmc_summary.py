#!/usr/bin/env python3
"""Summarise rocprofv3 outputs under a directory tree into one text file + profiles/traffic.json.

usage: pmc_summary.py <gpurun_out/prof_dir> <workload> <out_prefix>

Reads every */*_counter_collection.csv, *_kernel_trace.csv and *_kernel_stats.csv below the directory.
A frame is one launch of the product's render kernel (k_render_fast); counters are averaged per dispatch
and kernel (and summed over kernels, should a frame ever take more than one).  The instrumented instantiations (STATS = true) are left out.

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 sections, and tools/valu_calib.hip):
  * FETCH_SIZE / WRITE_SIZE are in KiB.  FETCH_SIZE counts 64 B per request and reads half the bytes of
    wide (16 B per lane) coalesced streams on gfx950; this kernel's reads are 4- and 8-byte gathers, an
    uncalibrated width, so the raw figure is kept and twice the figure is listed as an upper bound.
  * SQ_ACTIVE_INST_VALU adds one unit (a quad-cycle) per VALU instruction whatever its class and four
    per fp64 transcendental (calibration: profiles/r02_valu_calibration.txt), so x4 it is the pipe
    occupancy only if every instruction held the pipe for 4 cycles: an UPPER bound.  The weighted
    figure prices the classes the counters can tell apart as measured with 8 waves per SIMD: 4.15
    cycles for fp64 add/mul/fma, 16.2 for fp64 rcp/sqrt, 4.2 for conversions, 2.25 for everything else
    -- a LOWER bound, because fp64 compares / min / max / fract, compares that write an SGPR pair,
    v_mul_lo_u32 and VOP3 selects also take 4.1-4.2 cycles but are counted in no class of their own.
  * lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU).
The JSON entry carries the hash of the kernel sources (lib.kernel_src_sha) so that bench.py can refuse
numbers collected on another kernel."""
import collections, csv, glob, importlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root, workload, out_prefix = sys.argv[1], sys.argv[2], sys.argv[3]

SIMDS = 1024
PEAK_CLOCK_HZ = 2.4e9
CYC_F64, CYC_TRANS64, CYC_CVT, CYC_OTHER = 4.15, 16.2, 4.2, 2.25


def product_kernel(name):
    if "k_render_fast<" in name or "k_render<" in name:
        args = name.split("<", 1)[1]
        fields = [a.strip() for a in args.split(">")[0].split(",")]
        return len(fields) < 2 or fields[1] != "true"  # STATS is the second template argument
    return False


def short(name):
    return name.split("(")[0].replace("void hmrm::", "")


def newest_per_dir(pattern):
    """gpurun merges a call's output INTO the local copy of the directory: a tag used twice leaves the older run's
    files (other pids) beside the new ones.  One rocprofv3 pass writes one file of a kind per directory: keep the newest."""
    best = {}
    for f in glob.glob(pattern, recursive=True):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


# tools/prof_run.py spends its first launches on the launch-order calibration (measured launches, trial orders): those
# dispatches are left out of the counter averages
SKIP_FIRST = int(os.environ.get("PMC_SKIP_FIRST", "12"))
per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
for f in newest_per_dir(os.path.join(root, "**", "*_counter_collection.csv")):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        if product_kernel(r["Kernel_Name"]):
            per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
    skipped = set(sorted(names, key=int)[:SKIP_FIRST])
    for (disp, cname), v in per_dispatch.items():
        if disp not in skipped:
            per_kernel[names[disp]][cname].append(v)

durations = collections.defaultdict(list)
for f in newest_per_dir(os.path.join(root, "trace", "**", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if product_kernel(r["Kernel_Name"]):
            durations[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

# One kernel renders a frame.  Others of the product's instantiations may appear a couple of times in a run -- the scene's
# one-time kernel probe launches the plain-groups instantiation twice during the calibration -- and are left out: the
# frame's kernel is the one with the most launches.
if durations:
    main = max(durations, key=lambda k: len(durations[k]))
    durations = {main: durations[main]}
if per_kernel:
    main_pk = max(per_kernel, key=lambda k: max(len(v) for v in per_kernel[k].values()))
    per_kernel = {main_pk: per_kernel[main_pk]}

lines, frame = [], collections.defaultdict(float)
for k in sorted(per_kernel):
    lines.append(f"== {k}")
    for c in sorted(per_kernel[k]):
        v = per_kernel[k][c]
        m = sum(v) / len(v)
        frame[c] += m
        lines.append(f"   {c:28s} {m:18.1f}  (mean of {len(v)} dispatches)")
frame_ns = 0.0
for k in sorted(durations):
    v = durations[k]
    frame_ns += sum(v) / len(v)
    lines.append(f"== {k}: {len(v)} launches in the kernel trace, mean {sum(v) / len(v) / 1e3:.2f} us, "
                 f"min {min(v) / 1e3:.2f}, max {max(v) / 1e3:.2f}")

entry = {"kernels": sorted(per_kernel), "counters_per_frame": dict(sorted(frame.items())),
         "kernel_us_per_frame_rocprof": frame_ns / 1e3 if frame_ns else None}
sys.path.insert(0, ROOT)
os.environ.setdefault("HMRM_NO_TORCH_PRELOAD", "1")  # (only the source hash is needed, not torch)
entry["kernel_src_sha"] = importlib.import_module("heightmap-ray-marcher_amd.lib").kernel_src_sha()
try:
    entry["git_commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True,
                                         text=True).stdout.strip() or None
except OSError:
    entry["git_commit"] = None

if "FETCH_SIZE" in frame and "WRITE_SIZE" in frame:
    fetch, write = frame["FETCH_SIZE"] * 1024.0, frame["WRITE_SIZE"] * 1024.0
    entry.update(hbm_bytes_per_launch=fetch + write, fetch_bytes_raw=fetch, write_bytes=write,
                 fetch_bytes_if_undercounted_2x=2 * fetch)
    lines.append(f"HBM traffic per frame: FETCH {fetch / 1e6:.1f} MB (raw; <= {2 * fetch / 1e6:.1f} MB if the 2x "
                 f"under-count of wide streams applied) + WRITE {write / 1e6:.1f} MB")
if "SQ_INSTS_VALU" in frame:
    n = frame["SQ_INSTS_VALU"]
    f64 = sum(frame.get("SQ_INSTS_VALU_" + c, 0.0) for c in ("ADD_F64", "MUL_F64", "FMA_F64"))
    trans = frame.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
    cvt = frame.get("SQ_INSTS_VALU_CVT", 0.0)
    have_classes = "SQ_INSTS_VALU_ADD_F64" in frame
    valu = {"insts": n, "f64_add_mul_fma": f64 if have_classes else None, "f64_trans": trans if have_classes else None,
            "cvt": cvt if have_classes else None,
            "busy_cycles_upper": 4.0 * frame.get("SQ_ACTIVE_INST_VALU", n)}
    if have_classes:
        valu["busy_cycles_weighted"] = (CYC_F64 * f64 + CYC_TRANS64 * trans + CYC_CVT * cvt +
                                        CYC_OTHER * max(n - f64 - trans - cvt, 0.0))
    # Best estimate: every instruction priced at its calibrated pipe cost, with the kernel's own instruction mix --
    # the march loop of the instantiation that ran, read from the compiler's assembly (tools/isa_cost.py)
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import isa_cost
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "heightmap-ray-marcher_amd", "csrc"), "asm"], check=True,
                       capture_output=True)
        fast = [k for k in per_kernel if k.startswith("k_render_fast<")] or [k for k in durations if k.startswith("k_render_fast<")]
        lc = isa_cost.loop_cost(isa_cost.mangled(fast[0][fast[0].index("<") + 1:fast[0].rindex(">")])) if fast else None
        if lc:
            valu["loop_valu_insts_static"], valu["loop_cycles_static"] = lc[0], lc[1]
            valu["cycles_per_inst_isa"] = lc[1] / lc[0]
            valu["busy_cycles_isa"] = n * lc[1] / lc[0]
    except (OSError, subprocess.CalledProcessError, ValueError, IndexError) as e:
        lines.append(f"(no instruction-priced estimate: {e})")
    entry["valu"] = valu
    lines.append(f"VALU per frame: {n:.0f} wave-instructions; fp64 add/mul/fma {f64:.0f}, fp64 rcp/sqrt {trans:.0f}, conversions {cvt:.0f}")
if "SQ_THREAD_CYCLES_VALU" in frame and "SQ_ACTIVE_INST_VALU" in frame:
    entry["lane_util"] = frame["SQ_THREAD_CYCLES_VALU"] / (64.0 * frame["SQ_ACTIVE_INST_VALU"])
    lines.append(f"lane utilisation (SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU): {entry['lane_util']:.3f}")
if "SQ_WAVE_CYCLES" in frame and frame["SQ_WAVE_CYCLES"] > 0:
    # where a resident wave's time goes (SQ_WAVE_CYCLES counts quad-cycles of residency over all waves)
    wc = frame["SQ_WAVE_CYCLES"]
    wt = {"issuing": frame.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, "waiting_on_memory_or_dependency": frame.get("SQ_WAIT_INST_ANY", 0.0) / wc}
    if "TCC_HIT_sum" in frame and "TCC_MISS_sum" in frame and frame["TCC_HIT_sum"] + frame["TCC_MISS_sum"] > 0:
        wt["tcc_hit_rate"] = frame["TCC_HIT_sum"] / (frame["TCC_HIT_sum"] + frame["TCC_MISS_sum"])
    if frame_ns:
        wt["resident_waves_mean"] = wc * 4.0 / (frame_ns * 1e-9 * PEAK_CLOCK_HZ)
    entry["wave_time"] = wt
    lines.append("wave time: " + ", ".join(f"{k} {v:.3f}" for k, v in wt.items()))
if frame_ns and "valu" in entry:
    simd_cycles = SIMDS * frame_ns * 1e-9 * PEAK_CLOCK_HZ
    lines.append(f"SIMD-cycles available per frame at 2.4 GHz over {frame_ns / 1e3:.1f} us: {simd_cycles:.3e}; "
                 f"VALU busy: upper {entry['valu']['busy_cycles_upper'] / simd_cycles:.3f}"
                 + (f", instruction-priced {entry['valu']['busy_cycles_isa'] / simd_cycles:.3f} "
                    f"({entry['valu']['cycles_per_inst_isa']:.2f} cycles per instruction of the loop's mix)"
                    if entry['valu'].get('busy_cycles_isa') else "")
                 + (f", weighted {entry['valu']['busy_cycles_weighted'] / simd_cycles:.3f}"
                    if entry['valu'].get('busy_cycles_weighted') else ""))

stats = [open(f).read() for f in newest_per_dir(os.path.join(root, "**", "*_kernel_stats.csv"))]
text = "\n".join(lines) + "\n\n" + "\n".join(stats)
open(out_prefix + ".txt", "w").write(text)
print(text)
tj = os.path.join(os.path.dirname(out_prefix), "traffic.json")
data = json.load(open(tj)) if os.path.exists(tj) else {}
entry["source"] = os.path.basename(out_prefix) + ".txt"
data[workload] = entry
json.dump(data, open(tj, "w"), indent=1, sort_keys=True)
