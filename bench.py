#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X (BASELINE.json).

A "step" is one pass of the hot path over one frame of synthetic input: the
render kernel over every pixel of the workload's framebuffer.  Default workload
= BASELINE.json configs[2], the configuration the metric is quoted on:
3840x2160 over a 4096^2 heightmap, spherical hfov 180, step_dist 0.25 ("C3";
`--workload C3h` is the north_star's step_dist 0.5 variant, C2/C5 also exist).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3]
  torchrun --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1: K frames of the workload's static pose back to back, maps resident in HBM,
       output to a device buffer.
N > 1: one process per GPU, maps replicated.  Default `--mode frames`: frames are
       independent units, sharded as BASELINE config C5 shards its recording: a
       64-frame orbit of the workload's camera around the map centre, frame k on
       GPU k mod N (rank r renders frames r, r+N, ... and wraps after 64), no
       data-path collective, scaling "weak" (K frames per GPU whatever N).
       `--mode strips`: ONE frame is tiled into cyclic 16-row bands, strips gathered
       on rank 0 over RCCL (config C4's pattern), scaling "strong" -- at ~0.1 ms per
       4K frame that mode measures the gather, not the kernel.

Rank 0 prints ONE JSON line.  value = REFERENCE-EQUIVALENT ray-steps/s: a ray-step is
one execution of the reference's height load (main/hmap.cpp:1013-1014) and the count
is what the reference executes for the same frame (instrumented kernel variant,
bit-identical pixels, run once, untimed; equal to the oracle's count).  The
production kernel proves most of those loads unnecessary and skips them, so the
line also says `"equivalent_steps": true` and carries the height samples and
pyramid look-ups the kernel really executes.

roofline: the kernel is bound by VALU issue, not by HBM, so
`bound` = "valu-issue": achieved = VALU pipe-busy SIMD-cycles per second (PMC counts
from profiles/traffic.json priced with tools/valu_calib.hip's cycles per instruction
class, over the kernel duration measured live with HIP events on the launch stream),
peak = 1024 SIMDs x 2.4 GHz.  The PMC summary is only used when the hash of the
kernel sources it was collected on equals this tree's (else frac is null).
`hbm` holds measured HBM traffic / duration against the 8 TB/s peak; the
BASELINE.md algorithmic-bytes figure (8 B per reference step) is kept under
`algorithmic_equivalent` -- it exceeds the HBM peak because those loads are not
executed, and is not a roofline fraction.
cpu_baseline = the oracle (C port of the reference loop, OpenMP) timed on a bounded
row sample of the same frame on this host.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
SIMDS, PEAK_CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs; max clock (MI355X_MICROARCH.md)
BAND_ROWS = 16
ORBIT_FRAMES = 64  # BASELINE config C5


def _pmc_from_profiles(workload, src_sha):
    """The committed rocprofv3 PMC summary of this workload (profiles/traffic.json, written by
    tools/pmc_summary.py from separate --pmc passes) -- only if it was collected on the kernel
    sources of this tree.  Returns (entry or None, provenance dict)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    prov = {"file": "profiles/traffic.json", "tree_kernel_src_sha": src_sha}
    try:
        with open(p) as f:
            e = json.load(f).get(workload)
    except (OSError, ValueError):
        e = None
    if not e:
        prov["status"] = "no PMC summary for this workload"
        return None, prov
    prov.update(kernel_src_sha=e.get("kernel_src_sha"), git_commit=e.get("git_commit"), source=e.get("source"))
    if e.get("kernel_src_sha") != src_sha:
        prov["status"] = "stale: collected on other kernel sources; not used"
        return None, prov
    prov["status"] = "ok: collected on these kernel sources (separate rocprofv3 --pmc passes, not this run)"
    return e, prov


def cpu_baseline(hmrm, wl, rgb, cmap, params, cam, target_s=15.0):
    """Oracle on a bounded sample: every `stride`-th framebuffer row, stride chosen from a
    quick probe so that the timed run is ~target_s seconds of work on all host cores."""
    from oracle import oracle_py as oracle
    heights = oracle.update_heightmap(rgb, params)
    cfg = oracle.make_cfg(cam, params, wl.map_size, wl.map_size)
    cores = oracle.max_threads()
    probe_stride = max(1, cam.height // 16)
    t0 = time.perf_counter()
    _, probe_steps, _, _, _ = oracle.render(cfg, heights, cmap, row_stride=probe_stride)
    probe_t = max(time.perf_counter() - t0, 1e-6)
    rows_probe = len(range(0, cam.height, probe_stride))
    per_row = probe_t / rows_probe
    rows_target = int(min(cam.height, max(rows_probe, target_s / per_row)))
    stride = max(1, cam.height // rows_target)
    nrows = len(range(0, cam.height, stride))
    # many-core hosts finish the whole frame in well under a second: repeat it (median of 3)
    reps = 3 if per_row * nrows < 2.0 else 1
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        _, steps, _, _, _ = oracle.render(cfg, heights, cmap, row_stride=stride)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    # the reference's own makefile builds without -O: the same port at -O0 on a quarter of the sample
    stride0 = max(1, stride * 4)
    t0 = time.perf_counter()
    _, steps0, _, _, _ = oracle.render(cfg, heights, cmap, row_stride=stride0, opt="O0")
    dt0 = max(time.perf_counter() - t0, 1e-9)
    return {"value": steps / dt, "unit": "ray-steps/s", "cores": cores, "kind": "port",
            "value_at_reference_flags_O0": steps0 / dt0,
            "mrays_per_s": nrows * cam.width / dt / 1e6,
            "sample": f"every {stride}th row of the {cam.width}x{cam.height} frame ({nrows} rows, "
                      f"{steps} ray-steps, {dt:.2f} s wall x {cores} threads, median of {reps}; "
                      f"gcc -O2 -fopenmp -ffp-contract=off)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--mode", choices=["frames", "strips"], default="frames")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="frames mode: consecutive frames go to this many HIP streams round-robin, so that the thin tail "
                         "of one launch overlaps the start of the next (1 = one stream, launches back to back)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import numpy as np
    import torch

    # stdout carries exactly one line, the JSON: libraries that print banners there (RCCL's version
    # block at communicator creation) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    hmrm = importlib.import_module("heightmap-ray-marcher_amd")
    strips = importlib.import_module("heightmap-ray-marcher_amd.strips")
    if not torch.cuda.is_available() or hmrm.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    hmrm.set_device(local_rank)
    dist = None
    # HMRM_FORCE_DIST=1 runs the torch.distributed (RCCL) code path even with one rank, so that
    # the N>1 plumbing can be exercised on a single-GPU box
    force_dist = os.environ.get("HMRM_FORCE_DIST", "") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    wl = hmrm.synth.WORKLOADS[args.workload]
    rgb, cmap = hmrm.synth.synth_maps(wl.map_size)
    params, cam = wl.scene_params(), wl.camera()
    scene = hmrm.Scene(rgb, cmap, params)  # maps resident in HBM from here on
    W, H = cam.width, cam.height
    stream = torch.cuda.current_stream().cuda_stream

    # step / hit counts of the frame (instrumented kernel, untimed; pixels identical)
    fb_ref, st, _, _ = scene.render_stats(cam)
    frame_steps, frame_rays, frame_hits = int(st.steps), int(st.rays), int(st.hits)
    algo_bytes = 8 * frame_steps + 4 * frame_rays + 4 * frame_hits

    orbit = None
    in_flight = 1
    if (world == 1 and not force_dist) or args.mode == "frames":
        # Frames are independent (hmap.cpp:978-983, and the recording loop :1131-1144 renders one after the other):
        # consecutive frames go round-robin to `in_flight` streams, each with its own device frame.
        in_flight = max(1, min(args.frames_in_flight, 4))
        # (with frames in flight every lane gets a stream of its own: the default stream would be one of them otherwise)
        own_streams = [torch.cuda.Stream() for _ in range(in_flight)] if in_flight > 1 else []  # (alive until main returns)
        lanes = [(h, torch.empty((H, W, 4), dtype=torch.uint8, device="cuda"))
                 for h in ([x.cuda_stream for x in own_streams] or [stream])]
        out = lanes[0][1]
        issued = {"n": 0}

        def next_lane():
            lane = lanes[issued["n"] % in_flight]
            issued["n"] += 1
            return lane
        if world == 1 and not force_dist:
            def step():
                st_, o_ = next_lane()
                scene.render_rows_device(cam, o_.data_ptr(), W * 4, 0, H, stream=st_)
            my_steps_timed = frame_steps * args.steps
            parallelism = "1 GPU" + (f", {in_flight} frames in flight on {in_flight} HIP streams" if in_flight > 1 else "")
        else:
            # BASELINE config C5's sharding: frame k of the 64-frame orbit on GPU k mod world
            mine = strips.orbit_frames_of_rank(rank, world, args.warmup + args.steps, ORBIT_FRAMES)
            cams = {k: wl.camera(k, ORBIT_FRAMES) for k in sorted(set(mine))}
            steps_of = {k: int(scene.render_stats(c)[1].steps) for k, c in cams.items()}  # untimed, instrumented
            # every camera of this rank once on the timed stream, untimed: the library keeps the per-frame
            # host set-up (libm calls, spherical tables) of up to 64 cameras per stream, so that the timed
            # laps around the orbit measure the GPU path like the static pose at N = 1 does
            for c in cams.values():
                for st_, o_ in lanes:
                    scene.render_rows_device(c, o_.data_ptr(), W * 4, 0, H, stream=st_)
            orbit = {"it": iter(mine), "last": [None] * in_flight}

            def step():
                k = next(orbit["it"])
                orbit["last"][issued["n"] % in_flight] = k
                st_, o_ = next_lane()
                scene.render_rows_device(cams[k], o_.data_ptr(), W * 4, 0, H, stream=st_)
            my_steps_timed = sum(steps_of[k] for k in mine[args.warmup:])
            parallelism = (f"{ORBIT_FRAMES}-frame orbit, frame k on GPU k mod {world} "
                           f"({args.steps} frames per GPU), no collective"
                           + (f"; {in_flight} frames in flight per GPU" if in_flight > 1 else ""))
        rays_per_step = frame_rays * world
        scaling = "weak"
    else:
        plan = strips.BandPlan(height=H, width=W, band_rows=BAND_ROWS, world=world)
        strip = torch.zeros((plan.strip_rows, W, 4), dtype=torch.uint8, device="cuda")
        block = torch.zeros((world, plan.strip_rows, W, 4), dtype=torch.uint8, device="cuda") if rank == 0 else None
        result = {}

        def render_rows(strip_t, band_rows, band_index, band_count):
            scene.render_rows_device(cam, strip_t.data_ptr(), W * 4, band_rows=band_rows,
                                     band_index=band_index, band_count=band_count, stream=stream)

        def step():
            result["frame"] = strips.render_frame_distributed(plan, rank, render_rows, dist, strip, block)
        my_steps_timed = frame_steps * args.steps if rank == 0 else 0  # one frame per step for the whole job
        rays_per_step = frame_rays
        scaling = "strong"
        parallelism = f"cyclic {BAND_ROWS}-row bands over {world} GPUs + RCCL gather to rank 0"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # 50 untimed launches of the static pose on every rank: the GPUs enter the timed region at steady clocks
    scene.bench_kernel_ms(cam, 50)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    total_steps_timed = my_steps_timed
    # dominant kernel: mean launch duration by HIP events on the launch stream (full frame, static pose, 1 GPU's view)
    kernel_ms = scene.bench_kernel_ms(cam, 50) if rank == 0 else None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        n = torch.tensor([my_steps_timed], dtype=torch.int64, device="cuda")
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        total_steps_timed = int(n.item())

    # correctness of what was just timed: every rank's last frame equals what the instrumented
    # kernel (another instantiation, host read-back path) renders for the same camera
    if orbit is not None:
        for (st_, o_), k in zip(lanes, orbit["last"]):
            if k is not None and not np.array_equal(o_.cpu().numpy(), scene.render_stats(cams[k])[0]):
                raise SystemExit(f"bench.py: rank {rank} rendered a different orbit frame {k} than hmrm_render_stats")
    elif (world > 1 or force_dist) and args.mode == "strips":
        if rank == 0 and not np.array_equal(result["frame"].cpu().numpy(), fb_ref):
            raise SystemExit("bench.py: timed path produced a different frame than hmrm_render_stats")
    else:
        for st_, o_ in lanes[:min(in_flight, args.warmup + args.steps)]:
            if not np.array_equal(o_.cpu().numpy(), fb_ref):
                raise SystemExit("bench.py: timed path produced a different frame than hmrm_render_stats")
    # the same K steps launched back to back on ONE stream (untimed region; reported beside `value`)
    single = None
    if in_flight > 1 and orbit is None and world == 1 and not force_dist:
        for _ in range(50 + args.warmup):  # (the frame checks above left the GPU idle: back to steady clocks first)
            scene.render_rows_device(cam, out.data_ptr(), W * 4, 0, H, stream=stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            scene.render_rows_device(cam, out.data_ptr(), W * 4, 0, H, stream=stream)
        torch.cuda.synchronize()
        single = (time.perf_counter() - t1) / args.steps


    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_steps_timed / elapsed
        kernel_s = kernel_ms * 1e-3
        pmc, prov = _pmc_from_profiles(wl.name, hmrm.kernel_src_sha())
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        valu = (pmc or {}).get("valu") or {}
        # VALU pipe-busy cycles per launch, three ways (tools/pmc_summary.py): `isa` = SQ_INSTS_VALU x the calibrated
        # pipe cost of the march loop's own instruction mix (tools/isa_cost.py on the compiler's assembly,
        # profiles/r02_valu_calibration.txt) -- the best estimate, and `frac`; `weighted` prices only the classes the
        # counters separate and everything else as a 2.3-cycle instruction -- a lower bound; `upper` =
        # SQ_ACTIVE_INST_VALU x 4, i.e. rocprofv3's derived metric VALUBusy, which charges every instruction 4 cycles.
        busy_lower = valu.get("busy_cycles_weighted")
        busy_upper = valu.get("busy_cycles_upper")
        busy = valu.get("busy_cycles_isa") or busy_lower
        peak = SIMDS * PEAK_CLOCK_GHZ  # G SIMD-cycles/s
        achieved = busy / kernel_s / 1e9 if busy else None
        frac = achieved / peak if achieved else None
        if frac is not None and frac > 1.0:  # a fraction above 1 would mean the pricing is wrong: do not print it
            prov["status"] += "; the priced VALU figure exceeded the peak and was dropped"
            achieved = frac = None
        roofline = {
            "bound": "valu-issue", "achieved": achieved, "peak": peak, "unit": "G SIMD-cycles/s", "frac": frac,
            "frac_basis": ("SQ_INSTS_VALU x %.2f cycles (the loop's instruction mix at calibrated pipe costs)"
                           % valu["cycles_per_inst_isa"]) if valu.get("busy_cycles_isa") else
                          ("counter-separable classes priced, the rest at 2.3 cycles (lower bound)" if busy_lower else None),
            "frac_lower_bound": (busy_lower / kernel_s / 1e9 / peak) if busy_lower else None,
            "frac_if_every_valu_held_the_pipe_4_cycles": (busy_upper / kernel_s / 1e9 / peak) if busy_upper else None,  # = VALUBusy
            "traffic": traffic,
            "hbm": {"achieved": (traffic / kernel_s / 1e9) if traffic else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (traffic / kernel_s / 1e9 / HBM_PEAK_GBS) if traffic else None},
            "lane_util": (pmc or {}).get("lane_util"),
            "kernel": "k_render_fast", "kernel_ms": kernel_ms,
            "kernel_ray_steps_per_s": frame_steps / kernel_s,
            # BASELINE.md's nominal figure: bytes the REFERENCE's loop would move for this frame over the
            # measured duration.  Not executed traffic (the loads are skipped), hence not a fraction of a peak.
            "algorithmic_equivalent": {"bytes_per_launch": algo_bytes, "gbs": algo_bytes / kernel_s / 1e9,
                                       "times_hbm_peak": algo_bytes / kernel_s / 1e9 / HBM_PEAK_GBS},
            "pmc": prov,
        }
        line = {
            "metric": "ray-steps/s at 3840x2160, 4096^2 heightmap" if wl.map_size == 4096 else
                      f"ray-steps/s at {W}x{H}, {wl.map_size}^2 heightmap",
            "value": value, "unit": "ray-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "equivalent_steps": True,
            "frames_in_flight": in_flight,
            "one_stream": ({"ms_per_step": single * 1e3, "value": frame_steps / single} if single else None),
            "executed_per_frame": {"height_samples": int(st.groups) * 4 if st.groups else None,
                                   "pyramid_lookups": int(st.leap_attempts),
                                   "steps_covered_by_exact_leaps": int(st.leaped_steps)},
            "mrays_per_s": rays_per_step * args.steps / elapsed / 1e6,
            "config": {"workload": f"{wl.name}: {wl.map_size}x{wl.map_size} heightmap, {W}x{H}, "
                                   f"{('perspective', 'spherical', 'orthographic')[wl.projection - 1]} "
                                   f"hfov {wl.hfov_deg:g}, step_dist {wl.step_dist:g}, grid_width 1",
                       "ray_steps_per_frame": frame_steps, "rays_per_frame": frame_rays,
                       "hits_per_frame": frame_hits, "parallelism": parallelism,
                       "maps_sha256": hmrm.synth.maps_sha256(rgb, cmap)[:16]},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(hmrm, wl, rgb, cmap, params, cam, args.cpu_seconds)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    scene.close()


if __name__ == "__main__":
    main()
