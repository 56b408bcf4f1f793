#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X (BASELINE.json).

A "step" is one pass of the hot path over one frame of synthetic input: the
render kernel over every pixel of the workload's framebuffer.  Default workload
= BASELINE.json configs[2], the configuration the metric is quoted on:
3840x2160 over a 4096^2 heightmap, spherical hfov 180, step_dist 0.25 ("C3";
`--workload C3h` is the north_star's step_dist 0.5 variant, C2/C5 also exist).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3]
  torchrun --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1: `value` / `ms_per_step` = K frames of the workload's static pose launched BACK TO BACK ON ONE
       STREAM, maps resident in HBM, output to a device buffer -- the operating point `roofline`
       describes (`roofline.kernel_ms` = the same launches timed with HIP events on the launch stream).
       Secondary blocks in the same line (never `value`):
         frames_in_flight  the same K frames round-robin over 3 HIP streams (one launch's tail overlaps
                           the next launch's start: a throughput mode for sequences of frames);
         fresh_camera      a moving camera: every timed frame has a camera the library has never seen
                           (orbit positions; host set-up -- libm calls, spherical sin/cos tables, upload --
                           inside the timed loop), then the same cameras again out of the per-stream cache;
                           for this workload and for C5 (perspective, same maps).
N > 1: one process per GPU over RCCL, maps replicated.  `value` = frames of the workload's 64-frame
       orbit, frame k on GPU k mod N (BASELINE config C5's sharding rule), K frames per GPU on one
       stream each, no data-path collective, scaling "weak".  The same line carries `rccl_ranks`
       (distinct (host, device) pairs seen by an all_gather: did RCCL really see N ranks) and the two
       BASELINE multi-GPU configs as secondary blocks: `c5_frames` (C5's camera, same sharding) and
       `c4_strips` (ONE 7680x4320 frame over the 8192^2 map in cyclic 16-row bands, twice: strips
       gathered to rank 0 over RCCL, and every rank copying its strip to its own pinned host memory
       over its own PCIe link, no collective).  `--mode strips` makes the gather variant of the
       workload `value` instead (scaling "strong").

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
row sample of the same frame on this host, median of 3 repetitions.
"""
import argparse
import importlib
import json
import os
import socket
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


def cpu_baseline(hmrm, wl, rgb, cmap, params, cam, target_s=15.0, reps=3):
    """Oracle on a bounded sample: every `stride`-th framebuffer row, stride chosen from a warm probe so that
    the `reps` timed repetitions together are ~target_s seconds of work on all host cores; median reported."""
    from oracle import oracle_py as oracle
    heights = oracle.update_heightmap(rgb, params)
    cfg = oracle.make_cfg(cam, params, wl.map_size, wl.map_size)
    cores = oracle.max_threads()
    probe_stride = max(1, cam.height // 16)
    oracle.render(cfg, heights, cmap, row_stride=probe_stride)  # (thread pool start-up, page faults: not timed)
    t0 = time.perf_counter()
    oracle.render(cfg, heights, cmap, row_stride=probe_stride)
    probe_t = max(time.perf_counter() - t0, 1e-6)
    rows_probe = len(range(0, cam.height, probe_stride))
    per_row = probe_t / rows_probe
    rows_target = int(min(cam.height, max(rows_probe, target_s / reps / per_row)))
    stride = max(1, cam.height // rows_target)
    nrows = len(range(0, cam.height, stride))
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        _, steps, _, _, _ = oracle.render(cfg, heights, cmap, row_stride=stride)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    # the reference's own makefile builds without -O: the same port at -O0 on a quarter of the sample
    stride0 = max(1, stride * 4)
    times0 = []
    for _ in range(reps):
        t0 = time.perf_counter()
        _, steps0, _, _, _ = oracle.render(cfg, heights, cmap, row_stride=stride0, opt="O0")
        times0.append(max(time.perf_counter() - t0, 1e-9))
    dt0 = sorted(times0)[len(times0) // 2]
    return {"value": steps / dt, "unit": "ray-steps/s", "cores": cores, "kind": "port",
            "value_at_reference_flags_O0": steps0 / dt0,
            "mrays_per_s": nrows * cam.width / dt / 1e6,
            "repetitions_s": [round(t, 4) for t in times],
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
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="streams the HEADLINE loop sends consecutive frames to (1 = back to back on one stream, the "
                         "operating point of `roofline`; the 3-stream figure is always reported as a secondary block)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip frames_in_flight / fresh_camera / c5_frames / c4_strips")
    ap.add_argument("--no-c4", action="store_true", help="N > 1: skip the c4_strips block (8192^2 maps)")
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
    multi = world > 1 or force_dist
    rccl_ranks = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
        # did RCCL see `world` ranks on `world` different GPUs?  Every rank contributes (host hash, device index,
        # PCI bus id hash); the gathered rows are counted on rank 0.
        props = torch.cuda.get_device_properties(local_rank)
        bus = getattr(props, "pci_bus_id", local_rank)
        me = torch.tensor([hash(socket.gethostname()) & 0x7fffffff, torch.cuda.current_device(), int(bus) & 0x7fffffff],
                          dtype=torch.int64, device="cuda")
        seen = [torch.zeros_like(me) for _ in range(world)]
        dist.all_gather(seen, me)
        rccl_ranks = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                      "distinct_devices": len({tuple(int(v) for v in t.tolist()) for t in seen})}

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, n):
        """n calls of step() bracketed by barrier + synchronize; max over ranks -> seconds."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def total_over_ranks(v):
        if dist is None:
            return int(v)
        n = torch.tensor([int(v)], dtype=torch.int64, device="cuda")
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        return int(n.item())

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

    def make_lanes(n, width=W, height=H):
        """n (stream handle, device frame) pairs; with n > 1 every lane gets a stream of its own."""
        own = [torch.cuda.Stream() for _ in range(n)] if n > 1 else []
        handles = [x.cuda_stream for x in own] or [stream]
        return own, [(h, torch.empty((height, width, 4), dtype=torch.uint8, device="cuda")) for h in handles]

    def round_robin(lanes):
        state = {"n": 0}

        def nxt():
            lane = lanes[state["n"] % len(lanes)]
            state["n"] += 1
            return lane
        return nxt, state

    def orbit_leg(the_wl, the_scene, n_frames, n_warm, in_flight):
        """Frame k of the workload's 64-frame orbit on rank k mod world, n_frames timed per rank after n_warm untimed;
        cameras pre-rendered once (cached host set-up: this leg measures the GPU path).  -> (seconds, total steps,
        check callable)."""
        mine = strips.orbit_frames_of_rank(rank, world, n_warm + n_frames, ORBIT_FRAMES)
        cams = {k: the_wl.camera(k, ORBIT_FRAMES) for k in sorted(set(mine))}
        steps_of = {k: int(the_scene.render_stats(c)[1].steps) for k, c in cams.items()}  # untimed, instrumented
        keep, lanes = make_lanes(in_flight, the_wl.width, the_wl.height)
        for c in cams.values():
            for st_, o_ in lanes:
                the_scene.render_rows_device(c, o_.data_ptr(), the_wl.width * 4, 0, the_wl.height, stream=st_)
        nxt, state = round_robin(lanes)
        it = iter(mine)
        last = [None] * len(lanes)

        def step():
            k = next(it)
            last[state["n"] % len(lanes)] = k
            st_, o_ = nxt()
            the_scene.render_rows_device(cams[k], o_.data_ptr(), the_wl.width * 4, 0, the_wl.height, stream=st_)
        for _ in range(n_warm):
            step()
        dt = timed(step, n_frames)

        def check():
            for (st_, o_), k in zip(lanes, last):
                if k is not None and not np.array_equal(o_.cpu().numpy(), the_scene.render_stats(cams[k])[0]):
                    raise SystemExit(f"bench.py: rank {rank} rendered a different orbit frame {k} than hmrm_render_stats")
        del keep
        return dt, total_over_ranks(sum(steps_of[k] for k in mine[n_warm:])), check

    # Untimed, before any warm-up step: the library calibrates the launch order of a camera it sees repeatedly from a few
    # measured launches (csrc/launch_order.cpp plan_order_from_measurement; each needs the previous one finished, which a
    # renderer that presents its frames does by itself and a loop that queues hundreds of launches ahead does not):
    # a dozen launches of the static pose with a synchronisation in between, then 50 more so that the GPUs enter the
    # timed region at steady clocks.
    for _ in range(12):
        scene.bench_kernel_ms(cam, 1)
    scene.bench_kernel_ms(cam, 50)

    result = {}
    in_flight = max(1, min(args.frames_in_flight, 4))
    if not multi:
        keep0, lanes = make_lanes(in_flight)
        nxt, _ = round_robin(lanes)

        def step():
            st_, o_ = nxt()
            scene.render_rows_device(cam, o_.data_ptr(), W * 4, 0, H, stream=st_)
        for _ in range(args.warmup):
            step()
        elapsed = timed(step, args.steps)
        total_steps_timed = frame_steps * args.steps
        rays_timed = frame_rays * args.steps
        scaling = "weak"
        parallelism = "1 GPU, launches back to back on one stream" if in_flight == 1 else \
                      f"1 GPU, {in_flight} frames in flight on {in_flight} HIP streams"

        def check_headline():
            for st_, o_ in lanes[:min(in_flight, args.warmup + args.steps)]:
                if not np.array_equal(o_.cpu().numpy(), fb_ref):
                    raise SystemExit("bench.py: timed path produced a different frame than hmrm_render_stats")
    elif args.mode == "frames":
        elapsed, total_steps_timed, check_headline = orbit_leg(wl, scene, args.steps, args.warmup, in_flight)
        rays_timed = frame_rays * args.steps * world
        scaling = "weak"
        parallelism = (f"{ORBIT_FRAMES}-frame orbit, frame k on GPU k mod {world} ({args.steps} frames per GPU), "
                       f"no collective" + (f"; {in_flight} frames in flight per GPU" if in_flight > 1 else ""))
    else:
        plan = strips.BandPlan(height=H, width=W, band_rows=BAND_ROWS, world=world)
        strip = torch.zeros((plan.strip_rows, W, 4), dtype=torch.uint8, device="cuda")
        block = torch.zeros((world, plan.strip_rows, W, 4), dtype=torch.uint8, device="cuda") if rank == 0 else None

        def render_rows(strip_t, band_rows, band_index, band_count):
            scene.render_rows_device(cam, strip_t.data_ptr(), W * 4, band_rows=band_rows,
                                     band_index=band_index, band_count=band_count, stream=stream)

        def step():
            result["frame"] = strips.render_frame_distributed(plan, rank, render_rows, dist, strip, block)
        for _ in range(args.warmup):
            step()
        elapsed = timed(step, args.steps)
        total_steps_timed = frame_steps * args.steps  # one frame per step for the whole job
        rays_timed = frame_rays * args.steps
        scaling = "strong"
        parallelism = f"cyclic {BAND_ROWS}-row bands over {world} GPUs + RCCL gather to rank 0"

        def check_headline():
            if rank == 0 and not np.array_equal(result["frame"].cpu().numpy(), fb_ref):
                raise SystemExit("bench.py: timed path produced a different frame than hmrm_render_stats")

    # dominant kernel: mean launch duration by HIP events on the launch stream (full frame, static pose, back to back)
    kernel_ms = scene.bench_kernel_ms(cam, 50) if rank == 0 else None
    # correctness of what was just timed, against the instrumented kernel (another instantiation, host read-back path)
    check_headline()

    secondary = {}
    if not args.no_secondary and not multi:
        # ---- three frames in flight (a throughput mode for sequences of independent frames, hmap.cpp:1131-1144)
        keep3, lanes3 = make_lanes(3)
        nxt3, _ = round_robin(lanes3)

        def step3():
            st_, o_ = nxt3()
            scene.render_rows_device(cam, o_.data_ptr(), W * 4, 0, H, stream=st_)
        for _ in range(50 + args.warmup):
            step3()
        dt3 = timed(step3, args.steps)
        for st_, o_ in lanes3:
            if not np.array_equal(o_.cpu().numpy(), fb_ref):
                raise SystemExit("bench.py: a frame in flight differs from hmrm_render_stats")
        secondary["frames_in_flight"] = {"streams": 3, "ms_per_step": dt3 * 1e3 / args.steps,
                                         "value": frame_steps * args.steps / dt3,
                                         "note": "same K frames round-robin over 3 HIP streams; not the operating point of `roofline`"}
        del keep3, lanes3

        # ---- a moving camera: every timed frame's camera is new to the library (no pre-render), then the same
        # cameras again (per-stream cache of 64 records).  One stream, back to back, like the headline.
        def fresh_leg(the_wl, tag):
            n = min(args.steps, 60)
            out_t = torch.empty((the_wl.height, the_wl.width, 4), dtype=torch.uint8, device="cuda")
            base = the_wl.camera()
            # orbit positions no other part of this run has used (a 1000003-frame orbit, frames 1..n)
            cams = [the_wl.camera(k, 1000003) for k in range(1, n + 1)]
            for _ in range(30):  # steady clocks
                scene.render_rows_device(base, out_t.data_ptr(), the_wl.width * 4, 0, the_wl.height, stream=stream)
            it = iter(cams)

            def step_f():
                scene.render_rows_device(next(it), out_t.data_ptr(), the_wl.width * 4, 0, the_wl.height, stream=stream)
            fresh = timed(step_f, n)
            it = iter(cams)
            cached = timed(step_f, n)
            if not np.array_equal(out_t.cpu().numpy(), scene.render_stats(cams[-1])[0]):
                raise SystemExit(f"bench.py: fresh-camera leg ({tag}) rendered a different frame than hmrm_render_stats")
            return {"frames": n, "ms_per_step": fresh * 1e3 / n, "cached_ms_per_step": cached * 1e3 / n,
                    "fresh_over_cached": fresh / cached}
        fc = fresh_leg(wl, wl.name)
        fc["note"] = ("every timed frame has a camera the library has not seen: per-frame host set-up (libm, spherical "
                      "sin/cos tables on the host pool, upload) inside the loop; then the same cameras from the cache")
        if wl.map_size == 4096 and wl.name != "C5":
            fc["C5"] = fresh_leg(hmrm.synth.WORKLOADS["C5"], "C5")  # (perspective; same maps and scene parameters)
        secondary["fresh_camera"] = fc
        torch.cuda.synchronize()
    elif not args.no_secondary and multi and args.mode == "frames":
        # ---- BASELINE config C5: the recording's camera (perspective), frame k on GPU k mod N; same maps
        if wl.map_size == 4096 and wl.name != "C5":
            wl5 = hmrm.synth.WORKLOADS["C5"]
            n5 = min(args.steps, 64)
            dt5, steps5, check5 = orbit_leg(wl5, scene, n5, min(args.warmup, 8), 1)
            check5()
            secondary["c5_frames"] = {"workload": "C5: 4096^2, 3840x2160, perspective hfov 90, step_dist 0.5, 64-frame orbit, frame k on GPU k mod N",
                                      "frames_per_gpu": n5, "ms_per_step": dt5 * 1e3 / n5, "value": steps5 / dt5,
                                      "unit": "ray-steps/s", "scaling": "weak"}
        # ---- BASELINE config C4: one 7680x4320 orthographic frame over the 8192^2 map in cyclic 16-row bands
        if not args.no_c4:
            scene.close()
            scene = None
            wl4 = hmrm.synth.WORKLOADS["C4"]
            rgb4, cmap4 = hmrm.synth.synth_maps(wl4.map_size)
            cam4 = wl4.camera()
            scene4 = hmrm.Scene(rgb4, cmap4, wl4.scene_params())
            del rgb4, cmap4
            W4, H4 = cam4.width, cam4.height
            fb4, st4, _, _ = scene4.render_stats(cam4)  # (every rank: its own check frame and the step count)
            plan = strips.BandPlan(height=H4, width=W4, band_rows=BAND_ROWS, world=world)
            strip = torch.zeros((plan.strip_rows, W4, 4), dtype=torch.uint8, device="cuda")
            block = torch.zeros((world, plan.strip_rows, W4, 4), dtype=torch.uint8, device="cuda") if rank == 0 else None
            host_strip = torch.zeros((plan.strip_rows, W4, 4), dtype=torch.uint8).pin_memory()

            def render_rows4(strip_t, band_rows, band_index, band_count):
                scene4.render_rows_device(cam4, strip_t.data_ptr(), W4 * 4, band_rows=band_rows,
                                          band_index=band_index, band_count=band_count, stream=stream)
            n4 = max(5, min(args.steps, 40))

            def step_gather():
                result["frame4"] = strips.render_frame_distributed(plan, rank, render_rows4, dist, strip, block)

            def step_own():
                strips.render_strip_to_host(plan, rank, render_rows4, strip, host_strip)

            def step_kernel():
                render_rows4(strip, plan.band_rows, rank, plan.world)
            legs = {}
            for name, fn in (("kernel_only", step_kernel), ("gather_to_rank0_over_rccl", step_gather), ("own_pcie_link_no_collective", step_own)):
                for _ in range(3):
                    fn()
                dt4 = timed(fn, n4)
                legs[name] = {"ms_per_frame": dt4 * 1e3 / n4, "value": int(st4.steps) * n4 / dt4}
            ok4 = strips.strip_rows_match(plan, rank, host_strip.numpy(), fb4)
            if rank == 0:
                ok4 = ok4 and np.array_equal(result["frame4"].cpu().numpy(), fb4)
            if total_over_ranks(0 if ok4 else 1):
                raise SystemExit("bench.py: a c4_strips leg produced different pixels than hmrm_render_stats")
            secondary["c4_strips"] = {"workload": f"C4: 8192^2, {W4}x{H4}, orthographic, step_dist 0.5, cyclic {BAND_ROWS}-row bands over {world} GPUs",
                                      "frames": n4, "ray_steps_per_frame": int(st4.steps), "unit": "ray-steps/s",
                                      "scaling": "strong", **legs}
            scene4.close()

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
            "operating_point": "launches back to back on one stream (the same as `value` at N = 1)",
            # BASELINE.md's nominal figure: bytes the REFERENCE's loop would move for this frame over the
            # measured duration.  Not executed traffic (the loads are skipped), hence not a fraction of a peak.
            "algorithmic_equivalent": {"bytes_per_launch": algo_bytes, "gbs": algo_bytes / kernel_s / 1e9,
                                       "times_hbm_peak": algo_bytes / kernel_s / 1e9 / HBM_PEAK_GBS},
            "pmc": prov,
        }
        if pmc and pmc.get("in_flight"):
            # VALU-busy over the overlapped window of three frames in flight (its own PMC pass, tools/profile_round.sh)
            secondary.setdefault("frames_in_flight", {})["pmc"] = pmc["in_flight"]
        line = {
            "metric": "ray-steps/s at 3840x2160, 4096^2 heightmap" if wl.map_size == 4096 else
                      f"ray-steps/s at {W}x{H}, {wl.map_size}^2 heightmap",
            "value": value, "unit": "ray-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "equivalent_steps": True,
            "frames_in_flight": in_flight,
            "executed_per_frame": {"height_samples": int(st.groups) * 4 if st.groups else None,
                                   "pyramid_lookups": int(st.leap_attempts),
                                   "steps_covered_by_exact_leaps": int(st.leaped_steps)},
            "mrays_per_s": rays_timed / elapsed / 1e6,
            "config": {"workload": f"{wl.name}: {wl.map_size}x{wl.map_size} heightmap, {W}x{H}, "
                                   f"{('perspective', 'spherical', 'orthographic')[wl.projection - 1]} "
                                   f"hfov {wl.hfov_deg:g}, step_dist {wl.step_dist:g}, grid_width 1",
                       "ray_steps_per_frame": frame_steps, "rays_per_frame": frame_rays,
                       "hits_per_frame": frame_hits, "parallelism": parallelism,
                       "maps_sha256": hmrm.synth.maps_sha256(rgb, cmap)[:16]},
            "roofline": roofline,
        }
        if rccl_ranks is not None:
            line["rccl_ranks"] = rccl_ranks
        line.update(secondary)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(hmrm, wl, rgb, cmap, params, cam, args.cpu_seconds)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if scene is not None:
        scene.close()


if __name__ == "__main__":
    main()
