#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X (BASELINE.json).

A "step" is one pass of the hot path over one frame of synthetic input: the
render kernel over every pixel of the workload's framebuffer.  Default workload
at N = 1 = BASELINE.json configs[2], the configuration the metric is quoted on:
3840x2160 over a 4096^2 heightmap, spherical hfov 180, step_dist 0.25 ("C3").

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3]

With --gpus N > 1 and no WORLD_SIZE in the environment this process is only a LAUNCHER: before anything
touches the GPU (torch is not even imported) it starts `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same flags>` as a CHILD process,
relays rank 0's one JSON line and exits with the child's code.  Launched under torch.distributed.run
directly (WORLD_SIZE set: what the driver does) the same file is a rank.  `--dry-launch` runs launcher,
rendezvous, sharding and the gather on CPU (gloo, a pattern instead of the renderer): the CPU test of
the launch path (tests/test_bench_launch.py); its line says "dry_launch": true and carries no value.
`--share-gpu` is the rehearsal of the N > 1 line on a box with fewer GPUs than ranks: the ranks share the GPUs that are
there, the collectives run over gloo on host copies, every block and pixel check of the line runs
("share_gpu": true; the rates are no scaling figures).

N = 1: `value` / `ms_per_step` = K frames of the workload's static pose launched BACK TO BACK ON ONE
       STREAM, maps resident in HBM, output to a device buffer -- the operating point `roofline`
       describes (`roofline.kernel_ms` = the same launches timed with HIP events on the launch stream).
       Before the warm-up the pose is rendered 12 times with a synchronisation in between so that the
       library's launch-order calibration has settled (`preconditioning`).
       Secondary blocks in the same line (never `value`):
         value_moving_camera  ray-steps/s of `fresh_camera`'s timed frames: every frame a camera the
                           library has never seen (host set-up inside the loop) -- what a moving
                           camera gets; the headline's static pose is the best case;
         frames_in_flight  the same K frames round-robin over 3 HIP streams;
         fresh_camera      the moving-camera leg's timings (fresh, then the same cameras cached), also for C5;
         workloads         the other BASELINE configurations on this GPU, each with kernel_ms (HIP
                           events), ms_per_step (host clock, one stream), Mrays/s, reference-equivalent
                           and executed steps: C3h (the north_star target: step_dist 0.5), C5 (static
                           pose), C2, C4 (7680x4320 over 8192^2, the whole frame on one GPU), and the reference's
                           own operating points: C3/gw0.05 (its default grid_width) and REFDEF (its sample config);
         operating_range   best / moving camera / reference defaults / worst content, ms per frame at a glance;
         literal_kernel    HMRM_KERNEL=simple on the headline frame: the kernel that executes every
                           one of the reference's loads, with SURVEY 8(d)'s algorithmic-bytes fraction
                           of the HBM peak (the figure that formula is defined for);
         every_load_kernel HMRM_KERNEL=group on the headline frame: the fastest kernel that still executes
                           every one of those loads (speculative groups, no leaps), same byte roofline;
         rough_terrain     the headline camera over maps built to defeat the traversal (white noise,
                           a 255-spike per 256^2 block, needles on a plateau, a canyon flown at low
                           altitude): the library as shipped against the plain speculative groups.
N > 1: one process per GPU over RCCL, maps replicated.  `value` = frames of BASELINE configs[4]'s
       64-frame orbit ("C5"; `--workload` overrides), frame k on GPU k mod N, K frames per GPU on one
       stream each, no data-path collective, scaling "weak" (`per_rank_ms_per_step`: every rank's own time; `ms_per_step` is
       their maximum).  The same line carries `rccl_ranks`
       (distinct (host, device, bus) triples seen by an all_gather), `one_gpu_same_leg` (rank 0 alone
       rendering K frames of the same orbit in the same run) with the speed-up and efficiency against
       it, `static_pose_replicas` (the N = 1 line's own step on every GPU: comparable with that line's
       `value`) and `c4_strips`: ONE 7680x4320 frame over the 8192^2 map (BASELINE configs[3]) in
       cyclic 16-row bands, STRONG scaling -- kernel only, strips gathered to rank 0 over RCCL, every
       rank copying its strip to its own pinned host memory (no collective) -- each with its time on
       one GPU measured in the same run and the speed-up / efficiency against it.
       `--mode strips` makes the gather variant of the workload `value` instead (scaling "strong").

Rank 0 prints ONE JSON line.  value = REFERENCE-EQUIVALENT ray-steps/s: a ray-step is
one execution of the reference's height load (main/hmap.cpp:1013-1014) and the count
is what the reference executes for the same frame (instrumented kernel variant,
bit-identical pixels, run once, untimed; equal to the oracle's count).  The
production kernel proves most of those loads unnecessary and skips them, so the
line also says `"equivalent_steps": true` and carries the height samples and
pyramid look-ups the kernel really executes; `mrays_per_s` is the algorithm-neutral rate.

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
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
SIMDS, PEAK_CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs; max clock (MI355X_MICROARCH.md)
BAND_ROWS = 16
ORBIT_FRAMES = 64  # BASELINE config C5
PRECONDITION_LAUNCHES = 12
PROJ_NAMES = ("perspective", "spherical", "orthographic")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, help="default: C3 at N = 1, C5 (BASELINE configs[4]) at N > 1")
    ap.add_argument("--mode", choices=["frames", "strips"], default="frames")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="streams the HEADLINE loop sends consecutive frames to (1 = back to back on one stream, the "
                         "operating point of `roofline`; the 3-stream figure is always reported as a secondary block)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline only: skip every secondary block")
    ap.add_argument("--no-c4", action="store_true", help="skip the blocks that need the 8192^2 maps (workloads.C4, c4_strips)")
    ap.add_argument("--no-rough", action="store_true", help="N = 1: skip the rough_terrain block")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the N > 1 line on a box with fewer GPUs than ranks: every rank renders on GPU (local rank mod "
                         "visible devices), collectives run over gloo on host copies.  Every block and every pixel check of the N > 1 "
                         "line runs; the rates say nothing about scaling (the line carries \"share_gpu\": true)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="CPU rehearsal of the launch path: gloo instead of RCCL, a pattern instead of the renderer, no GPU touched")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """--gpus N > 1 without a torch.distributed environment: start the N ranks as a CHILD process (never an exec:
    nothing in this process has touched the GPU, and nothing will), relay rank 0's JSON line, return its exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on this pool)
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)  # (stderr passes through)
    lines = [ln for ln in proc.stdout.decode("utf-8", "replace").splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank job failed (exit code {proc.returncode}, "
                         f"{len(lines)} JSON line(s) on its stdout)\n")
        return proc.returncode or 1
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()
    return 0


def _host_id():
    return int.from_bytes(hashlib.sha256(socket.gethostname().encode()).digest()[:4], "little") & 0x7fffffff


class Job:
    """Rank, world and the timing harness of the contract: barrier + synchronize on both sides of the timed region,
    maximum over ranks."""

    def __init__(self, args, torch):
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dry = args.dry_launch
        self.share_gpu = args.share_gpu and not args.dry_launch
        self.device = "cpu" if (self.dry or self.share_gpu) else "cuda"  # where the tensors of the collectives live
        # HMRM_FORCE_DIST=1 runs the torch.distributed (RCCL) code path even with one rank, so that
        # the N>1 plumbing can be exercised on a single-GPU box
        self.multi = self.world > 1 or os.environ.get("HMRM_FORCE_DIST", "") == "1"
        self.dist = None
        self.rccl_ranks = None
        self.last_rank_seconds = []

    def init_dist(self):
        torch = self.torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if self.dry or self.share_gpu:
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            me = torch.tensor([_host_id(), torch.cuda.current_device() if self.share_gpu else self.rank, 0], dtype=torch.int64)
        else:
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                    device_id=torch.device("cuda", self.local_rank))
            # did RCCL see `world` ranks on `world` different GPUs?  Every rank contributes (a stable host id,
            # device index, PCI bus id); the gathered rows are counted on rank 0.
            props = torch.cuda.get_device_properties(self.local_rank)
            bus = getattr(props, "pci_bus_id", self.local_rank)
            dom = getattr(props, "pci_domain_id", 0)
            me = torch.tensor([_host_id(), torch.cuda.current_device(), (int(dom) << 16 | int(bus)) & 0x7fffffff],
                              dtype=torch.int64, device="cuda")
        self.dist = dist
        seen = [torch.zeros_like(me) for _ in range(self.world)]
        dist.all_gather(seen, me)
        self.rccl_ranks = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                           "distinct_devices": len({tuple(int(v) for v in t.tolist()) for t in seen})}

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if not self.dry:
            self.torch.cuda.synchronize()

    def timed(self, step, n):
        """n calls of step() bracketed by barrier + synchronize; max over ranks -> seconds."""
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        self.barrier()
        dt = time.perf_counter() - t0
        self.last_rank_seconds = [dt]
        if self.dist is not None:
            # every rank's own time (the contract's figure is their maximum; the list shows a straggler when there is one)
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.device)
            every = [self.torch.zeros_like(t) for _ in range(self.world)]
            self.dist.all_gather(every, t)
            self.last_rank_seconds = [float(v.item()) for v in every]
            dt = max(self.last_rank_seconds)
        return dt

    def total(self, v):
        if self.dist is None:
            return int(v)
        n = self.torch.tensor([int(v)], dtype=self.torch.int64, device=self.device)
        self.dist.all_reduce(n, op=self.dist.ReduceOp.SUM)
        return int(n.item())

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


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
    prov["status"] = ("ok: collected on these kernel sources (separate rocprofv3 --pmc passes on another box / run than this "
                      "one's kernel_ms; boxes of the pool differ by +-3 %)")
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


def workload_text(wl):
    return (f"{wl.name}: {wl.map_size}x{wl.map_size} heightmap, {wl.width}x{wl.height}, {PROJ_NAMES[wl.projection - 1]} "
            f"hfov {wl.hfov_deg:g}, step_dist {wl.step_dist:g}" + (" cells" if wl.grid_width != 1.0 else "") + f", grid_width {wl.grid_width:g}"
            + ("" if wl.content == "smooth" else f", {wl.content} map"))


# ------------------------------------------------------------------------------------------ dry launch (CPU)
def dry_main(args, job):
    """The N > 1 control flow on CPU: gloo, a deterministic pattern instead of the renderer.  Exercises the launcher,
    the rendezvous, the timing harness, the orbit sharding, the band plan and the gather; measures nothing."""
    import numpy as np
    torch = job.torch
    strips = importlib.import_module("heightmap-ray-marcher_amd.strips")
    if job.multi:
        job.init_dist()
    rank, world = job.rank, job.world
    W, H = 64, 100

    def pattern_rows(rows, frame=0):
        r = np.asarray(rows, dtype=np.int64)[:, None, None]
        x = np.arange(W, dtype=np.int64)[None, :, None]
        c = np.arange(4, dtype=np.int64)[None, None, :]
        return ((r * 131 + x * 7 + c * 3 + frame * 17) & 255).astype(np.uint8)

    # frames: frame k on rank k mod world
    mine = strips.orbit_frames_of_rank(rank, world, args.warmup + args.steps, ORBIT_FRAMES)
    it = iter(mine)
    rendered = []

    def step_frames():
        k = next(it)
        rendered.append(int(pattern_rows(range(H), k).sum()))
    for _ in range(args.warmup):
        step_frames()
    dt_frames = job.timed(step_frames, args.steps)
    frames_total = job.total(args.steps)
    # strips: cyclic bands + gather to rank 0
    plan = strips.BandPlan(height=H, width=W, band_rows=BAND_ROWS, world=world)
    strip = torch.zeros((plan.strip_rows, W, 4), dtype=torch.uint8)
    block = torch.zeros((world, plan.strip_rows, W, 4), dtype=torch.uint8) if (rank == 0 and job.dist is not None) else None
    out = {}

    def render_rows(strip_t, band_rows, band_index, band_count):
        rows = plan.rows_of(band_index)
        host = strip_t.numpy()
        host[:] = 0
        k = 0
        for b in plan.bands_of(band_index):
            g0, g1 = b * band_rows, min((b + 1) * band_rows, H)
            host[k * band_rows:k * band_rows + (g1 - g0)] = pattern_rows(range(g0, g1))
            k += 1
        assert len(rows) <= plan.strip_rows

    def step_strips():
        out["frame"] = strips.render_frame_distributed(plan, rank, render_rows, job.dist, strip, block)
    dt_strips = job.timed(step_strips, max(1, min(args.steps, 3)))
    ok = True
    if rank == 0:
        ok = bool(np.array_equal(out["frame"].numpy(), pattern_rows(range(H))))
    # the pipelined sequence (double-buffered strips, chunked gathers, rotating root) with the same pattern
    pipe = strips.StripPipeline(plan, rank, job.dist if world > 1 else None, torch, "cpu", depth=2, chunks=2, rotate_root=True)

    def render_chunk_of(k):
        def render_chunk(chunk_t, band_rows, band_index, band_count):
            vplan = strips.BandPlan(height=H, width=W, band_rows=band_rows, world=band_count)
            host = chunk_t.numpy()
            host[:] = 0
            for j, b in enumerate(vplan.bands_of(band_index)):
                g0, g1 = b * band_rows, min((b + 1) * band_rows, H)
                host[j * band_rows:j * band_rows + (g1 - g0)] = pattern_rows(range(g0, g1), k)
        return render_chunk
    piped = {}
    pipe.run(range(5), render_chunk_of, on_frame=lambda k, f: piped.__setitem__(k, f))
    for k in range(5):
        if rank == k % world:
            ok = ok and piped[k] is not None and bool(np.array_equal(piped[k].numpy(), pattern_rows(range(H), k)))
        else:
            ok = ok and piped[k] is None
    bad = job.total(0 if ok else 1)
    if rank == 0:
        if bad:
            raise SystemExit("bench.py --dry-launch: the gathered frame differs from the pattern")
        line = {"metric": "ray-steps/s at 3840x2160, 4096^2 heightmap", "value": None, "unit": "ray-steps/s",
                "dry_launch": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": dt_frames * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "dry launch: no renderer, no GPU (gloo rehearsal of the launch path)",
                           "parallelism": f"frame k on rank k mod {world}; cyclic {BAND_ROWS}-row bands + gather to rank 0"},
                "rccl_ranks": job.rccl_ranks, "frames_rendered_by_all_ranks": frames_total,
                "strips_gather_ms": dt_strips * 1e3 / max(1, min(args.steps, 3))}
        os.write(job.json_fd, (json.dumps(line) + "\n").encode())
    job.finish()
    return 0


# ------------------------------------------------------------------------------------------------ real ranks
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)

    import numpy as np
    import torch

    # stdout carries exactly one line, the JSON: libraries that print banners there (RCCL's version
    # block at communicator creation) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    job = Job(args, torch)
    job.json_fd = json_fd
    if job.world != args.gpus:
        args.gpus = job.world  # (launched under torch.distributed.run with another --nproc-per-node: the environment wins)
    if args.dry_launch:
        return dry_main(args, job)
    world, rank = job.world, job.rank

    hmrm = importlib.import_module("heightmap-ray-marcher_amd")
    strips = importlib.import_module("heightmap-ray-marcher_amd.strips")
    n_dev = hmrm.device_count() if torch.cuda.is_available() else 0
    if n_dev < 1:
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if job.local_rank >= n_dev and not job.share_gpu:
        raise SystemExit(f"bench.py: rank {rank} has no GPU of its own (local rank {job.local_rank}, {n_dev} device(s) visible)")
    torch.cuda.set_device(job.local_rank % n_dev)
    hmrm.set_device(job.local_rank % n_dev)
    multi = job.multi
    if multi:
        job.init_dist()
    timed, total_over_ranks = job.timed, job.total
    synth = hmrm.synth

    wl_name = args.workload or ("C5" if (multi and args.mode == "frames") else "C3")
    wl = synth.WORKLOADS[wl_name]
    maps_cache = {}

    def maps_of(w):
        key = (w.map_size, w.content)
        if key not in maps_cache:
            maps_cache[key] = w.maps()
        return maps_cache[key]
    rgb, cmap = maps_of(wl)
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

    def precondition(the_scene, the_cam):
        """Untimed: the library calibrates the launch order of a camera it sees repeatedly from a few measured launches
        (csrc/launch_order.cpp; each needs the previous one finished, which a renderer that presents its frames does by
        itself and a loop that queues hundreds of launches ahead does not): a dozen launches of the pose with a
        synchronisation in between, then 50 more so that the GPU enters the timed region at steady clocks."""
        for _ in range(PRECONDITION_LAUNCHES):
            the_scene.bench_kernel_ms(the_cam, 1)
        the_scene.bench_kernel_ms(the_cam, 50)

    def static_leg(the_scene, the_cam, n_frames, n_warm, in_flight, ref_frame, active=True):
        """n_frames launches of one pose on `in_flight` streams (1 = back to back), timed by the contract's harness.
        Ranks with active=False take part in the barriers only.  -> (seconds (max over ranks), check callable: the
        frames just timed against the instrumented kernel's -- called after any HIP-event timing that should see the
        GPU in the same state, the read-back idles it)."""
        if not active:
            return timed(lambda: None, n_frames), lambda: None
        keep, lanes = make_lanes(in_flight, the_cam.width, the_cam.height)
        nxt, _ = round_robin(lanes)

        def step():
            st_, o_ = nxt()
            the_scene.render_rows_device(the_cam, o_.data_ptr(), the_cam.width * 4, 0, the_cam.height, stream=st_)
        for _ in range(n_warm):
            step()
        dt = timed(step, n_frames)

        def check():
            for st_, o_ in lanes[:min(in_flight, n_warm + n_frames)]:
                if not np.array_equal(o_.cpu().numpy(), ref_frame):
                    raise SystemExit("bench.py: a timed static-pose leg produced a different frame than hmrm_render_stats")
            keep.clear()
        return dt, check

    def orbit_leg(the_wl, the_scene, n_frames, n_warm, in_flight, solo=False):
        """Frame k of the workload's 64-frame orbit on rank k mod world, n_frames timed per rank after n_warm untimed;
        cameras pre-rendered once (cached host set-up: this leg measures the GPU path).  solo: rank 0 alone renders
        frames 0, 1, 2, .. (the one-GPU reference of the same leg); the others only stand in the barriers.
        -> (seconds, total steps, check callable)."""
        if solo and rank != 0:
            return timed(lambda: None, n_frames), total_over_ranks(0), lambda: None
        mine = strips.orbit_frames_of_rank(0 if solo else rank, 1 if solo else world, n_warm + n_frames, ORBIT_FRAMES)
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

    def frame_over_ranks(plan, the_rank, render_rows_fn, dist_or_none, strip, block):
        """strips.render_frame_distributed; under --share-gpu the gather runs over gloo on host copies of the strips."""
        if dist_or_none is None or not job.share_gpu:
            return strips.render_frame_distributed(plan, the_rank, render_rows_fn, dist_or_none, strip, block)
        render_rows_fn(strip, plan.band_rows, the_rank, plan.world)
        host = strip.cpu()
        parts = [torch.empty_like(host) for _ in range(plan.world)] if the_rank == 0 else None
        dist_or_none.gather(host, gather_list=parts, dst=0)
        if the_rank != 0:
            return None
        return strips.reassemble_torch(plan, torch.stack(parts, 0))

    def strips_block(the_wl, the_scene, n_frames):
        """ONE frame of the_wl in cyclic 16-row bands over the ranks (BASELINE configs[3]'s sharding): kernel only, strips
        gathered to rank 0 over RCCL, every rank's strip to its own pinned host memory; and the same three on ONE GPU
        (rank 0 renders the whole frame while the others wait) for the speed-up of this run."""
        cam4 = the_wl.camera()
        W4, H4 = cam4.width, cam4.height
        fb4, st4, _, _ = the_scene.render_stats(cam4)  # (every rank: its own check frame and the step count)
        steps4 = int(st4.steps)
        result = {}

        def legs_for(plan, the_rank, active):
            strip = torch.zeros((plan.strip_rows, W4, 4), dtype=torch.uint8, device="cuda") if active else None
            block = torch.zeros((plan.world, plan.strip_rows, W4, 4), dtype=torch.uint8, device="cuda") if (active and the_rank == 0 and plan.world > 1) else None
            host_strip = torch.zeros((plan.strip_rows, W4, 4), dtype=torch.uint8).pin_memory() if active else None

            def render_rows4(strip_t, band_rows, band_index, band_count):
                the_scene.render_rows_device(cam4, strip_t.data_ptr(), W4 * 4, band_rows=band_rows,
                                             band_index=band_index, band_count=band_count, stream=stream)

            def step_kernel():
                render_rows4(strip, plan.band_rows, the_rank, plan.world)

            def step_gather():
                result["frame4"] = frame_over_ranks(plan, the_rank, render_rows4, job.dist if plan.world > 1 else None, strip, block)

            def step_own():
                strips.render_strip_to_host(plan, the_rank, render_rows4, strip, host_strip)
            # the pipelined sequence (strips.StripPipeline): two strip buffers per rank, the gather on its own stream; to rank 0,
            # and with the root rotating (frame k on GPU k mod N).  One band set per frame: rendering a rank's strip in 4
            # interleaved sets (chunks=4, so that the root starts receiving earlier) costs 4 launches with 4 tails -- measured on
            # one GPU 0.94 against 0.60 ms per C4 frame -- and buys latency only, which a pipelined sequence does not need.
            pipes = {}

            def make_pipe(rotate):
                if not active:
                    return None
                use_dist = job.dist if plan.world > 1 else None
                if use_dist is not None and job.share_gpu:  # (rehearsal: gloo gathers host tensors)
                    return strips.StripPipeline(plan, the_rank, use_dist, torch, "cpu", depth=2, chunks=1, rotate_root=rotate)
                return strips.StripPipeline(plan, the_rank, use_dist, torch, "cuda", depth=2, chunks=1, rotate_root=rotate,
                                            render_stream=torch.cuda.current_stream() if use_dist is not None else None,
                                            comm_stream=torch.cuda.Stream() if use_dist is not None else None)

            def render_chunk(chunk_t, band_rows, band_index, band_count):
                if chunk_t.is_cuda:
                    render_rows4(chunk_t, band_rows, band_index, band_count)
                else:  # (share-gpu rehearsal: render on the GPU that is there, hand the gather a host copy)
                    tmp = torch.empty(chunk_t.shape, dtype=torch.uint8, device="cuda")
                    render_rows4(tmp, band_rows, band_index, band_count)
                    chunk_t.copy_(tmp)

            def pipe_step(tag):
                st_ = pipes[tag]

                def step():
                    # (this leg has never run over RCCL with more than one rank -- no multi-GPU box in rounds 1-5: an error in it
                    # is recorded in the leg and must not take the rest of the line down)
                    if st_.get("error"):
                        return
                    try:
                        pipe, pending = st_["pipe"], st_["pending"]
                        if len(pending) == pipe.depth:
                            j = pending.pop(0)
                            f = pipe.collect(j)
                            if f is not None:
                                st_["last"] = f
                        pipe.submit(st_["k"], render_chunk)
                        pending.append(st_["k"])
                        st_["k"] += 1
                    except Exception as e:  # noqa: BLE001
                        st_["error"] = f"{type(e).__name__}: {e}"
                return step

            def pipe_drain(tag):
                st_ = pipes[tag]
                try:
                    for j in st_["pending"]:
                        f = st_["pipe"].collect(j)
                        if f is not None:
                            st_["last"] = f
                except Exception as e:  # noqa: BLE001
                    st_["error"] = st_.get("error") or f"{type(e).__name__}: {e}"
                st_["pending"] = []
            for tag, rotate in (("pipelined_to_rank0", False), ("pipelined_rotating_root", True)):
                pipes[tag] = {"pipe": make_pipe(rotate), "pending": [], "k": 0, "last": None}
            legs = {}
            for name, fn in (("kernel_only", step_kernel), ("gather_to_rank0_over_rccl", step_gather), ("own_pcie_link_no_collective", step_own),
                             ("pipelined_to_rank0", pipe_step("pipelined_to_rank0")), ("pipelined_rotating_root", pipe_step("pipelined_rotating_root"))):
                if active:
                    for _ in range(3):
                        fn()
                dt4 = timed(fn if active else (lambda: None), n_frames)
                if active and name in pipes:
                    pipe_drain(name)
                legs[name] = {"ms_per_frame": dt4 * 1e3 / n_frames, "value": steps4 * n_frames / dt4}
            for tag in pipes:
                if pipes[tag].get("error"):
                    continue
                legs[tag]["note"] = ("strips.StripPipeline: 2 frames in flight (double-buffered strips, gather on its own stream); "
                                     "throughput of a SEQUENCE of frames, each still rendered by all ranks" +
                                     ("; frame k is gathered to rank k mod N" if tag.endswith("root") else ""))
            ok = True
            if active:
                ok = strips.strip_rows_match(plan, the_rank, host_strip.numpy(), fb4)
                if the_rank == 0:
                    ok = ok and np.array_equal(result["frame4"].cpu().numpy(), fb4)
                for tag in pipes:  # (every rank that was a root of some frame checks the last one it received)
                    if pipes[tag].get("error"):
                        legs[tag] = {"error": pipes[tag]["error"]}
                    elif pipes[tag]["last"] is not None:
                        ok = ok and np.array_equal(pipes[tag]["last"].cpu().numpy(), fb4)
                    elif the_rank == 0:
                        ok = False
            if total_over_ranks(0 if ok else 1):
                raise SystemExit("bench.py: a c4_strips leg produced different pixels than hmrm_render_stats")
            return legs
        legs_n = legs_for(strips.BandPlan(height=H4, width=W4, band_rows=BAND_ROWS, world=world), rank, True)
        block = {"workload": workload_text(the_wl) + f"; cyclic {BAND_ROWS}-row bands over {world} GPU(s)",
                 "frames": n_frames, "ray_steps_per_frame": steps4, "unit": "ray-steps/s", "scaling": "strong"}
        if world > 1:
            legs_1 = legs_for(strips.BandPlan(height=H4, width=W4, band_rows=BAND_ROWS, world=1), 0, rank == 0)
            legs_1["gather_to_rank0_over_rccl"]["note"] = "one GPU: no gather, the strip is the frame (reassembly only)"
            block["one_gpu_same_run"] = legs_1
            for name, leg in legs_n.items():
                if "ms_per_frame" not in leg or "ms_per_frame" not in legs_1.get(name, {}):
                    continue
                leg["speedup_vs_one_gpu"] = legs_1[name]["ms_per_frame"] / leg["ms_per_frame"]
                leg["efficiency"] = leg["speedup_vs_one_gpu"] / world
        block.update(legs_n)
        return block

    def executed(st_):
        # (positions per speculative group: 4 in the production kernel, 6 in the plain-groups kernel -- render_fast.hip kGroup / kGroupPlain)
        return {"height_samples": int(st_.groups) * (4 if st_.leap_attempts else 6) if st_.groups else None, "pyramid_lookups": int(st_.leap_attempts),
                "steps_covered_by_exact_leaps": int(st_.leaped_steps)}

    def workload_block(w, the_scene, n_frames, n_warm):
        """One more BASELINE configuration on this GPU: static pose, one stream, like the headline."""
        c = w.camera()
        fb, s_, _, _ = the_scene.render_stats(c)
        precondition(the_scene, c)
        # (the shorter of two repetitions: these legs are a few milliseconds long, and a single host hiccup -- one was seen: 58 ms
        # inside a 3 ms leg -- would otherwise be the number)
        dt, check = static_leg(the_scene, c, n_frames, n_warm, 1, fb)
        dt_b, check_b = static_leg(the_scene, c, n_frames, n_warm, 1, fb)
        dt = min(dt, dt_b)
        kms = the_scene.bench_kernel_ms(c, min(50, max(10, n_frames)))  # (straight behind the timed launches: same clocks)
        check()
        check_b()
        return {"workload": workload_text(w), "frames": n_frames, "ms_per_step": dt * 1e3 / n_frames, "repetitions": 2, "kernel_ms": kms,
                "value": int(s_.steps) * n_frames / dt, "unit": "ray-steps/s", "equivalent_steps": True,
                "mrays_per_s": int(s_.rays) * n_frames / dt / 1e6, "ray_steps_per_frame": int(s_.steps),
                "rays_per_frame": int(s_.rays), "hits_per_frame": int(s_.hits), "executed_per_frame": executed(s_)}

    def with_kernel(name, fn):
        """Run fn() with HMRM_KERNEL=name (the library re-reads the knob for live scenes; launch orders calibrated for
        another variant are forgotten), then restore."""
        old = os.environ.get("HMRM_KERNEL")
        os.environ["HMRM_KERNEL"] = name
        try:
            return fn()
        finally:
            if old is None:
                os.environ.pop("HMRM_KERNEL", None)
            else:
                os.environ["HMRM_KERNEL"] = old

    precondition(scene, cam)

    result = {}
    extra = {}
    in_flight = max(1, min(args.frames_in_flight, 4))
    if not multi:
        elapsed, check_headline = static_leg(scene, cam, args.steps, args.warmup, in_flight, fb_ref)
        total_steps_timed = frame_steps * args.steps
        rays_timed = frame_rays * args.steps
        scaling = "weak"
        parallelism = "1 GPU, launches back to back on one stream" if in_flight == 1 else \
                      f"1 GPU, {in_flight} frames in flight on {in_flight} HIP streams"
    elif args.mode == "frames":
        elapsed, total_steps_timed, check_orbit = orbit_leg(wl, scene, args.steps, args.warmup, in_flight)
        extra["per_rank_ms_per_step"] = [round(v * 1e3 / args.steps, 5) for v in job.last_rank_seconds]
        check_orbit()
        check_headline = lambda: None  # noqa: E731
        rays_timed = frame_rays * args.steps * world
        scaling = "weak"
        parallelism = (f"{ORBIT_FRAMES}-frame orbit, frame k on GPU k mod {world} ({args.steps} frames per GPU), "
                       f"no collective" + (f"; {in_flight} frames in flight per GPU" if in_flight > 1 else ""))
        # the same leg on ONE GPU, same run: rank 0 renders frames 0 .. K-1 of the orbit alone
        dt1, steps1, check1 = orbit_leg(wl, scene, args.steps, min(args.warmup, 8), in_flight, solo=True)
        check1()
        extra["one_gpu_same_leg"] = {"ms_per_step": dt1 * 1e3 / args.steps, "value": steps1 / dt1, "frames": args.steps,
                                     "note": "rank 0 alone renders frames 0 .. K-1 of the same orbit while the other ranks wait"}
        extra["speedup_vs_one_gpu"] = (total_steps_timed / elapsed) / (steps1 / dt1)
        extra["efficiency_vs_one_gpu"] = extra["speedup_vs_one_gpu"] / world
    else:
        plan = strips.BandPlan(height=H, width=W, band_rows=BAND_ROWS, world=world)
        strip = torch.zeros((plan.strip_rows, W, 4), dtype=torch.uint8, device="cuda")
        block = torch.zeros((world, plan.strip_rows, W, 4), dtype=torch.uint8, device="cuda") if rank == 0 else None

        def render_rows(strip_t, band_rows, band_index, band_count):
            scene.render_rows_device(cam, strip_t.data_ptr(), W * 4, band_rows=band_rows,
                                     band_index=band_index, band_count=band_count, stream=stream)

        def step():
            result["frame"] = frame_over_ranks(plan, rank, render_rows, job.dist, strip, block)
        for _ in range(args.warmup):
            step()
        elapsed = timed(step, args.steps)
        total_steps_timed = frame_steps * args.steps  # one frame per step for the whole job
        rays_timed = frame_rays * args.steps
        scaling = "strong"
        parallelism = f"cyclic {BAND_ROWS}-row bands over {world} GPUs + RCCL gather to rank 0"
        if rank == 0 and not np.array_equal(result["frame"].cpu().numpy(), fb_ref):
            raise SystemExit("bench.py: timed path produced a different frame than hmrm_render_stats")
        check_headline = lambda: None  # noqa: E731

    # dominant kernel: mean launch duration by HIP events on the launch stream (full frame, static pose, back to back)
    kernel_ms = scene.bench_kernel_ms(cam, 50) if rank == 0 else None
    # correctness of what was just timed, against the instrumented kernel (another instantiation, host read-back path)
    check_headline()

    secondary = {}

    def guarded(name, fn):
        """A secondary block must not void the headline: an error of the runtime (out of memory, a refused call) is recorded
        in the line under `secondary_errors`; a WRONG PIXEL is a SystemExit and still ends the run."""
        try:
            fn()
        except Exception as e:  # noqa: BLE001  (SystemExit is not an Exception)
            secondary.setdefault("secondary_errors", {})[name] = f"{type(e).__name__}: {e}"

    if not args.no_secondary and not multi:
        # ---- three frames in flight (a throughput mode for sequences of independent frames, hmap.cpp:1131-1144)
        def blk_in_flight():
            dt3, check3 = static_leg(scene, cam, args.steps, 50 + args.warmup, 3, fb_ref)
            check3()
            secondary["frames_in_flight"] = {"streams": 3, "ms_per_step": dt3 * 1e3 / args.steps,
                                             "value": frame_steps * args.steps / dt3,
                                             "note": "same K frames round-robin over 3 HIP streams; not the operating point of `roofline`"}
        guarded("frames_in_flight", blk_in_flight)

        # ---- a moving camera: every timed frame's camera is new to the library (no pre-render), then the same
        # cameras again (per-stream cache of 64 records).  One stream, back to back, like the headline.
        def fresh_leg(the_wl, the_scene, tag):
            n = min(args.steps, 60)
            out_t = torch.empty((the_wl.height, the_wl.width, 4), dtype=torch.uint8, device="cuda")
            base = the_wl.camera()
            # orbit positions no other part of this run has used (a 1000003-frame orbit, frames 1..n)
            cams = [the_wl.camera(k, 1000003) for k in range(1, n + 1)]
            for _ in range(30):  # steady clocks
                the_scene.render_rows_device(base, out_t.data_ptr(), the_wl.width * 4, 0, the_wl.height, stream=stream)
            it = iter(cams)

            def step_f():
                the_scene.render_rows_device(next(it), out_t.data_ptr(), the_wl.width * 4, 0, the_wl.height, stream=stream)
            fresh = timed(step_f, n)
            it = iter(cams)
            cached = timed(step_f, n)
            last_fb, last_st, _, _ = the_scene.render_stats(cams[-1])
            if not np.array_equal(out_t.cpu().numpy(), last_fb):
                raise SystemExit(f"bench.py: fresh-camera leg ({tag}) rendered a different frame than hmrm_render_stats")
            steps_all = sum(int(the_scene.render_stats(c)[1].steps) for c in cams)  # (after the timing: untimed, instrumented)
            return {"frames": n, "ms_per_step": fresh * 1e3 / n, "cached_ms_per_step": cached * 1e3 / n,
                    "fresh_over_cached": fresh / cached, "value": steps_all / fresh, "unit": "ray-steps/s",
                    "mrays_per_s": n * the_wl.width * the_wl.height / fresh / 1e6}
        def blk_fresh():
            fc = fresh_leg(wl, scene, wl.name)
            fc["note"] = ("every timed frame has a camera the library has not seen: per-frame host set-up (libm, spherical "
                          "sin/cos tables on the host pool, upload) inside the loop; then the same cameras from the cache")
            extra["value_moving_camera"] = fc["value"]
            if wl.map_size == 4096 and wl.content == "smooth" and wl.name != "C5":
                fc["C5"] = fresh_leg(synth.WORKLOADS["C5"], scene, "C5")  # (perspective; same maps and scene parameters)
            secondary["fresh_camera"] = fc
        guarded("fresh_camera", blk_fresh)

        # ---- the kernel that executes every one of the reference's loads, on the headline frame: the one kernel
        # SURVEY 8(d)'s algorithmic-bytes fraction is defined for
        def blk_literal():
            def literal():
                scene.bench_kernel_ms(cam, 2)
                return scene.bench_kernel_ms(cam, 3)
            lit_ms = with_kernel("simple", literal)
            lit_pmc, lit_prov = _pmc_from_profiles(wl.name + "_literal", hmrm.kernel_src_sha())
            lit_traffic = lit_pmc.get("hbm_bytes_per_launch") if lit_pmc else None
            secondary["literal_kernel"] = {
                "kernel": "k_render (HMRM_KERNEL=simple): main/hmap.cpp:1000-1038 as written, one dependent 8-byte load per ray-step",
                "launches": 3, "kernel_ms": lit_ms, "value": frame_steps / (lit_ms * 1e-3), "unit": "ray-steps/s (executed, not equivalent)",
                "production_speedup": lit_ms / kernel_ms,
                "roofline": {"bound": "hbm", "achieved": algo_bytes / (lit_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": algo_bytes / (lit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo_bytes,
                             "traffic": lit_traffic,
                             "traffic_over_algorithmic": (lit_traffic / algo_bytes) if lit_traffic else None, "pmc": lit_prov}}
        guarded("literal_kernel", blk_literal)

        # ---- ... and the fastest kernel of the library that still executes every one of those loads: the speculative groups
        # without leaps (what a scene renders with when its content admits no jumps).  Same byte roofline.
        def blk_every_load():
            def group():
                scene.bench_kernel_ms(cam, 3)
                return scene.bench_kernel_ms(cam, 10)
            grp_ms = with_kernel("group", group)
            grp_pmc, grp_prov = _pmc_from_profiles(wl.name + "_group", hmrm.kernel_src_sha())
            grp_traffic = grp_pmc.get("hbm_bytes_per_launch") if grp_pmc else None
            secondary["every_load_kernel"] = {
                "kernel": "k_render_fast<.., LEAP = false> (HMRM_KERNEL=group): groups of 6 speculative positions, their height loads issued "
                          "together, tests resolved in order; every load of main/hmap.cpp:1013 is executed, none is proved away",
                "launches": 10, "kernel_ms": grp_ms, "value": frame_steps / (grp_ms * 1e-3), "unit": "ray-steps/s (executed, not equivalent)",
                "production_speedup": grp_ms / kernel_ms,
                "roofline": {"bound": "hbm", "achieved": algo_bytes / (grp_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": algo_bytes / (grp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo_bytes,
                             "traffic": grp_traffic,
                             "traffic_over_algorithmic": (grp_traffic / algo_bytes) if grp_traffic else None,
                             "wave_time": (grp_pmc or {}).get("wave_time"), "pmc": grp_prov}}
        guarded("every_load_kernel", blk_every_load)

        # ---- the other BASELINE configurations on this GPU
        n_w = max(5, min(args.steps, 50))
        blocks = {}
        secondary["workloads"] = blocks

        def blk_workloads():
            if wl.name == "C3":
                blocks["C3h"] = workload_block(synth.WORKLOADS["C3h"], scene, n_w, 5)
                blocks["C5"] = workload_block(synth.WORKLOADS["C5"], scene, n_w, 5)
                w2 = synth.WORKLOADS["C2"]
                scene2 = hmrm.Scene(*maps_of(w2), w2.scene_params())
                blocks["C2"] = workload_block(w2, scene2, n_w, 5)
                scene2.close()
                # the reference's own operating points: its default grid_width 0.05 (main/hmap.cpp:65) under the headline's
                # camera and cells, and its sample config's grid_width 0.01 with step_dist 0.05 = 5 cells per step
                # (sample_config.txt:5-7) on the C5 frame.  Same maps; the scene parameters differ (grid width, heights).
                for wr in (synth.grid_workload("C3", 0.05), synth.WORKLOADS["REFDEF"]):
                    scene_r = hmrm.Scene(rgb, cmap, wr.scene_params())
                    blocks[wr.name] = workload_block(wr, scene_r, n_w, 5)
                    scene_r.close()
        guarded("workloads", blk_workloads)

        # ---- maps built to defeat the traversal: the library as shipped against the plain speculative groups
        def blk_rough():
            rough = {}
            n_r = max(3, min(args.steps, 10))
            for kind in ("white", "spikes", "needles", "canyon"):
                wr = synth.content_workload(wl.name, kind)
                sc = hmrm.Scene(*wr.maps(), wr.scene_params())
                cr = wr.camera()
                fbr, sr, _, _ = sc.render_stats(cr)
                precondition(sc, cr)  # (includes the scene's one-time kernel probe: pyramid leaps or groups + window records, whichever measured faster)
                chosen = ("leaps", "groups", "literal", "groups + window records")[sc.kernel_choice()]
                leap_ms = sc.bench_kernel_ms(cr, n_r)

                def group():
                    sc.bench_kernel_ms(cr, 2)
                    return sc.bench_kernel_ms(cr, n_r)
                group_ms = with_kernel("group", group)
                if not np.array_equal(sc.render(cr), fbr):
                    raise SystemExit(f"bench.py: rough_terrain/{kind}: hmrm_render and hmrm_render_stats disagree")
                rough[kind] = {"workload": workload_text(wr), "kernel_ms": leap_ms, "kernel_chosen_by_probe": chosen,
                               "group_kernel_ms": group_ms, "over_group": leap_ms / group_ms, "ray_steps_per_frame": int(sr.steps),
                               "value": int(sr.steps) / (leap_ms * 1e-3), "mrays_per_s": int(sr.rays) / (leap_ms * 1e-3) / 1e6,
                               "executed_per_frame": executed(sr)}
                if kind == "needles" and wl.name == "C3":
                    # the record kernel's own rocprofv3 / PMC round (tools/profile_r05.sh, last leg): what bounds it
                    rp, rprov = _pmc_from_profiles("C3_needles_rec", hmrm.kernel_src_sha())

                    def rec():
                        sc.bench_kernel_ms(cr, 2)
                        return sc.bench_kernel_ms(cr, n_r)
                    rec_ms = with_kernel("rec", rec)
                    rough[kind]["record_kernel"] = {
                        "kernel": "k_render_fast<.., LEAP = 2> (HMRM_KERNEL=rec): groups of 6 positions + leaps over 16-cell windows whose 8 highest "
                                  "cells the path misses (frame.hpp WindowRecord)",
                        "kernel_ms": rec_ms, "over_group": rec_ms / group_ms,
                        "rocprof_mean_us": (rp or {}).get("kernel_us_per_frame_rocprof"), "lane_util": (rp or {}).get("lane_util"),
                        "hbm_bytes_per_launch": (rp or {}).get("hbm_bytes_per_launch"), "wave_time": (rp or {}).get("wave_time"), "pmc": rprov}
                sc.close()
            rough["note"] = ("same camera as the headline (canyon: low and level, down the corridor); kernel_ms = the library as shipped "
                             f"(HIP events, {n_r} launches): the production kernel unless the scene's one-time probe measured the other kernel -- "
                             "the speculative groups with leaps over window records (a 16-cell window's maximum without its 8 highest cells, and "
                             "where those stand) -- at least 3 % faster; group_kernel_ms = HMRM_KERNEL=group (speculative groups of 6 positions, "
                             "no leaps of any kind: every load of main/hmap.cpp:1013)")
            secondary["rough_terrain"] = rough
        if not args.no_rough and wl.content == "smooth":
            guarded("rough_terrain", blk_rough)

        # ---- C4 on one GPU (8192^2 maps: the scene of the headline is released first)
        def blk_c4():
            w4 = synth.WORKLOADS["C4"]
            scene4 = hmrm.Scene(*w4.maps(), w4.scene_params())
            blocks["C4"] = workload_block(w4, scene4, max(5, min(args.steps, 20)), 3)
            scene4.close()
        if not args.no_c4 and wl.name == "C3":
            guarded("workloads.C4", blk_c4)
        torch.cuda.synchronize()
    elif not args.no_secondary and multi and args.mode == "frames":
        # ---- the N = 1 line's own step on every GPU (C3's static pose, one stream): comparable with that line's `value`
        w3 = synth.WORKLOADS["C3"]
        if w3.map_size == wl.map_size and wl.content == "smooth":
            c3 = w3.camera()
            fb3, st3, _, _ = scene.render_stats(c3)
            precondition(scene, c3)
            dt_all, check_all = static_leg(scene, c3, args.steps, args.warmup, 1, fb3)
            check_all()
            dt_one, check_one = static_leg(scene, c3, args.steps, args.warmup, 1, fb3, active=rank == 0)
            check_one()
            secondary["static_pose_replicas"] = {
                "workload": workload_text(w3), "frames_per_gpu": args.steps, "ms_per_step": dt_all * 1e3 / args.steps,
                "value": int(st3.steps) * args.steps * world / dt_all, "unit": "ray-steps/s", "scaling": "weak",
                "one_gpu_same_run": {"ms_per_step": dt_one * 1e3 / args.steps, "value": int(st3.steps) * args.steps / dt_one},
                "efficiency_vs_one_gpu": dt_one / dt_all,
                "note": "every GPU renders the N = 1 line's step (same pose, K frames, one stream); no collective"}
        # ---- BASELINE configs[3]: one 7680x4320 orthographic frame over the 8192^2 map in cyclic 16-row bands
        if not args.no_c4:
            scene.close()
            scene = None
            maps_cache.clear()
            wl4 = synth.WORKLOADS["C4"]
            scene4 = hmrm.Scene(*wl4.maps(), wl4.scene_params())
            secondary["c4_strips"] = strips_block(wl4, scene4, max(5, min(args.steps, 40)))
            scene4.close()

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_steps_timed / elapsed
        kernel_s = kernel_ms * 1e-3
        pmc, prov = _pmc_from_profiles(wl.name, hmrm.kernel_src_sha())
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        valu = (pmc or {}).get("valu") or {}
        cnt = (pmc or {}).get("counters_per_frame") or {}
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
            # the same fraction with numerator AND denominator from the profile round's box (rocprofv3's mean launch duration
            # of that run): boxes of the pool differ by +-3 %, `frac` mixes this run's kernel_ms with that box's counters
            "frac_on_the_profiled_box": (busy / ((pmc or {}).get("kernel_us_per_frame_rocprof") * 1e-6) / 1e9 / peak)
                                        if busy and (pmc or {}).get("kernel_us_per_frame_rocprof") else None,
            "traffic": traffic,
            "hbm": {"achieved": (traffic / kernel_s / 1e9) if traffic else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (traffic / kernel_s / 1e9 / HBM_PEAK_GBS) if traffic else None},
            "lane_util": (pmc or {}).get("lane_util"),
            # what the busy pipe is busy with (utilisation is not useful work): shares of the VALU wave-instructions
            # by the classes the counters separate, and the fraction with idle lanes taken out
            "valu_mix": ({k: (cnt.get(c, 0.0) / valu["insts"]) for k, c in (("fp64_add_mul_fma", None), ("int32", "SQ_INSTS_VALU_INT32"),
                                                                               ("int64", "SQ_INSTS_VALU_INT64"), ("conversions", "SQ_INSTS_VALU_CVT"),
                                                                               ("fp64_rcp_sqrt", "SQ_INSTS_VALU_TRANS_F64")) if c}
                         | {"fp64_add_mul_fma": valu.get("f64_add_mul_fma", 0.0) / valu["insts"]}) if valu.get("insts") else None,
            "frac_lane_weighted": (frac * pmc["lane_util"]) if (frac and (pmc or {}).get("lane_util")) else None,
            "wave_time": (pmc or {}).get("wave_time"),
            "kernel": "k_render_fast", "kernel_ms": kernel_ms,
            "kernel_ray_steps_per_s": frame_steps / kernel_s,
            "operating_point": "launches of the workload's static pose back to back on one stream (the same as `value` at N = 1)",
            # BASELINE.md's nominal figure: bytes the REFERENCE's loop would move for this frame over the
            # measured duration.  Not executed traffic (the loads are skipped), hence not a fraction of a peak.
            # (The kernel that does execute them: `literal_kernel.roofline`.)
            "algorithmic_equivalent": {"bytes_per_launch": algo_bytes, "gbs": algo_bytes / kernel_s / 1e9,
                                       "times_hbm_peak": algo_bytes / kernel_s / 1e9 / HBM_PEAK_GBS},
            "pmc": prov,
        }
        if pmc and pmc.get("in_flight"):
            # VALU-busy over the overlapped window of three frames in flight (its own PMC pass, tools/profile_round.sh)
            secondary.setdefault("frames_in_flight", {})["pmc"] = pmc["in_flight"]
        for name, blk in (secondary.get("workloads") or {}).items():
            p2, prov2 = _pmc_from_profiles(name, hmrm.kernel_src_sha())
            if p2 and (p2.get("valu") or {}).get("busy_cycles_isa"):
                blk["valu_frac"] = p2["valu"]["busy_cycles_isa"] / (blk["kernel_ms"] * 1e-3) / 1e9 / peak
                blk["hbm_bytes_per_launch"] = p2.get("hbm_bytes_per_launch")
                blk["pmc_source"] = prov2.get("source")
        line = {
            "metric": "ray-steps/s at 3840x2160, 4096^2 heightmap" if wl.map_size == 4096 else
                      f"ray-steps/s at {W}x{H}, {wl.map_size}^2 heightmap",
            "value": value, "unit": "ray-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "equivalent_steps": True,
            "frames_in_flight": in_flight,
            "preconditioning": f"{PRECONDITION_LAUNCHES} synchronised launches of the pose (launch-order calibration), then 50 queued ones, untimed",
            "executed_per_frame": executed(st),
            "mrays_per_s": rays_timed / elapsed / 1e6,
            "config": {"workload": workload_text(wl),
                       "ray_steps_per_frame": frame_steps, "rays_per_frame": frame_rays,
                       "hits_per_frame": frame_hits, "parallelism": parallelism,
                       "maps_sha256": synth.maps_sha256(rgb, cmap)[:16]},
            "roofline": roofline,
        }
        line.update(extra)
        if job.share_gpu:
            line["share_gpu"] = True  # (a rehearsal: ranks share GPUs, collectives over gloo on host copies; not a scaling figure)
        if job.rccl_ranks is not None:
            line["rccl_ranks"] = job.rccl_ranks
        line.update(secondary)
        if world == 1 and secondary.get("workloads"):
            # the span of operating points at a glance (every figure is also in its own block above): ms per frame on one
            # stream (kernel_ms where the block has one), 3840x2160 over the 4096^2 map throughout
            wk = secondary["workloads"]
            rough = {k: v for k, v in (secondary.get("rough_terrain") or {}).items() if isinstance(v, dict)}
            worst = max(rough.items(), key=lambda kv: kv[1]["kernel_ms"]) if rough else None
            fc = secondary.get("fresh_camera") or {}
            line["operating_range"] = {
                "best": {"what": "static pose, calibrated launch order, grid_width 1 (the headline)", "ms_per_step": ms_per_step, "kernel_ms": kernel_ms},
                "moving": {"what": "a camera the library has never seen, every frame (host set-up in the loop)", "ms_per_step": fc.get("ms_per_step")},
                "reference_defaults": {name: {"what": workload_text(synth.grid_workload("C3", 0.05) if name != "REFDEF" else synth.WORKLOADS["REFDEF"]),
                                              "ms_per_step": wk[name]["ms_per_step"], "kernel_ms": wk[name]["kernel_ms"]}
                                       for name in ("C3/gw0.05", "REFDEF") if name in wk},
                "worst_content": ({"what": "the headline camera over `" + worst[0] + "` (rough_terrain)", "kernel_ms": worst[1]["kernel_ms"],
                                   "kernel": worst[1]["kernel_chosen_by_probe"]} if worst else None),
            }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(hmrm, wl, rgb, cmap, params, cam, args.cpu_seconds)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    job.finish()
    if scene is not None:
        scene.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
