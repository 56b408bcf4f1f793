"""bench.py's own launcher: `python bench.py --gpus N` with no torch.distributed environment must start
its N ranks itself -- as a child process, before anything touches a GPU -- relay rank 0's one JSON line and its exit
code.  Rehearsed here on CPU with --dry-launch (gloo; a pattern instead of the renderer)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HMRM_FORCE_DIST")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _run(*flags, timeout=300):
    return subprocess.run([sys.executable, BENCH, *flags], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          env=_clean_env(), timeout=timeout)


def test_gpus_2_without_torchrun_launches_two_ranks_and_relays_one_json_line():
    p = _run("--gpus", "2", "--dry-launch", "--steps", "4", "--warmup", "1")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["dry_launch"] is True and line["value"] is None
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1
    assert line["rccl_ranks"] == {"world_size": 2, "backend": "gloo", "distinct_devices": 2}
    assert line["frames_rendered_by_all_ranks"] == 8  # K frames per rank
    assert line["scaling"] == "weak" and line["ms_per_step"] > 0


def test_four_ranks_dry_launch_incl_the_pipelined_strips():
    """The rank count the driver uses between 2 and 8.  With 4 ranks and 2 band sets per frame the dry launch's 100-row frame has
    ONE band per virtual rank -- the case in which StripPipeline's reassembly is a view of the block it gathers into, and a frame
    handed out used to be overwritten by the frame two later (found by this rehearsal in round 5)."""
    p = _run("--gpus", "4", "--dry-launch", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["rccl_ranks"]["world_size"] == 4 and line["frames_rendered_by_all_ranks"] == 12


def test_under_torchrun_the_same_file_is_a_rank():
    """What the driver does for N > 1: python -m torch.distributed.run ... bench.py --gpus N."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--dry-launch",
                        "--steps", "2", "--warmup", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_clean_env(), timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0])["rccl_ranks"]["world_size"] == 2


def test_one_rank_dry_launch_needs_no_launcher():
    p = _run("--gpus", "1", "--dry-launch", "--steps", "2", "--warmup", "0")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip())
    assert line["n_gpus"] == 1 and line["rccl_ranks"] is None


def test_failing_ranks_give_a_non_zero_exit_code_and_no_json():
    """Without --dry-launch the ranks need GPUs: on this CPU-only box they exit with a message, and the launcher must
    pass the failure on instead of printing a line.  (Skipped where GPUs exist: the real thing would run.)"""
    import importlib
    hmrm = importlib.import_module("heightmap-ray-marcher_amd")
    try:
        have = hmrm.device_count() > 0
    except hmrm.HmrmError:
        have = False
    if have:
        import pytest
        pytest.skip("GPUs present: the launcher would start a real run")
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-secondary", "--no-cpu-baseline")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert b"needs a GPU" in p.stderr or b"failed" in p.stderr
