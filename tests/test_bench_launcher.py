"""bench.py's multi-process contract without a GPU (VERDICT r3 "next round" 2): `python bench.py --gpus N` started the way the driver
starts `--gpus 1` -- plain python, no WORLD_SIZE -- must launch one process per GPU itself (torch.distributed.run, 127.0.0.1, a free
port), time exactly K steps between barriers, report the MAX over ranks, and print ONE JSON line from rank 0; on a box with fewer
devices than asked for it must print one JSON error line and exit 2. VT_BENCH_STUB=1 swaps the GPU step for a host stub (gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_plain_python_gpus2_launches_itself_and_reports_the_slowest_rank():
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1"], {"VT_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["self_launched"] is True
    assert d["ms_per_step"] >= 4.0                                    # rank 1's stub step sleeps 4 ms, rank 0's 2 ms: max over ranks
    assert abs(d["value"] - 2 * 5120 * 5 / (d["ms_per_step"] * 5e-3)) / d["value"] < 1e-6      # whole-job tokens / max-over-ranks time


def test_world_size_from_the_environment_is_honoured_under_torchrun():
    """The driver's own form for N > 1: python -m torch.distributed.run ... bench.py --gpus N (no self-launch, no double spawn)."""
    env = dict(os.environ, VT_BENCH_STUB="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2 and json.loads(lines[0])["config"]["self_launched"] is False


def test_more_gpus_than_the_box_has_is_one_json_error_line_and_exit_2():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(["--gpus", str(have + 3), "--steps", "1", "--warmup", "0"], {})
    assert r.returncode == 2, (r.returncode, r.stderr[-1000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == have + 3 and "GPU(s) visible" in d["error"]
