"""The N>1 path of bench.py on CPU: world_size 2 over gloo through torch.distributed.run, exactly as the driver launches it
(no model, no GPU: `--selftest-gloo` swaps the step for a rank-dependent sleep).  Checks the barrier + MAX-over-ranks timing and
the whole-job aggregation of `value`."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(nproc, steps=5, warmup=1):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", str(steps),
           "--warmup", str(warmup), "--selftest-gloo"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # rank 0 prints ONE json line
    return json.loads(lines[0])


def test_two_ranks_gloo_max_over_ranks_and_aggregate():
    r = _run(2)
    assert r["n_gpus"] == 2 and r["steps"] == 5 and r["scaling"] == "weak"
    # rank 1 sleeps 20 ms per step, rank 0 only 10 ms: the reported time is the MAX over ranks
    assert 19.0 <= r["ms_per_step"] <= 60.0, r
    # whole-job aggregate: both ranks' 256-image batches per step
    assert abs(r["value"] - 2 * 256 / (r["ms_per_step"] * 1e-3)) / r["value"] < 0.02


def test_single_process_selftest():
    r = _run(1, steps=3)
    assert r["n_gpus"] == 1 and 9.0 <= r["ms_per_step"] <= 40.0


def test_bench_launches_its_own_ranks_when_started_without_a_launcher():
    """`python bench.py --gpus 2` -- the N = 1 command with another number, no torchrun on the command line: the parent process starts
    the two ranks itself (children through torch.distributed.run), relays rank 0's line and the exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--selftest-gloo"],
                         capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and 19.0 <= r["ms_per_step"] <= 60.0, r


def test_self_launch_relays_a_failing_rank_as_a_non_zero_exit():
    """A rank that dies (test hook: rank 1 exits 7 before the rendezvous) ends the launcher non-zero, and the parent relays that."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", TR_BENCH_SELFTEST_FAIL_RANK="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-gloo", "--steps", "2"],
                         capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], out.stdout
