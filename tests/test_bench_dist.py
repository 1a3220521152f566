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
