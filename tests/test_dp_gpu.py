"""Row a24 / e2 on the GPU: two data-parallel ranks SHARING the box's one MI355X (RCCL refuses two ranks on one device, so the
collectives go through gloo, which reduces HIP tensors too).  Everything else is the production path: each rank runs the HIP
training forward / backward on its own shard of images, the backward is walked in block ranges, every finished bucket is
reduce-scattered + all-gathered on the side stream while the next range runs, and the result must be the mean of the two ranks'
single-process gradients."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from tokenreduction_amd.dp import FlatGradReducer
    from tests._params import GOLDEN_CASES, grad_labels, make_images
    from tests.test_hip_model import build_model
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    name = os.environ["TR_DP_CASE"]
    case = GOLDEN_CASES[name]
    model, _, _ = build_model(case)
    model.viz_mode = False
    model.train()

    def shard(r):
        x = make_images(case["batch"], 224, case["xseed"] + 100 * r).cuda()
        y = grad_labels(dict(case, xseed=case["xseed"] + 100 * r)).cuda()
        return x, y

    def backward(x, y):
        model.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(model(x), y).backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in model.named_parameters()}

    local = [backward(*shard(r)) for r in range(world)]           # every rank computes both shards' gradients on its own (no reducer)
    want = {n: sum(g[n] for g in local) / world for n in local[0]}
    comm = {"": None, "bf16": torch.bfloat16}[os.environ.get("TR_DP_COMM", "")]
    red = FlatGradReducer(bucket_bytes=256 * 1024, algorithm=os.environ["TR_DP_ALGO"], comm_dtype=comm).attach(model)
    red.broadcast_parameters(model)
    for rep in range(2):
        got = backward(*shard(rank))                              # this rank's shard, reduced in buckets during the backward
        assert len(red.launched) >= 3, red.launched
        for n in want:
            if comm is None:
                assert torch.allclose(got[n], want[n], rtol=1e-5, atol=1e-7), (rep, n, float((got[n] - want[n]).abs().max()))
            else:     # bf16 payload on the links: within 2^-7 of the larger contribution per element (stated in dp.FlatGradReducer)
                scale = torch.stack([g[n].abs() for g in local]).max(0).values
                assert ((got[n] - want[n]).abs() <= scale * 2.0 ** -7 + 1e-12).all(), (rep, n)
    if rank == 0:
        print("dp gpu ok", name, len(red.launched), "buckets")
    dist.destroy_process_group()
""") % ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("case,algo,comm", [("topk_micro", "rs_ag", ""), ("dpcknn_micro", "rs_ag", ""), ("evit_micro", "all_reduce", ""),
                                            ("topk_micro", "rs_ag", "bf16")])
def test_two_ranks_average_their_hip_gradients(tmp_path, case, algo, comm):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    script = tmp_path / "dp_gpu_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TR_DP_CASE=case, TR_DP_ALGO=algo, TR_DP_COMM=comm)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29641", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "dp gpu ok" in out.stdout


@pytest.mark.gpu
def test_bench_runs_its_two_rank_path_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` (self-launching; the watchdog test below uses the driver's torch.distributed.run form) with both ranks on the box's one GPU
    (TR_BENCH_SHARE_GPU=1: gloo collectives): the headline line and all three data-parallel fine-tune legs must come out."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    # started WITHOUT a launcher -- `python bench.py --gpus 2`, the N = 1 command with another number: the parent (which never touches the
    # GPU) starts the ranks itself and relays rank 0's line
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TR_BENCH_SHARE_GPU="1", TR_BENCH_FINETUNE_STEPS="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["scaling"] == "weak" and rec["value"] > 0
    # what the process group reports, not the command line: two ranks, and (shared-GPU rig) both on device 0
    assert rec["dp"]["world"] == 2 and rec["dp"]["cuda_device_by_rank"] == {"0": 0, "1": 0}, rec["dp"]
    assert len(rec["finetune"]) == 3 and all("error" not in v and v["n_gpus"] == 2 for v in rec["finetune"].values()), rec["finetune"]
    # the first real 8-GPU run must explain itself: every fine-tune leg carries the per-bucket timing of its gradient mean
    for label, v in rec["finetune"].items():
        dp = v["dp"]
        assert dp is not None and dp["buckets"] >= 1 and dp["world"] == 2, (label, dp)
        assert len(dp["bucket_bytes"]) == len(dp["collective_ms"]) == len(dp["range_ms"]) == dp["buckets"], (label, dp)
        assert all(t >= 0 for t in dp["collective_ms"] + dp["range_ms"]) and dp["exposed_ms"] is not None and dp["exposed_ms"] >= 0, (label, dp)
        assert dp["algorithm"] in ("rs_ag", "all_reduce") and dp["comm_dtype"] == "float32" and "rccl_version" in dp, (label, dp)


@pytest.mark.gpu
def test_bench_watchdog_ends_a_hung_run_non_zero():
    """A collective that never returns must not look like a successful run: the N > 1 watchdog flushes the headline line (the forward WAS
    measured) and ends the process with a non-zero code.  Provoked here with a one-second limit on the fine-tune legs."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TR_BENCH_SHARE_GPU="1", TR_BENCH_FINETUNE_STEPS="2", TR_BENCH_FINETUNE_TIMEOUT="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29647", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode != 0, out.stdout[-1500:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:] + out.stderr[-1500:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and "timed out" in json.dumps(rec)
