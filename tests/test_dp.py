"""Row a24: bucketed in-place gradient mean over a data-parallel group, exercised on CPU with gloo, world_size 2.

The HIP backward cannot run here, so each rank fills the model's flat gradient buffer (training.TrainState: the real layout,
the real bucket plan) with the gradients torch.autograd gives over the oracle on the rank's own shard of images -- the same
values `loss.backward()` produces on the GPU up to bf16 rounding (tests/test_hip_train.py) -- and the reducer must turn them
into the mean over the ranks, bucket by bucket, in place.
"""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, types
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import tokenreduction_amd as tra
    from tokenreduction_amd import training
    from tokenreduction_amd.dp import FlatGradReducer
    from tests._params import GOLDEN_CASES, case_params, oracle_param_grads
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    case = dict(GOLDEN_CASES["topk_micro"])
    args = types.SimpleNamespace(keep_rate=case["keep_rate"], reduction_loc=case["reduction_loc"])
    model = tra.TopKVisionTransformer(img_size=224, patch_size=16, embed_dim=128, depth=4, num_heads=2, mlp_ratio=4, qkv_bias=True,
                                      num_classes=16, args=args)
    cfg, params = case_params(case)
    model.load_state_dict(params)
    st = training.TrainState(model)                        # flat buffer in backward order, p.grad views
    comm = {"": None, "bf16": torch.bfloat16}[os.environ.get("TR_DP_COMM", "")]
    red = FlatGradReducer(bucket_bytes=256 * 1024, algorithm=os.environ["TR_DP_ALGO"], comm_dtype=comm).attach(model)
    assert red.algorithm == os.environ["TR_DP_ALGO"]
    # the default bucket size: a sixth of the gradient bytes clamped to [8, 64] MiB -- this micro model is one bucket, DeiT-S is six
    assert len(FlatGradReducer().plan(st.block_slices, model.depth)) == 1
    red.broadcast_parameters(model)
    plan = red.plan(st.block_slices, model.depth)
    assert len(plan) >= 3, plan                            # several buckets
    assert plan[0][0] == model.depth - 1 and plan[-1][1] == 0 and plan[0][2] == 0 and plan[-1][3] == st.flat.numel()
    assert all(a[3] == b[2] and a[1] == b[0] + 1 for a, b in zip(plan, plan[1:])), plan      # contiguous slices, consecutive ranges
    assert all((b[3] - b[2]) %% 64 == 0 for b in plan)

    def local_grads(r):
        c = dict(case, xseed=case["xseed"] + 100 * r)      # every rank its own shard of images / labels
        return oracle_param_grads(c)[2]

    mine = local_grads(rank)
    for n, p in st.order:
        st.views[n].copy_(mine[n])
        p.grad = st.views[n]
    for hi, lo, start, stop in plan:                       # what training._VitTrainFn.backward does after each block range
        red.reduce_slice(st.flat, start, stop)
    red.finish(st.flat)
    want = {n: sum(local_grads(r)[n] for r in range(world)) / world for n, _ in st.order}
    for n, p in model.named_parameters():
        if comm is None:
            assert torch.allclose(p.grad, want[n], rtol=1e-6, atol=1e-8), (n, (p.grad - want[n]).abs().max())
        else:
            # bf16 payload: every rank's slice is rounded to bf16 (2^-9 relative), summed in bf16 by the collective and divided:
            # per element within 2^-7 of the largest contribution
            scale = torch.stack([local_grads(r)[n].abs() for r in range(world)]).max(0).values
            assert ((p.grad - want[n]).abs() <= scale * 2.0 ** -7 + 1e-12).all(), (n, ((p.grad - want[n]).abs() / (scale + 1e-30)).max())
            assert not torch.equal(p.grad, want[n]) or want[n].abs().max() == 0
    # no_sync: nothing is reduced (gradient accumulation micro-steps)
    with red.no_sync():
        assert red.sync is False
    assert red.sync is True
    if rank == 0:
        print("dp ok", len(plan), "buckets")
    dist.destroy_process_group()
""") % ROOT


import pytest


@pytest.mark.parametrize("world,algo,comm", [(2, "all_reduce", ""), (2, "rs_ag", ""), (4, "rs_ag", ""), (2, "rs_ag", "bf16")])
def test_bucketed_gradient_mean_gloo(tmp_path, world, algo, comm):
    """rs_ag = the RCCL path (reduce-scatter into a shard + all-gather in place; this torch's gloo implements both collectives, so
    the shard arithmetic is exercised at world sizes 2 and 4 on CPU); all_reduce = the fallback for buckets the world size does not divide."""
    script = tmp_path / "dp_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", TR_DP_ALGO=algo, TR_DP_COMM=comm)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                          "--master-port", str(29617 + world + (7 if algo == "rs_ag" else 0) + (13 if comm else 0)), str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "dp ok" in out.stdout
