"""Row a24: bucketed gradient mean over a data-parallel group, exercised on CPU with gloo, world_size 2."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from tokenreduction_amd.dp import GradientAllReducer
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(0)                                   # identical replicas
    model = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(),
                                torch.nn.Linear(64, 10))
    unused = torch.nn.Parameter(torch.ones(7))             # a parameter no loss touches
    params = list(model.parameters()) + [unused]
    red = GradientAllReducer(params, bucket_bytes=8 * 1024).attach()
    assert len(red.buckets) >= 3, len(red.buckets)         # several buckets, reverse parameter order
    assert red.buckets[0][0] is unused and red.buckets[-1][-1] is params[0]
    for step in range(2):
        g = torch.Generator().manual_seed(100 + 10 * step + rank)          # every rank its own shard
        x, y = torch.randn(16, 32, generator=g), torch.randint(0, 10, (16,), generator=g)
        for p in params:
            p.grad = None
        red.start()
        torch.nn.functional.cross_entropy(model(x), y).backward()
        red.finish()
        # expected: mean over ranks of the per-rank gradients, recomputed locally from both shards
        want = [torch.zeros_like(p) for p in params]
        for r in range(world):
            g = torch.Generator().manual_seed(100 + 10 * step + r)
            xr, yr = torch.randn(16, 32, generator=g), torch.randint(0, 10, (16,), generator=g)
            grads = torch.autograd.grad(torch.nn.functional.cross_entropy(model(xr), yr), list(model.parameters()))
            for w, gr in zip(want, grads):
                w += gr / world
        for p, w in zip(params, want):
            assert torch.allclose(p.grad, w, atol=1e-6), (step, (p.grad - w).abs().max())
    if rank == 0:
        print("dp ok", len(red.buckets), "buckets")
    dist.destroy_process_group()
""") % ROOT


def test_bucketed_gradient_mean_gloo_world2(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29617", str(script)], capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "dp ok" in out.stdout
