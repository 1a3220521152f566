import os, sys
sys.path.insert(0, ".")
import torch, torch.distributed as dist
from tokenreduction_amd.dp import GradientAllReducer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Linear(384, 1536), torch.nn.GELU(), torch.nn.Linear(1536, 384), torch.nn.GELU(), torch.nn.Linear(384, 1000)).cuda()
red = GradientAllReducer(model.parameters(), bucket_bytes=1 << 20, comm_dtype=torch.bfloat16).attach()
x, y = torch.randn(256, 384, device="cuda"), torch.randint(0, 1000, (256,), device="cuda")
for step in range(3):
    for p in model.parameters():
        p.grad = None
    red.start()
    torch.nn.functional.cross_entropy(model(x), y).backward()
    ref = [p.grad.clone() for p in model.parameters()]
    red.finish()
    torch.cuda.synchronize()
    err = max(((p.grad - r).abs().max() / (r.abs().max() + 1e-12)).item() for p, r in zip(model.parameters(), ref))
    print("step", step, "buckets", len(red.buckets), "max rel diff after bf16 all-reduce (world 1):", f"{err:.2e}")
    assert err < 1e-2
dist.destroy_process_group()
print("dp gpu ok")
