import sys, os
sys.path.insert(0, "/root/repo")
import torch, bench
from torch.profiler import profile, ProfilerActivity
model = bench.build_model(bench.MODEL, [0.7], [3,6,9], "cuda").train()
x = torch.randn(64, 3, 224, 224, device="cuda"); y = torch.randint(0, 1000, (64,), device="cuda")
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05, fused=True)
def step():
    out = model(x); loss = torch.nn.functional.cross_entropy(out, y)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(3): step()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
ev = [e for e in prof.events() if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::copy_", "aten::_to_copy")]
from collections import Counter
c = Counter()
for e in ev:
    st = [s for s in (e.stack or []) if "tokenreduction_amd" in s or "bench" in s or "optim" in s]
    c[(e.name, st[0] if st else (e.stack[0] if e.stack else "?"))] += 1
for k, v in c.most_common(20): print(v, k)
