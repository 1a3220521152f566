import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from tokenreduction_amd import ops, _lib
B, H, N = 256, 6, 197
qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5).bfloat16()
out = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device="cuda")
dbg = torch.zeros(B * H * 4 * 8 * 2 + 64, dtype=torch.float32, device="cuda")
lib = _lib.load()
for _ in range(3):
    lib.tr_attention_bf16(qkv.data_ptr(), out.data_ptr(), dbg.data_ptr(), None, None, B, N, H, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
d = dbg.cpu().numpy().view(np.int64)[: B * H * 4 * 8].reshape(B * H, 4, 8)
names = ["start", "staged", "q loaded", "QK done", "softmax done", "PV done", "stored", "end"]
for w in range(4):
    med = np.median(d[:, w, :], axis=0)
    print("wave", w, " ".join(f"{n}:{int(v)}" for n, v in zip(names, med)))
print("(cycles of clock64 since kernel start of that wave; first query block only, 'end' = after all blocks)")
