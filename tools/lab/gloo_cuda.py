import os, torch, torch.distributed as dist
dist.init_process_group("gloo")
r = dist.get_rank()
torch.cuda.set_device(0)
x = torch.full((8,), float(r + 1), device="cuda")
for name, fn in (("all_reduce", lambda: dist.all_reduce(x)), ("broadcast", lambda: dist.broadcast(x, 0)),
                 ("reduce_scatter_tensor", lambda: dist.reduce_scatter_tensor(torch.empty(4, device="cuda"), x)),
                 ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(torch.empty(16, device="cuda"), x))):
    try:
        fn(); torch.cuda.synchronize(); print(r, name, "ok", x[:2].tolist())
    except Exception as e:
        print(r, name, "FAIL", type(e).__name__, str(e)[:80])
dist.destroy_process_group()
