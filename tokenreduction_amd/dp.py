"""Data-parallel gradient reduction.

The reference wraps the model in `torch.nn.parallel.DistributedDataParallel` (train.py:406): fp32 gradients are averaged over
the ranks in ~25 MB buckets while the backward pass is still running.  This module is the MI355X-side equivalent for a
one-process-per-GPU job on RCCL (`torch.distributed`, backend "nccl" on ROCm; "gloo" in the CPU tests):

* buckets are filled in REVERSE parameter order (the order gradients become ready in a backward pass) and launched as soon as
  their last gradient has been accumulated (`register_post_accumulate_grad_hook`), on a side stream when the gradients live on
  the GPU, so the collective overlaps the rest of the backward;
* xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by ONE link, so buckets are larger than
  NCCL-on-NVSwitch habits (default 64 MiB: ~0.45 ms of wire time per bucket at 8 GPUs, against ~20 us of launch latency) and the
  payload can be sent as bf16 (`comm_dtype`) to halve the bytes -- the mean is still accumulated into the fp32 `.grad`;
* `finish()` waits for the outstanding collectives, divides by the world size and scatters the flat buckets back.

It is plumbing, not compute: the backward kernels that would feed it are not built yet, so it is exercised
by the world_size-2 gloo test in tests/test_dp.py only.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradientAllReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, process_group=None,
                 comm_dtype: Optional[torch.dtype] = None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.group = process_group
        self.comm_dtype = comm_dtype
        self.world = dist.get_world_size(process_group)
        # buckets in reverse parameter order, closed when they reach bucket_bytes (a parameter is never split)
        self.buckets: List[List[torch.nn.Parameter]] = [[]]
        size = 0
        for p in reversed(self.params):
            nbytes = p.numel() * (torch.finfo(comm_dtype).bits // 8 if comm_dtype else p.element_size())
            if self.buckets[-1] and size + nbytes > bucket_bytes:
                self.buckets.append([])
                size = 0
            self.buckets[-1].append(p)
            size += nbytes
        self._bucket_of = {id(p): b for b, ps in enumerate(self.buckets) for p in ps}
        self._pending = [0] * len(self.buckets)
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._hooks = []
        self._stream = None
        self._armed = False

    # ---- lifecycle ------------------------------------------------------------------------------------------------------
    def attach(self):
        """Install the per-parameter hooks (once); call `start()` before every backward pass."""
        if self._hooks:
            return self
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        return self

    def detach(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def start(self):
        self._pending = [len(ps) for ps in self.buckets]
        self._work = [None] * len(self.buckets)
        self._armed = True

    def _on_grad(self, p: torch.nn.Parameter):
        if not self._armed:
            return
        b = self._bucket_of[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b: int):
        ps = self.buckets[b]
        dev = ps[0].grad.device
        dtype = self.comm_dtype or ps[0].grad.dtype
        n = sum(p.numel() for p in ps)
        flat = self._flat[b]
        if flat is None or flat.numel() != n or flat.dtype != dtype or flat.device != dev:
            flat = self._flat[b] = torch.empty(n, dtype=dtype, device=dev)
        if dev.type == "cuda":
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=dev)
            self._stream.wait_stream(torch.cuda.current_stream(dev))       # the gradients of this bucket are complete
            ctx = torch.cuda.stream(self._stream)
        else:
            ctx = _NullCtx()
        with ctx:
            off = 0
            for p in ps:
                flat[off: off + p.numel()].copy_(p.grad.reshape(-1))
                off += p.numel()
            self._work[b] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Wait for every bucket, write the mean back into `.grad`.  Buckets whose gradients never all arrived (unused
        parameters) are reduced here with zeros in the gaps, so every rank issues the same collectives."""
        if not self._armed:
            raise RuntimeError("finish() without start()")
        for b, ps in enumerate(self.buckets):
            if self._work[b] is None:
                for p in ps:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                self._launch(b)
        for b, ps in enumerate(self.buckets):
            self._work[b].wait()
            flat = self._flat[b]
            dev = flat.device
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).wait_stream(self._stream)
            off = 0
            for p in ps:
                p.grad.copy_(flat[off: off + p.numel()].reshape(p.shape).to(p.grad.dtype) / self.world)
                off += p.numel()
        self._armed = False


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
