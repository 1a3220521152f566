"""Data-parallel gradient reduction.

The reference wraps the model in `torch.nn.parallel.DistributedDataParallel` (train.py:405-407): fp32 gradients are averaged over
the ranks in ~25 MB buckets while the backward pass is still running, on every micro-step (engine.py:41-84 uses no `no_sync`).
This is the MI355X-side equivalent for a one-process-per-GPU job on RCCL (`torch.distributed`, backend "nccl" on ROCm; "gloo" in
the CPU tests), built on what the HIP backward executor provides (training.py):

* every parameter gradient lives in ONE flat fp32 buffer, in the order the backward finishes them, so a bucket is a contiguous
  slice [start, stop) of that buffer: the collective runs IN PLACE on the slice -- no flatten / unflatten copies;
* the backward runs in block ranges (tr_vit_backward blk_hi..blk_lo); after each range an event is recorded and that range's
  bucket is reduced on a side stream while the next range's kernels run (the DDP overlap);
* xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by ONE link, so the mean is taken as
  reduce-scatter + all-gather (RCCL drives all links for these with its direct algorithms) and buckets are larger than
  NCCL-on-NVSwitch habits: by default a sixth of the model's gradient bytes, clamped to [8, 64] MiB and cut at block boundaries
  (DeiT-B's 346 MB of gradients -> 6 buckets of ~58 MB, DeiT-S's 88 MB -> 6 of ~15 MB: every model overlaps all but the last bucket's
  collective with the remaining backward ranges);
* `no_sync()` skips the reduction for gradient-accumulation micro-steps (the reference reduces on every micro-step; same result,
  1/accum of the traffic).
"""
from __future__ import annotations

import contextlib
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class FlatGradReducer:
    """Bucketed in-place mean of a flat gradient buffer over the ranks of `process_group`."""

    TARGET_BUCKETS, MIN_BUCKET, MAX_BUCKET = 6, 8 << 20, 64 << 20

    def __init__(self, process_group=None, bucket_bytes: Optional[int] = None, comm_dtype: Optional[torch.dtype] = None,
                 algorithm: str = "auto"):
        """bucket_bytes None: chosen per model in plan() -- total gradient bytes / TARGET_BUCKETS, clamped to [MIN_BUCKET, MAX_BUCKET].
        comm_dtype (e.g. torch.bfloat16): the payload on the links is a copy of the slice in that type (half the bytes for bf16); the
        mean is then rounded to it once per rank pair -- relative error <= 2^-8 per element for bf16 (tests/test_dp.py)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_bytes = None if bucket_bytes is None else int(bucket_bytes)
        self.comm_dtype = comm_dtype
        backend = dist.get_backend(process_group)
        if algorithm == "auto":
            algorithm = "rs_ag" if backend == "nccl" else "all_reduce"      # xGMI: reduce-scatter + all-gather drive every link; CPU/gloo: one all-reduce
        if algorithm not in ("rs_ag", "all_reduce"):
            raise ValueError("algorithm must be 'auto', 'rs_ag' or 'all_reduce'")
        self.algorithm = algorithm
        self.avg_op = dist.ReduceOp.AVG if backend == "nccl" else None       # gloo: SUM then divide
        self.sync = True
        self._stream = None
        self._stage = None
        self._shard = None
        self.launched: List[Tuple[int, int]] = []          # (start, stop) of the buckets of the last backward, in launch order
        # record_timing: HIP events around every bucket of the LAST backward -- on the compute stream where the bucket's block range ends, on
        # the side stream around its collective, and around finish() -- so that timing() can tell overlap from exposure (bench.py `dp`)
        self.record_timing = False
        self._t_begin = None
        self._t_events: List[Tuple] = []
        self._t_finish = None

    # ---- model wiring ---------------------------------------------------------------------------------------------------
    def attach(self, model):
        """Make `loss.backward()` of this model reduce its gradients (training._VitTrainFn.backward consults the reducer)."""
        model._grad_reducer = self
        return self

    def broadcast_parameters(self, model, src: int = 0):
        """DDP's construction-time broadcast: every rank starts from rank `src`'s parameters and buffers."""
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=src, group=self.group)
        if hasattr(model, "weights_changed"):
            model.weights_changed()              # writes through .data do not bump the version counters the executor's pack cache reads

    @contextlib.contextmanager
    def no_sync(self):
        old, self.sync = self.sync, False
        try:
            yield
        finally:
            self.sync = old

    # ---- bucketing ------------------------------------------------------------------------------------------------------
    def plan(self, block_slices, depth: int):
        """block_slices: [(event_block, start, stop)] contiguous slices of the flat buffer in backward order (training.TrainState).
        Returns [(blk_hi, blk_lo, start, stop)]: block ranges of tr_vit_backward and the bucket each completes."""
        out, cur_hi, cur_start, cur_stop = [], None, None, None
        bucket_bytes = self.bucket_bytes
        if bucket_bytes is None:
            total = 4 * max(stop for _, _, stop in block_slices)
            bucket_bytes = min(self.MAX_BUCKET, max(self.MIN_BUCKET, total // self.TARGET_BUCKETS))
        for e, start, stop in block_slices:
            blk = min(e, depth - 1) if e < depth else 0            # the final slice (embedding + family modules) ends with block 0
            if cur_hi is None:
                cur_hi, cur_start = depth - 1, start
            cur_stop = stop
            if e < depth and (cur_stop - cur_start) * 4 >= bucket_bytes and blk > 0:
                out.append((cur_hi, blk, cur_start, cur_stop))
                cur_hi, cur_start = blk - 1, stop
        if cur_hi is not None and cur_start < cur_stop:
            out.append((cur_hi, 0, cur_start, cur_stop))
        return out

    # ---- the collective -------------------------------------------------------------------------------------------------
    def begin(self, flat: torch.Tensor):
        """Start of a backward pass that will reduce (training._VitTrainFn.backward): forget the last pass's bucket list and timing events."""
        self.launched = []
        self._t_events, self._t_finish, self._t_begin = [], None, None
        if self.record_timing and flat.is_cuda:
            self._t_begin = torch.cuda.Event(enable_timing=True)
            self._t_begin.record(torch.cuda.current_stream(flat.device))

    def reduce_slice(self, flat: torch.Tensor, start: int, stop: int):
        """Mean over the ranks of flat[start:stop], in place, on the side stream (GPU) / synchronously (CPU)."""
        sl = flat[start:stop]
        self.launched.append((start, stop))
        if flat.is_cuda:
            dev = flat.device
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=dev)
            timing = self.record_timing
            ev = torch.cuda.Event(enable_timing=timing)
            ev.record(torch.cuda.current_stream(dev))               # the range's gradient kernels are enqueued before this point
            self._stream.wait_event(ev)
            with torch.cuda.stream(self._stream):
                if timing:
                    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    c0.record(self._stream)
                self._collective(sl)
                if timing:
                    c1.record(self._stream)
                    self._t_events.append((ev, c0, c1, (stop - start) * 4))
        else:
            self._collective(sl)

    def _collective(self, sl: torch.Tensor):
        # _stage and _shard are shared by all buckets.  That is race-free because every collective of this reducer is enqueued on ONE side
        # stream (and ProcessGroupNCCL serialises a communicator's collectives on its own stream behind it): bucket k+1's copy into _stage
        # is ordered after bucket k's copy back out of it.  A second side stream, or async_op=True, would need a staging pair per bucket.
        n = sl.numel()
        buf = sl
        if self.comm_dtype is not None and self.comm_dtype != sl.dtype:            # e.g. bf16 payload: half the bytes on the links
            if self._stage is None or self._stage.numel() < n or self._stage.device != sl.device:
                self._stage = torch.empty(n, dtype=self.comm_dtype, device=sl.device)
            buf = self._stage[:n]
            buf.copy_(sl)
        op = self.avg_op if self.avg_op is not None else dist.ReduceOp.SUM
        if self.algorithm == "rs_ag" and n % self.world == 0:
            # the reduced shard lands in a small staging buffer (1/world of the bucket), the all-gather writes the bucket in place
            m = n // self.world
            if self._shard is None or self._shard.numel() < m or self._shard.dtype != buf.dtype or self._shard.device != buf.device:
                self._shard = torch.empty(m, dtype=buf.dtype, device=buf.device)
            shard = self._shard[:m]
            dist.reduce_scatter_tensor(shard, buf, op=op, group=self.group)
            dist.all_gather_into_tensor(buf, shard, group=self.group)
        else:
            dist.all_reduce(buf, op=op, group=self.group)
        if self.avg_op is None:
            buf.div_(self.world)
        if buf is not sl:
            sl.copy_(buf)

    def finish(self, flat: torch.Tensor):
        """Order the caller's stream after the outstanding bucket reductions (no host wait)."""
        if flat.is_cuda and self._stream is not None:
            cur = torch.cuda.current_stream(flat.device)
            if self.record_timing and self._t_events:
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f0.record(cur)
                cur.wait_stream(self._stream)
                f1.record(cur)
                self._t_finish = (f0, f1)
            else:
                cur.wait_stream(self._stream)

    def timing(self):
        """Per-bucket timing of the last backward that reduced (record_timing = True), after a device synchronisation.
        range_ms[k]: compute-stream time of bucket k's block range (range k ends where its collective may start); collective_ms[k]: the
        bucket's reduction on the side stream; exposed_ms: how long the compute stream waited in finish() -- the part of the collectives
        the backward did NOT hide (ideally the last bucket's collective only)."""
        if not self._t_events or self._t_begin is None:
            return None
        torch.cuda.synchronize()
        range_ms, prev = [], self._t_begin
        for ev, _, _, _ in self._t_events:
            range_ms.append(round(prev.elapsed_time(ev), 4))
            prev = ev
        rec = {"buckets": len(self._t_events), "bucket_bytes": [b for _, _, _, b in self._t_events],
               "range_ms": range_ms, "collective_ms": [round(c0.elapsed_time(c1), 4) for _, c0, c1, _ in self._t_events],
               "collective_start_after_range_end_ms": [round(ev.elapsed_time(c0), 4) for ev, c0, _, _ in self._t_events],
               "exposed_ms": round(self._t_finish[0].elapsed_time(self._t_finish[1]), 4) if self._t_finish else None,
               "algorithm": self.algorithm, "comm_dtype": str(self.comm_dtype or torch.float32).replace("torch.", ""),
               "world": self.world, "backend": dist.get_backend(self.group)}
        return rec
