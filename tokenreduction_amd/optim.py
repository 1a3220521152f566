"""AdamW for the HIP training path in ONE launch per step (csrc/tr_optim.hip).

The reference builds `torch.optim.AdamW` (optim.py:102-140 through timm's factory) and steps it after `loss.backward()`
(engine.py:76-91).  `FusedAdamW` is a `torch.optim.Optimizer` with the same constructor, parameter groups (`lr`, `weight_decay`,
`betas`, `eps` per group -- schedulers and `finetune.frozen_lr` edit them as usual) and state names (`step`, `exp_avg`,
`exp_avg_sq`), whose `step()` is one kernel over all parameters that

  * applies torch's fused-AdamW arithmetic, expression by expression (parameters stay bit-identical to
    `torch.optim.AdamW(fused=True)`: tests/test_hip_train.py),
  * optionally (`zero_grads=True`) zeroes the gradient it consumed -- not needed for training.TrainState, whose backward overwrites
    the flat gradient buffer after `zero_grad(set_to_none=True)`,
  * rewrites the bf16 operand copy and the transposed copy of every matrix the executor reads, when constructed with `model=` -- the
    next forward then finds its operands fresh and skips tr_cast_pack_bf16.

Per training step of DeiT-B this replaces torch's multi_tensor_apply kernels, ~69 fill launches and the cast-pack pass.
Not supported (raises): amsgrad, maximize, a gradient scaler, more than 8 parameter groups per step count.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib

_ITEM = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("dst", "<u8"), ("dst_t", "<u8"), ("rows", "<i4"), ("cols", "<i4"),
                  ("group", "<i4"), ("pad", "<i4")])


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, maximize=False, model=None, zero_grads=False):
        if amsgrad or maximize:
            raise NotImplementedError("FusedAdamW: amsgrad / maximize are not built")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._model = model
        self._zero_grads = bool(zero_grads)
        self._table = None

    # ---- state: exp_avg / exp_avg_sq of all parameters in two flat buffers (views per parameter, like the gradients)
    def _init_state(self, ps):
        dev = ps[0].device
        offs, off = [], 0
        for p in ps:
            offs.append(off)
            off += (p.numel() + 63) // 64 * 64
        m = torch.zeros(off, dtype=torch.float32, device=dev)
        v = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(ps, offs):
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise NotImplementedError("FusedAdamW: fp32 contiguous parameters only")
            st = self.state[p]
            st["step"] = 0
            st["exp_avg"] = m[o: o + p.numel()].view_as(p)
            st["exp_avg_sq"] = v[o: o + p.numel()].view_as(p)

    def _operand_slots(self):
        """{parameter storage address: (bf16 copy address, transposed copy address or 0)} of the model's packed matrices, if every bf16
        operand of the model is refreshed by the fused pack table (else None: the model keeps refreshing them itself)."""
        m = self._model
        if m is None or getattr(m, "precision", "bf16") != "bf16":
            return None
        tab = m.__dict__.get("_pack_table")
        if tab is None or not getattr(m, "_pack_all_fused", False):
            return None
        return {sp: (dp, tp_) for sp, dp, tp_, _r, _c in tab["sig"]}

    def _build(self, dev):
        slots = self._operand_slots()
        items, key = [], []
        n_refresh = 0
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.grad is None:
                    continue
                if p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                    raise NotImplementedError("FusedAdamW: fp32 contiguous gradients only")
                st = self.state[p]
                for k in ("exp_avg", "exp_avg_sq"):      # a loaded state_dict brings its own tensors: they are used where they are
                    if st[k].dtype != torch.float32 or not st[k].is_contiguous() or st[k].device != p.device:
                        raise NotImplementedError(f"FusedAdamW: optimizer state {k} must be fp32, contiguous and on the parameter's device")
                rows, cols = (p.shape[0], p.numel() // p.shape[0]) if p.dim() >= 2 else (1, p.numel())
                dst, dst_t = (slots or {}).get(p.data_ptr(), (0, 0))
                n_refresh += 1 if dst else 0
                items.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), dst, dst_t, rows, cols, gi, 0))
                key.append((p.data_ptr(), p.grad.data_ptr(), dst, dst_t, gi, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()))
        key = tuple(key)
        if self._table is not None and self._table["key"] == key:
            return self._table
        arr = np.array(items, dtype=_ITEM)
        first = np.zeros(len(items) + 1, dtype=np.int32)
        for n, it in enumerate(items):
            first[n + 1] = first[n] + ((it[6] + 63) // 64) * ((it[7] + 63) // 64)
        self._table = dict(key=key, items=torch.from_numpy(arr.view(np.uint8).copy()).to(dev), first=torch.from_numpy(first).to(dev),
                           n=len(items), tiles=int(first[-1]),
                           # does this step rewrite EVERY bf16 operand copy the executor reads?  (a matrix without a gradient this step --
                           # frozen, or outside this optimizer -- keeps its old copy: the model then refreshes for itself)
                           refreshes=slots is not None and n_refresh == len(slots),
                           params=[p for g in self.param_groups for p in g["params"] if p.grad is not None])
        return self._table

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        fresh = [p for g in self.param_groups for p in g["params"] if "exp_avg" not in self.state[p]]
        if fresh:                                 # first step, or a group added since (add_param_group): their moments start at zero
            self._init_state(fresh)
        if len(self.param_groups) > 8:
            raise NotImplementedError("FusedAdamW: at most 8 parameter groups")
        betas = {g["betas"] for g in self.param_groups}
        epss = {g["eps"] for g in self.param_groups}
        if len(betas) != 1 or len(epss) != 1:
            raise NotImplementedError("FusedAdamW: betas and eps must be the same in every parameter group")
        (beta1, beta2), eps = next(iter(betas)), next(iter(epss))
        ps = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not ps:
            return loss
        dev = ps[0].device
        tab = self._build(dev)
        steps = {int(self.state[p]["step"]) for p in tab["params"]}          # (a loaded state_dict may carry tensors here)
        if len(steps) != 1:
            raise NotImplementedError("FusedAdamW: parameters whose step counts differ (a parameter that had no gradient in some steps)")
        step = next(iter(steps)) + 1
        bc1 = 1.0 - math.pow(beta1, step)
        bc2_sqrt = math.sqrt(1.0 - math.pow(beta2, step))
        import ctypes as C
        lr8 = (C.c_double * 8)(*([float(g["lr"]) for g in self.param_groups] + [0.0] * (8 - len(self.param_groups))))
        wd8 = (C.c_double * 8)(*([float(g["weight_decay"]) for g in self.param_groups] + [0.0] * (8 - len(self.param_groups))))
        with torch.cuda.device(dev):
            _lib.check(_lib.load().tr_adamw_step(tab["items"].data_ptr(), tab["first"].data_ptr(), tab["n"], tab["tiles"], float(beta1), float(beta2),
                                                 float(eps), bc1, bc2_sqrt, lr8, wd8, 1 if self._zero_grads else 0,
                                                 torch.cuda.current_stream().cuda_stream), "tr_adamw_step")
        for p in tab["params"]:
            self.state[p]["step"] = step
        m = self._model
        if m is not None:
            # every operand copy was rewritten by the step (fp32 operands are read in place) -- but only the dirt this optimizer's own backward
            # left may be declared clean: a weights_changed() raised for another reason since (a manual p.data edit, an EMA / teacher copy
            # into a matrix this step did not touch) must survive
            if tab["refreshes"] and getattr(m, "_dirty_by_backward", False):
                m._weights_dirty = False
                m._dirty_by_backward = False
                m._mlp_pk_stale = True             # the fused eval Mlp's fragment-major copies derive from the bf16 copies just rewritten
            else:
                m.weights_changed()
        return loss

    def zero_grad(self, set_to_none: bool = True):
        """With set_to_none (torch's default) the views are dropped: the next backward of training.TrainState overwrites them."""
        if set_to_none:
            for g in self.param_groups:
                for p in g["params"]:
                    p.grad = None
        else:
            super().zero_grad(set_to_none=False)
