"""Training-mode forward/backward of the models through the HIP executors (csrc/tr_vit.hip, csrc/tr_train.hip).

The reference trains with `output = model(samples); loss = criterion(...); loss.backward()` (engine.py:50-76) and wraps the
model in DistributedDataParallel (train.py:405-407).  Here `model.train(); model(x)` runs `tr_vit_forward_train` (the forward
that keeps its activations on a tape) and returns logits that carry an autograd node; `loss.backward()` hands d logits to
`tr_vit_backward`, which writes every parameter gradient into ONE flat fp32 buffer.  `p.grad` of every parameter is a view
into that buffer, laid out in the order the backward finishes them (head, norm, blocks depth-1..0, embedding), so a
data-parallel reducer can all-reduce contiguous slices in place while the rest of the backward is still running
(dp.FlatGradReducer) -- no flatten/unflatten copies.

PyTorch is plumbing: memory, the stream, the autograd hook and the loss.  No arithmetic of the model happens in torch.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import torch

from . import _lib


def _align4(n: int) -> int:
    """Slot size of a parameter in the flat gradient buffer: 64 floats (256 B), so every bucket boundary is 16-byte aligned and
    divisible by the world size (reduce-scatter shards)."""
    return (n + 63) // 64 * 64


def backward_order(model) -> List[Tuple[str, torch.nn.Parameter]]:
    """(name, parameter) in the order the backward pass finishes their gradients; buckets are contiguous runs of this list."""
    named = dict(model.named_parameters())
    order, seen = [], set()

    def take(prefix):
        for n in named:
            if (n == prefix or n.startswith(prefix + ".")) and n not in seen:
                seen.add(n)
                order.append((n, named[n]))

    take("head")
    take("norm")
    for i in reversed(range(model.depth)):
        take(f"blocks.{i}")
    for n in list(named):            # family modules (score predictors, cluster layers): finished before the embedding
        if n not in seen and not n.startswith(("pos_embed", "cls_token", "patch_embed")):
            take(n)
    take("pos_embed")
    take("cls_token")
    take("patch_embed")
    return order


class TrainState:
    """Per-model training state: flat gradient buffer + views, transposed weights, tape and workspaces (one batch size)."""

    def __init__(self, model):
        dev = model.pos_embed.device
        self.order = backward_order(model)
        self.offsets, off = {}, 0
        for n, p in self.order:
            self.offsets[n] = off
            off += _align4(model._grad_slot_numel(n, p))
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.views = {n: self.flat[o: o + p.numel()].view_as(p) for (n, p), o in zip(self.order, self.offsets.values())}
        # bucket boundaries: [head+norm+last block] ... per block ... [block 0 + embedding]; event index that completes each
        self.block_slices = self._block_slices(model)
        self.B = None
        self.tape = self.bws = None
        self.key = None
        self.events = None
        self.gen = 0                 # counts training forwards: the tape belongs to the LAST one (see _VitTrainFn.backward)

    def _block_slices(self, model):
        """[(event_index, start, stop)] over the flat buffer: the slice event `e` of tr_vit_backward completes."""
        names = [n for n, _ in self.order]
        ends = {}
        for n, p in self.order:
            if n.startswith("blocks."):
                e = int(n.split(".")[1])
            elif n.startswith(("head", "norm")):
                e = model.depth - 1                     # finished before the last block's event
            else:
                e = model.depth                          # embedding + family modules: the final event
            ends[e] = self.offsets[n] + _align4(model._grad_slot_numel(n, p))
        out, start = [], 0
        for e in sorted(ends, key=lambda k: ends[k]):
            out.append((e, start, ends[e]))
            start = ends[e]
        del names
        return out

    def grads_struct(self, model):
        """tr_vit_weights-shaped struct whose pointers are the gradient views."""
        G = _lib.TrVitWeights()
        v = self.views

        def ptr(name):
            return v[name].data_ptr()

        G.patch_w, G.patch_b = ptr("patch_embed.proj.weight"), ptr("patch_embed.proj.bias")
        G.cls_token, G.pos_embed = ptr("cls_token"), ptr("pos_embed")
        G.norm_g, G.norm_b = ptr("norm.weight"), ptr("norm.bias")
        G.head_w, G.head_b = ptr("head.weight"), ptr("head.bias")
        for i in range(model.depth):
            b, pre = G.blocks[i], f"blocks.{i}."
            b.ln1_g, b.ln1_b = ptr(pre + "norm1.weight"), ptr(pre + "norm1.bias")
            b.qkv_w, b.qkv_b = ptr(pre + "attn.qkv.weight"), ptr(pre + "attn.qkv.bias")
            b.proj_w, b.proj_b = ptr(pre + "attn.proj.weight"), ptr(pre + "attn.proj.bias")
            b.ln2_g, b.ln2_b = ptr(pre + "norm2.weight"), ptr(pre + "norm2.bias")
            b.fc1_w, b.fc1_b = ptr(pre + "mlp.fc1.weight"), ptr(pre + "mlp.fc1.bias")
            b.fc2_w, b.fc2_b = ptr(pre + "mlp.fc2.weight"), ptr(pre + "mlp.fc2.bias")
        model._grad_stage_ptrs(G, ptr)
        return G

    def transposed(self, model, pk):
        """bf16 transposed copies of the block matrices (the dgrad GEMM operands): made by the model's own repack (models._pack with
        need_transposed: persistent buffers rewritten by the same fused launch as the operand copies); the reduction modules' few small
        matrices are transposed here whenever the pack generation changes."""
        if self.key == pk["gen"]:
            return self.wt
        WT = _lib.TrVitWeights()
        keep = []

        def t16(w):
            c = w.detach().t().to(torch.bfloat16).contiguous()
            keep.append(c)
            return c.data_ptr()

        for i, (tq, tp_, t1, t2) in enumerate(pk["tblocks"]):
            b = WT.blocks[i]
            b.qkv_w, b.proj_w, b.fc1_w, b.fc2_w = tq.data_ptr(), tp_.data_ptr(), t1.data_ptr(), t2.data_ptr()
        model._transposed_stage_weights(WT, t16)
        self.wt, self.wt_keep, self.key = WT, keep, pk["gen"]
        return WT

    def buffers(self, model, pk, B, dev):
        if self.B != B:
            lib = _lib.load()
            nt = lib.tr_vit_tape_bytes(C.byref(pk["cfg"]), B)
            nb = lib.tr_vit_backward_workspace_bytes(C.byref(pk["cfg"]), B)
            if nt == 0 or nb == 0:
                raise NotImplementedError(
                    f"{type(model).__name__}: this configuration has no training path in the HIP executor (built: every family at "
                    "224x224 and 384x384, up to 640 tokens; bf16); call model.eval() for inference")
            self.tape = torch.empty(nt, dtype=torch.uint8, device=dev)
            self.bws = torch.empty(nb, dtype=torch.uint8, device=dev)
            self.B = B
        return self.tape, self.bws


TAPE_FIELDS = ("x0", "x1", "xn1", "qkv", "ao", "dattn", "x2", "xn2", "pre", "h", "idx", "idx2", "scores", "size",
               "n_pre", "n_att", "n_mlp", "kk")


def tape_layout(model, blk: int) -> dict:
    """Offsets / token counts of block `blk` on the tape of the last training forward."""
    st = model._train_state()
    out = (C.c_size_t * 18)()
    _lib.check(_lib.load().tr_vit_tape_layout(C.byref(model._packed["cfg"]), st.B, blk, out), "tr_vit_tape_layout")
    return dict(zip(TAPE_FIELDS, (int(v) for v in out)))


def train_decisions(model) -> dict:
    """{blk: decision tensors} of the last training forward, read back from the tape (int64 on the host side):
    Top-K / EViT: idx [B,K]; ToMe: (unm [B,na-r], src [B,r], dst [B,r]); DPC-KNN: (centres [B,K], assignment [B,P_in]); ATS: ids."""
    st = model._train_state()
    B, out = st.B, {}
    for blk in range(model.depth):
        lay = tape_layout(model, blk)
        k = lay["kk"]
        if k <= 0:
            continue
        raw = st.tape[lay["idx"]: lay["idx"] + 4 * B * lay["n_att"]].view(torch.int32)
        if model._family in (_lib.TR_FAMILY_TOPK, _lib.TR_FAMILY_EVIT):
            out[blk] = raw[: B * k].view(B, k).long()
        elif model._family == _lib.TR_FAMILY_TOME:
            na = (lay["n_att"] + 1) // 2
            out[blk] = (raw[: B * (na - k)].view(B, na - k).long(), raw[B * (na - k): B * na].view(B, k).long(),
                        raw[B * na: B * (na + k)].view(B, k).long())
        elif model._family == _lib.TR_FAMILY_DPCKNN:          # (centres [B,K], assignment [B,P_in])
            p_in = lay["n_pre"] - 1
            raw2 = st.tape[lay["idx2"]: lay["idx2"] + 4 * B * p_in].view(torch.int32)
            out[blk] = (raw[: B * k].view(B, k).long(), raw2.view(B, p_in).long())
        elif model._family == _lib.TR_FAMILY_KMEDOIDS:        # medoid ids [B,K]
            out[blk] = raw[: B * k].view(B, k).long()
        elif model._family == _lib.TR_FAMILY_ATS:             # ids [B,Ks]: CLS id 0 first, 1-based token ids, 0 padding
            out[blk] = raw[: B * k].view(B, k).long()
        elif model._family == _lib.TR_FAMILY_DYVIT:           # the Gumbel one-hot's first entry per patch token [B,P] (before * prev_decision)
            n = lay["n_att"]
            out[blk] = st.tape[lay["scores"]: lay["scores"] + 4 * B * n].view(torch.float32).view(B, n)[:, 1:].clone()
    return out


def drop_path_scales(model, B: int, dev):
    """DropPath (timm 0.4.12 drop_path, call sites topk.py:78,87,95): block i drops a branch of an image with probability
    dpr[i] = linspace(0, drop_path_rate, depth)[i] (topk.py:157) and scales the survivors by 1/keep_prob; two independent draws per
    block (attention branch, MLP branch).  Returns fp32 [2*depth, B] of scales, or None when drop_path_rate == 0.
    `model.drop_path_draws` (tests): the uniform draws [2*depth, B] to use instead of torch.rand (the reference's, replayed)."""
    if not model.drop_path_rate:
        return None
    dpr = torch.linspace(0, model.drop_path_rate, model.depth)
    keep = (1.0 - dpr).repeat_interleave(2).to(dev).unsqueeze(1)                      # [2*depth, 1]
    draws = getattr(model, "drop_path_draws", None)
    u = torch.rand(2 * model.depth, B, device=dev) if draws is None else draws.to(device=dev, dtype=torch.float32).reshape(2 * model.depth, B)
    scale = (keep + u).floor() / keep                                                   # x.div(keep) * floor(keep + rand)
    scale[keep.squeeze(1) >= 1.0] = 1.0                                                 # drop_prob 0 (block 0): DropPath is the identity
    return scale.contiguous()


def dropout_keep_mask(model, pk, B: int, dev):
    """Dropout (timm's drop_rate; train.py:46 --drop): the keep mask of one training forward, uint8 (1 = keep), one byte per element in
    the order the executor consumes it -- the embedded tokens [B,N0,D] (pos_drop, topk.py:186), then per block proj's output rows
    (proj_drop, :53), the Mlp's hidden layer and its output (timm Mlp's two nn.Dropout): the order in which the reference module draws.
    Returns None when drop_rate == 0.  `model.dropout_draws` (tests): the reference's recorded keep masks, in call order, used instead
    of a fresh Bernoulli(1 - p) draw on the device."""
    if not model.drop_rate:
        return None
    n = int(_lib.load().tr_vit_dropout_mask_bytes(C.byref(pk["cfg"]), B))
    if n == 0:
        raise NotImplementedError(f"{type(model).__name__}: no training path for this configuration")
    draws = getattr(model, "dropout_draws", None)
    if draws is not None:
        keep = torch.cat([torch.as_tensor(d).reshape(-1).to(torch.uint8) for d in draws]).to(dev)
        if keep.numel() != n:
            raise ValueError(f"dropout_draws hold {keep.numel()} elements, this forward consumes {n}")
        return keep.contiguous()
    return torch.empty(n, dtype=torch.uint8, device=dev).bernoulli_(1.0 - float(model.drop_rate))


class _VitTrainFn(torch.autograd.Function):
    """logits (+ DyViT's extra outputs) = model(x) with the backward wired to tr_vit_backward.  `anchor` only makes autograd call
    backward.  Outputs: (logits,) or for DyViT (logits, pred_0 .. pred_{S-1} [, features]): pred_j = the stage's hard keep decision
    [B,P] (straight-through differentiable, dyvit.py:223-225), features = the final norm of the patch tokens [B,P,D]."""

    @staticmethod
    def forward(ctx, anchor, x, model):
        lib = _lib.load()
        pk = model._pack(need_transposed=True)
        cfg = pk["cfg"]
        B = x.shape[0]
        st = model._train_state()
        tape, bws = st.buffers(model, pk, B, x.device)
        st.gen += 1
        ctx.gen = st.gen
        ws = model._workspace(B, x.device)
        logits = torch.empty(B, model._classes_padded, dtype=torch.float32, device=x.device)
        tokens = (C.c_int * model.depth)()
        dyvit = model._family == _lib.TR_FAMILY_DYVIT
        distill = dyvit and bool(getattr(model, "dyvit_distillation", False))
        feats = torch.empty(B, model.patch_embed.num_patches + 1, model.embed_dim, dtype=torch.float32, device=x.device) if distill else None
        model._noise_slot = 0
        noise = model._gumbel_ptr(B, x.device) if dyvit else model._noise_ptr(B, x.device)
        ctx.drop = drop_path_scales(model, B, x.device)
        ctx.keep_mask = dropout_keep_mask(model, pk, B, x.device)
        with torch.cuda.device(x.device):
            rc = lib.tr_vit_forward_train(C.byref(cfg), C.byref(pk["W"]), x.data_ptr(), logits.data_ptr(), ws["buf"].data_ptr(), ws["nbytes"],
                                          tape.data_ptr(), tape.numel(), noise, None if feats is None else feats.data_ptr(),
                                          None if ctx.drop is None else ctx.drop.data_ptr(), tokens, B, torch.cuda.current_stream().cuda_stream,
                                          None if ctx.keep_mask is None else ctx.keep_mask.data_ptr(), float(model.drop_rate or 0.0))
        _lib.check(rc, "tr_vit_forward_train")
        model._last_tokens = list(tokens)
        ctx.model, ctx.B, ctx.pk = model, B, pk
        ctx.keep = x
        ctx.n_pred, ctx.distill = 0, distill
        if model._classes_padded != model.num_classes:          # the padded classifier columns never leave the executor
            logits = logits[:, :model.num_classes].contiguous()
        if not dyvit:
            return logits
        preds = []
        for blk in sorted(model.pruning_loc):
            lay = tape_layout(model, blk)
            n = lay["n_att"]
            pol = tape[lay["size"]: lay["size"] + 4 * B * n].view(torch.float32).view(B, n)
            preds.append(pol[:, 1:].clone())
        ctx.n_pred = len(preds)
        outs = [logits] + preds + ([feats[:, 1:].clone()] if distill else [])
        return tuple(outs)

    @staticmethod
    def backward(ctx, dlogits, *dextra):
        model, B, pk = ctx.model, ctx.B, ctx.pk
        lib = _lib.load()
        st = model._train_state()
        if ctx.gen != st.gen:
            # the tape, the Gumbel / DPC-KNN noise and the workspace belong to the model, not to the autograd node: a later train-mode
            # forward has overwritten the activations and decisions this backward would read
            raise RuntimeError(
                f"{type(model).__name__}: backward() of training forward #{ctx.gen}, but forward #{st.gen} has run since and overwritten "
                "the activation tape (one tape per model).  Call loss.backward() before the next model(x) in train mode; for several "
                "forwards per step (two views, accumulated losses) run forward + backward per view and let the gradients accumulate, or "
                "use model.eval() / torch.no_grad() for the forwards that need no gradient.")
        dev = st.flat.device
        if dlogits is None:
            dlogits = torch.zeros(B, model.num_classes, dtype=torch.float32, device=dev)
        dl = dlogits.detach().to(torch.float32).contiguous()
        if model._classes_padded != model.num_classes:          # zero gradient on the padded classifier columns
            dlp = torch.zeros(B, model._classes_padded, dtype=torch.float32, device=dev)
            dlp[:, :model.num_classes] = dl
            dl = dlp
        dpred = dfeat = None
        if ctx.n_pred:
            P = model.patch_embed.num_patches
            gp = [torch.zeros(B, P, dtype=torch.float32, device=dev) if g is None else g.detach().to(torch.float32) for g in dextra[:ctx.n_pred]]
            dpred = torch.stack(gp).contiguous()                              # [stages, B, P]
            if ctx.distill and dextra[ctx.n_pred] is not None:
                dfeat = torch.zeros(B, P + 1, model.embed_dim, dtype=torch.float32, device=dev)
                dfeat[:, 1:] = dextra[ctx.n_pred].detach()
        # gradient views: when NO parameter holds a gradient yet (zero_grad(set_to_none=True), torch's default) the backward OVERWRITES
        # the flat buffer -- nothing to clear, nothing to read back; otherwise the slices of the parameters without a gradient are zeroed
        # and the backward accumulates (engine.py:41-84: gradient accumulation over micro-steps).  Every parameter of the executor is
        # written exactly once per backward (csrc/tr_train.hip); parameters the executor does not touch keep the zeros they were born with.
        fresh = [n for n, p in st.order if p.grad is None or p.grad.data_ptr() != st.views[n].data_ptr()]
        accumulate = 0 if len(fresh) == len(st.order) else 1
        if accumulate:
            for n in fresh:
                st.views[n].zero_()
        G = st.grads_struct(model)
        WT = st.transposed(model, pk)
        reducer = getattr(model, "_grad_reducer", None)
        reduce_now = reducer is not None and reducer.sync
        # one call for the whole model, or -- under a gradient reducer -- one call per bucket's block range, each followed by
        # that bucket's in-place reduction on the reducer's stream while the next range runs
        ranges = reducer.plan(st.block_slices, model.depth) if reduce_now else [(model.depth - 1, 0, 0, st.flat.numel())]
        if reduce_now:
            reducer.begin(st.flat) if hasattr(reducer, "begin") else setattr(reducer, "launched", [])
        stream = torch.cuda.current_stream().cuda_stream
        with torch.cuda.device(dl.device):
            for hi, lo, start, stop in ranges:
                rc = lib.tr_vit_backward(C.byref(pk["cfg"]), C.byref(pk["W"]), C.byref(WT), C.byref(G), dl.data_ptr(),
                                         None if dpred is None else dpred.data_ptr(), None if dfeat is None else dfeat.data_ptr(),
                                         None if ctx.drop is None else ctx.drop.data_ptr(), st.tape.data_ptr(), st.tape.numel(), st.bws.data_ptr(), st.bws.numel(), accumulate, hi, lo, B, stream,
                                         None if ctx.keep_mask is None else ctx.keep_mask.data_ptr(), float(model.drop_rate or 0.0))
                _lib.check(rc, "tr_vit_backward")
                if reduce_now:
                    reducer.reduce_slice(st.flat, start, stop)
        if reduce_now:
            reducer.finish(st.flat)
        for n, p in st.order:
            if not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = st.views[n]
            elif p.grad.data_ptr() != st.views[n].data_ptr():
                p.grad.add_(st.views[n])
        was_dirty = bool(getattr(model, "_weights_dirty", False))
        model.weights_changed()          # an optimizer step follows; fused optimizers do not bump the version counters the pack cache reads
        model._dirty_by_backward = not was_dirty      # dirt that was there before this backward is somebody else's: only a full repack clears it
        return None, None, None


def train_forward(model, x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError(f"input is on {x.device}: tokenreduction_amd has no CPU path (HIP kernels only)")
    if model.precision != "bf16":
        raise NotImplementedError("the training path is bf16 (fp32 accumulate); set model.precision = 'bf16'")
    if model.attn_drop_rate:
        raise NotImplementedError("attn_drop_rate (dropout on the attention probabilities, topk.py:49) is not built in the HIP training path; "
                                  "the reference's command line cannot set it either (train.py:46 exposes --drop only, which IS supported, "
                                  "as is --drop-path)")
    x = x.detach().to(torch.float32).contiguous()
    anchor = torch.empty(0, dtype=torch.float32, device=x.device, requires_grad=True)
    out = _VitTrainFn.apply(anchor, x, model)
    if model._family != _lib.TR_FAMILY_DYVIT:
        return out
    # dyvit.py:257-261: (x, features, prev_decision.detach(), out_pred_prob) with the DyViT distillation scheme, else (x, out_pred_prob)
    n_st = len(model.pruning_loc)
    logits, preds = out[0], list(out[1: 1 + n_st])
    if getattr(model, "dyvit_distillation", False):
        return logits, out[1 + n_st], preds[-1].detach().unsqueeze(-1), preds
    return logits, preds
