"""Host-side mirror of the reference's model classes for the hot path.

Same constructor signature, attribute names, state-dict keys, helper methods and forward return
contract as
    models/deit_viz.py:75-212   VisionTransformer            (timm-0.4.12 ViT, the DeiT trunk)
    models/topk.py:102-212      TopKVisionTransformer
    models/evit.py:132-244      EfficientVisionTransformer
so a `train.py` / `validate.py`-style driver can swap `timm.models.create_model` for
`tokenreduction_amd.create_model` unchanged.  The nn.Modules below only HOLD parameters (so
load_state_dict / state_dict / optimizers see the reference's key names); `forward` never calls
them: it hands the image batch to the gfx950 executor (csrc/tr_vit.hip) through the C ABI.

There is no CPU path: forward() on a CPU tensor raises.  In train mode forward() runs the training executor
(activations kept on a tape) and `loss.backward()` runs the HIP backward executor (training.py) -- every family and
every factory width, 224 x 224 and 384 x 384; what the training path refuses (attn_drop_rate, the distillation token,
more than 640 tokens) raises there.
"""
from __future__ import annotations

import copy
import ctypes as C
import warnings
from functools import partial
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class PatchEmbed(nn.Module):
    """Parameter holder with timm PatchEmbed's attribute surface (num_patches, grid_size, proj)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size = _pair(img_size), _pair(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)


class _Pending:
    """Handle of a forward enqueued by VisionTransformer.forward_async: result() makes the caller's CURRENT stream wait for it."""

    def __init__(self, out, event):
        self._out, self._event = out, event

    def result(self):
        if self._event is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self._event)
            for t in (self._out if isinstance(self._out, (tuple, list)) else (self._out,)):
                if torch.is_tensor(t):
                    t.record_stream(cur)
            self._event = None
        return self._out


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=True, keep_rate=1.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.keep_rate = keep_rate
        self.init_n = 14 * 14   # topk.py:40 -- hard-coded in the reference, independent of img_size


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm, keep_rate=1.0):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, qkv_bias, keep_rate)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


def _init_vit_weights(m: nn.Module):
    """deit_viz.py:215-247 with name='' / jax_impl=False (what `self.apply(self._init_weights)` does)."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)


class VisionTransformer(nn.Module):
    """DeiT trunk, no reduction (deit_viz.py:75-212); also the base class of the reduction models."""

    _family = _lib.TR_FAMILY_DEIT
    _blocks_last = False
    GRAPH_CACHE = 8              # captured graphs kept per workspace (one per input buffer / output set)
    # consecutive forwards with a never-seen input address before a workspace stops capturing hipGraphs: one more than the cache holds, so
    # that a caller rotating up to GRAPH_CACHE static input buffers (prefetch ring, multi-crop eval) gets through its first pass
    GRAPH_MISS_LIMIT = GRAPH_CACHE + 1

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, representation_size=None, distilled=False,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., embed_layer=None, norm_layer=None,
                 act_layer=None, weight_init='', args=None):
        super().__init__()
        if distilled:
            # the reference's forwards only ever concatenate cls_token -- deit_viz.py:186-189 adds the [1, P + 2, D] position
            # embedding of a distilled model to P + 1 tokens (a broadcast error at the first forward), and head_dist is only created by
            # reset_classifier (deit_viz.py:178-181): a distilled model cannot run in the reference either
            raise NotImplementedError("distilled (dist_token) models are not on the hot path: the reference's own forward fails for them "
                                      "(deit_viz.py:186-189 adds a [1, P + 2, D] position embedding to P + 1 tokens)")
        if representation_size:
            raise NotImplementedError("representation_size / pre_logits is not used by any registered factory")
        if not qkv_bias:
            raise NotImplementedError("every registered factory uses qkv_bias=True")
        if act_layer not in (None, nn.GELU):
            raise NotImplementedError("only nn.GELU (erf) is implemented in the fc1 epilogue")
        if embed_dim != 64 * num_heads:
            raise ValueError("head_dim must be 64 (192/3, 384/6, 768/12 in models_act.py)")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        self.depth = depth
        self.num_heads = num_heads
        self.drop_rate, self.attn_drop_rate, self.drop_path_rate = drop_rate, attn_drop_rate, drop_path_rate
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = None
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + self.num_tokens, embed_dim))
        self.mlp_ratio, self.qkv_bias, self._norm_layer = mlp_ratio, qkv_bias, norm_layer
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        # registration order = parameter order = the positional keys of an optimizer state_dict.  The reference's topk / evit / tome /
        # dyvit / kmedoids classes delete the base class's `blocks` and build their own AFTER norm and head exist (topk.py:150-158):
        # their parameters run patch_embed, norm, head, blocks; the other families keep timm's order (deit_viz.py:113-119)
        if self._blocks_last:
            self._modules["blocks"] = self._modules.pop("blocks")
        self.viz_mode = getattr(args, 'viz_mode', False)
        self._keep = [0] * depth
        # "bf16" = the product path; "fp32" = validation path (reference arithmetic on the GPU, VALU); "bf16x3" = the fp32 executor
        # with its Linears and attention on the matrix cores as split-bf16 products (3 MFMAs per product, ~1e-5 relative per Linear)
        self.precision = "bf16"
        self.use_graph = True        # eval forward replays a captured hipGraph (False: plain launches)
        self._packed = None
        self._ws = {}
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.apply(_init_vit_weights)

    # ---- copies -----------------------------------------------------------------------------------
    # executor caches (ctypes structs, GPU workspaces with captured hipGraphs, the tape / flat gradient buffer, noise buffers): state of
    # THIS object's executor, rebuilt on demand.  A copy (copy.deepcopy: ModelEma, torch's swa_utils) starts without them: captured
    # torch.cuda.CUDAGraph objects cannot be deep-copied at all, and a copied workspace would not belong to the copy's own packed weights.
    _EXECUTOR_CACHES = {"_packed": None, "_ws": None, "_last_ws": None, "_tstate": None, "_grad_reducer": None, "_noise_buf": None,
                        "_gumbel_buf": None, "_kmed_draws": None, "_pack_slots": None, "_pack_table": None, "_mlp_pack_items": None,
                        "_pipe_streams": None, "_pipe_next": None, "_noise_bufs": None}

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in self._EXECUTOR_CACHES:
                if k in ("_pack_slots", "_pack_table", "_pipe_streams", "_pipe_next"):
                    continue
                new.__dict__[k] = {} if k == "_ws" else None
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    # ---- reference helper surface -------------------------------------------------------------
    def _set_keep(self, loc, k, rule):
        """Token / cluster count of the reduction at block `loc`.  0 is this package's "no reduction here" marker, so a schedule that
        rounds down to zero tokens (int(0.3**5 * 196) = 0: the reference would go on with the CLS token alone) raises instead of being
        skipped silently."""
        k = int(k)
        if k < 1:
            raise ValueError(f"block {loc}: {rule} keeps {k} tokens -- schedules that leave no patch token / cluster are not built")
        self._keep[loc] = k

    def _init_weights(self, m):
        _init_vit_weights(m)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'dist_token'}

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=''):
        self.num_classes = num_classes
        dev = self.pos_embed.device
        self.head = (nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()).to(dev)
        self._packed = None
        self._tstate = None          # the flat gradient buffer is laid out for the old head

    @property
    def _classes_padded(self):
        """The classifier as the kernels see it: rows padded with zeros to a multiple of 8 (any --num_classes works: NABirds 555,
        NUS-WIDE 81, train.py:334); the padded logits columns are cut off again before anything is returned."""
        return (self.num_classes + 7) // 8 * 8

    def get_new_module_names(self):
        return []

    def get_reduction_count(self):
        return getattr(self, "pruning_loc", [])

    # ---- packing: bf16 weight copies + the C structs ------------------------------------------------
    def _param_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def sync_pipeline(self):
        """Makes the caller's current stream wait for every forward that forward_async has put on its side streams.  Called before the
        operand copies are rewritten (_pack) and when the model goes to train mode; call it yourself before you modify parameters in place
        while handles may be outstanding -- like any tensor used on another stream, the weights of a forward in flight must not change under it."""
        for dev, streams in (self.__dict__.get("_pipe_streams") or {}).items():
            cur = torch.cuda.current_stream(dev)
            for st in streams:
                cur.wait_stream(st)

    def train(self, mode: bool = True):
        if mode:
            self.sync_pipeline()
        return super().train(mode)

    def weights_changed(self):
        """Tell the executor that parameter VALUES changed behind autograd's version counters.  Automatic after every backward pass (an
        optimizer step follows: torch's fused optimizers -- AdamW(fused=True) -- update the parameters without bumping `_version`, which
        round 2's packing cache keyed on: the bf16 operand copies went stale and training silently ran on the initial matrices).  Call it
        yourself after writing into `p.data` through an API that does not bump versions."""
        self._weights_dirty = True
        self._dirty_by_backward = False          # (training._VitTrainFn.backward sets it after its own call: optim.FusedAdamW may clear only that)

    def _pack(self, need_transposed=False):
        """bf16 (or fp32) operand copies of every parameter the executor reads + the C structs.  The copies live in PERSISTENT buffers:
        a repack after an optimizer step rewrites them in place -- one fused launch for all matrices (tr_cast_pack_bf16), transposed
        copies for the data-gradient GEMMs included once training asked for them -- so addresses, workspaces and captured graphs survive."""
        if self.precision not in ("bf16", "fp32", "bf16x3"):
            raise ValueError("precision must be 'bf16', 'bf16x3' or 'fp32'")
        self._pre_pack()
        if need_transposed:
            self._want_transposed = True
        want_t = bool(getattr(self, "_want_transposed", False)) and self.precision == "bf16"
        key = (self.precision, want_t) + self._param_key()
        if self._packed is not None and self._packed["key"] == key and not getattr(self, "_weights_dirty", False):
            return self._packed
        self.sync_pipeline()        # the operand copies are rewritten in place: no forward_async forward may still be reading them
        dev = self.pos_embed.device
        if dev.type != "cuda":
            raise RuntimeError(f"model is on {dev}: tokenreduction_amd runs on MI355X only (no CPU path); call .cuda()")
        if not isinstance(self.head, nn.Linear):
            raise NotImplementedError("num_classes == 0 (headless) is not supported by the executor")
        keep_alive = []
        wdt = torch.bfloat16 if self.precision == "bf16" else torch.float32
        slots = self.__dict__.setdefault("_pack_slots", {})          # persistent operand buffers: {(call index, kind): tensor}
        own = {p_.untyped_storage().data_ptr() for p_ in self.parameters()}      # storages that ARE parameters: always current, never copied

        def live(t):
            return t.dtype == torch.float32 and t.is_contiguous() and t.device == dev and t.untyped_storage().data_ptr() in own

        state = {"i": 0, "moved": False, "copied": False}
        fused = []                                                     # (fp32 source, bf16 slot, transposed bf16 slot or None)

        def slot(kind, shape, dtype):
            k = (state["i"], kind)
            t = slots.get(k)
            if t is None or t.shape != torch.Size(shape) or t.dtype != dtype or t.device != dev:
                t = slots[k] = torch.zeros(shape, dtype=dtype, device=dev)
                state["moved"] = True
            return t

        def w16(t, transposed=False):      # weight matrix in the executor's operand type; transposed=True: also its [cols, rows] copy
            state["i"] += 1
            t = t.detach()
            if wdt == torch.float32:
                if live(t):
                    keep_alive.append(t)
                    return t.data_ptr()           # the parameter's own storage: always current
                c = slot("w", t.shape, torch.float32)
                c.copy_(t)
                return c.data_ptr()
            c = slot("w", t.shape, torch.bfloat16)
            ct = slot("t", (t.shape[1], t.shape[0]), torch.bfloat16) if (transposed and want_t) else None
            # the fused cast kernel moves 16-byte vectors: rows must be whole vectors and start on one
            if t.dim() == 2 and live(t) and t.numel() >= 4096 and t.shape[1] % 4 == 0 and t.data_ptr() % 16 == 0 and (ct is None or t.shape[0] % 4 == 0):
                fused.append((t, c, ct))
            else:
                c.copy_(t)
                state["copied"] = True        # an operand copy the fused table does not cover (optim.FusedAdamW then leaves the refresh here)
                if ct is not None:
                    ct.copy_(t.t())
            if ct is not None:
                tslots[state["i"]] = ct
            return c.data_ptr()

        def f32(t):
            state["i"] += 1
            t = t.detach()
            if live(t):
                keep_alive.append(t)
                return t.data_ptr()               # the parameter's own storage: always current
            c = slot("f", t.shape, torch.float32)
            c.copy_(t)
            return c.data_ptr()

        tslots = {}
        W = _lib.TrVitWeights()
        D = self.embed_dim
        Hd = self.blocks[0].mlp.fc1.out_features
        lib = _lib.load()
        mlp_pk_bytes = int(lib.tr_mlp_pack_bytes(D, Hd)) if lib.tr_mlp_fused_supported(D, Hd) else 0
        mlp_items = []
        W.patch_w = w16(self.patch_embed.proj.weight.reshape(D, -1))
        W.patch_b = f32(self.patch_embed.proj.bias)
        W.cls_token = f32(self.cls_token.reshape(-1))
        W.pos_embed = f32(self.pos_embed.reshape(-1, D))
        W.norm_g, W.norm_b = f32(self.norm.weight), f32(self.norm.bias)
        cpad = self._classes_padded
        W.head_w = w16(self.head.weight if cpad == self.num_classes else _pad_rows(self.head.weight, cpad))
        W.head_b = f32(self.head.bias if cpad == self.num_classes else _pad_vec(self.head.bias, cpad))
        tblocks = []
        for i, blk in enumerate(self.blocks):
            b = W.blocks[i]
            b.ln1_g, b.ln1_b = f32(blk.norm1.weight), f32(blk.norm1.bias)
            b.qkv_w, b.qkv_b = w16(blk.attn.qkv.weight, True), f32(blk.attn.qkv.bias)
            tq = tslots.get(state["i"] - 1)
            b.proj_w, b.proj_b = w16(blk.attn.proj.weight, True), f32(blk.attn.proj.bias)
            tp_ = tslots.get(state["i"] - 1)
            b.ln2_g, b.ln2_b = f32(blk.norm2.weight), f32(blk.norm2.bias)
            b.fc1_w, b.fc1_b = w16(blk.mlp.fc1.weight, True), f32(blk.mlp.fc1.bias)
            t1 = tslots.get(state["i"] - 1)
            b.fc2_w, b.fc2_b = w16(blk.mlp.fc2.weight, True), f32(blk.mlp.fc2.bias)
            t2 = tslots.get(state["i"] - 1)
            tblocks.append((tq, tp_, t1, t2))
            if wdt == torch.bfloat16 and mlp_pk_bytes:
                # the fragment-major copy of the two Mlp matrices the fused eval Mlp kernel streams (csrc/tr_mlp_fused.hip), refreshed below /
                # by _refresh_mlp_pack() whenever the bf16 copies were rewritten
                state["i"] += 1
                pkb = slot("p", (mlp_pk_bytes,), torch.uint8)
                b.mlp_pk = pkb.data_ptr()
                mlp_items.append((b.fc1_w, b.fc2_w, b.fc2_b, pkb))
        self._pack_stages(W, w16, f32, keep_alive)
        # may an optimizer that rewrites the fused table's copies itself declare the operands fresh?  Only if that table is all there is:
        # no copied operand, and no reduction module whose transposed matrices training.TrainState keys on the pack generation
        self._pack_all_fused = (not state["copied"] and wdt == torch.bfloat16
                                and type(self)._transposed_stage_weights is VisionTransformer._transposed_stage_weights)
        if fused:
            self._run_fused_pack(fused, dev, state["moved"])
        self._mlp_pack_items = (mlp_items, D, Hd)
        if self.training:
            # a repack from the training forward (every step of a non-fused optimizer, every accumulation micro-step): the train executor
            # never reads the fragment-major Mlp copies -- 12 launches and ~56 MB per step for nothing; the next eval forward refreshes them
            self._mlp_pk_stale = bool(mlp_items)
        else:
            self._refresh_mlp_pack(dev)
        cfg = _lib.TrVitConfig()
        cfg.family = self._family
        cfg.img_size, cfg.patch = self.patch_embed.img_size[0], self.patch_embed.patch_size[0]
        cfg.in_chans = self.patch_embed.proj.in_channels
        cfg.embed_dim, cfg.depth, cfg.num_heads = D, self.depth, self.num_heads
        cfg.mlp_hidden = self.blocks[0].mlp.fc1.out_features
        cfg.num_classes = self._classes_padded
        cfg.ln_eps = float(self.norm.eps)
        cfg.precision = {"bf16": _lib.TR_PREC_BF16, "fp32": _lib.TR_PREC_FP32, "bf16x3": _lib.TR_PREC_BF16X3}[self.precision]
        cfg.knn_k = int(getattr(self, "k_neighbors", 0))
        cfg.cluster_iters = int(getattr(self, "sinkhorn_iters", 0))
        cfg.sinkhorn_eps = float(getattr(self, "sinkhorn_eps", 0.0))
        for i in range(self.depth):
            cfg.keep[i] = int(self._keep[i])
        old = self._packed
        gen = 1 if old is None else old.get("gen", 0) + 1
        self._packed = dict(key=key, W=W, cfg=cfg, keep_alive=keep_alive, gen=gen, tblocks=tblocks if want_t else None)
        self._weights_dirty = False
        # workspaces and captured graphs hold the operand addresses: they survive a repack unless a buffer had to be (re)allocated or the
        # configuration changed (first pack, precision switch, new head, new keep schedule)
        # ... or a LIVE pointer moved: biases, LayerNorm parameters, pos_embed / cls_token (and every weight in fp32 mode) are read through
        # the parameter's own storage, so `p.data = ...`, load_state_dict(assign=True) or an optimizer that flattens its parameters puts
        # a new address into W while every slot stays where it was -- a captured graph would keep reading the old (possibly freed) memory
        if state["moved"] or old is None or bytes(old["cfg"]) != bytes(cfg) or bytes(old["W"]) != bytes(W):
            self._ws = {}
        return self._packed

    def _refresh_mlp_pack(self, dev=None):
        """Rewrite the fragment-major Mlp copies from the (current) bf16 operand copies: one small launch per block, in place -- after every
        _pack() refresh and, on the first eval forward after an optim.FusedAdamW step (which rewrites the bf16 copies itself and marks these
        stale), from forward()."""
        items, D, Hd = self.__dict__.get("_mlp_pack_items") or ([], 0, 0)
        self._mlp_pk_stale = False
        if not items:
            return
        lib = _lib.load()
        dev = dev or items[0][3].device
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            for w1, w2, b2, pkb in items:
                _lib.check(lib.tr_mlp_pack_bf16(w1, w2, b2, pkb.data_ptr(), D, Hd, st), "tr_mlp_pack_bf16")

    def _run_fused_pack(self, fused, dev, moved):
        """All large matrices through ONE tr_cast_pack_bf16 launch; the item table lives on the device and is rebuilt only when an
        address changed."""
        tab = self.__dict__.get("_pack_table")
        sig = tuple((t.data_ptr(), c.data_ptr(), 0 if ct is None else ct.data_ptr(), t.shape[0], t.shape[1]) for t, c, ct in fused)
        if tab is None or tab["sig"] != sig:
            items = np.zeros(len(fused), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("dst_t", "<u8"), ("rows", "<i4"), ("cols", "<i4")]))
            first = np.zeros(len(fused) + 1, dtype=np.int32)
            for n, (sp, dp, tp_, r, c) in enumerate(sig):
                items[n] = (sp, dp, tp_, r, c)
                first[n + 1] = first[n] + ((r + 63) // 64) * ((c + 63) // 64)
            tab = self._pack_table = dict(sig=sig, items=torch.from_numpy(items.view(np.uint8).copy()).to(dev),
                                          first=torch.from_numpy(first).to(dev), n=len(fused), tiles=int(first[-1]))
        with torch.cuda.device(dev):
            _lib.check(_lib.load().tr_cast_pack_bf16(tab["items"].data_ptr(), tab["first"].data_ptr(), tab["n"], tab["tiles"],
                                                     torch.cuda.current_stream().cuda_stream), "tr_cast_pack_bf16")

    def _pack_stages(self, W, w16, f32, keep_alive):
        """Families with learned reduction modules fill W.stage[blk] (tr_stage_weights) here."""

    def _noise_ptr(self, B, dev):
        """Device pointer of the per-stage random inputs (DPC-KNN density noise), None for deterministic families."""
        return None

    def _grad_stage_ptrs(self, G, ptr):
        """Families with learned reduction modules point G.stage[blk] (tr_stage_weights layout) at their gradient views."""

    def _grad_slot_numel(self, name, p):
        """fp32 elements reserved for parameter `name` in the flat gradient buffer (>= p.numel(): matrices whose rows the kernels pad)."""
        if name == "head.weight":
            return self._classes_padded * p.shape[1]
        if name == "head.bias":
            return self._classes_padded
        return p.numel()

    def _pre_pack(self):
        """Parameter maintenance the reference does inside forward() (Sinkhorn re-normalises its centres in place)."""

    def _per_forward_config(self, cfg):
        """Host-side random draws of a forward that the executor takes as inputs (K-Medoids equal_weight)."""

    def _transposed_stage_weights(self, WT, t16):
        """Families whose reduction modules have a backward: transposed bf16 copies of their matrices (dgrad operands)."""

    def _soft_elems(self, B):
        """fp32 elements of the soft-assignment output (SiT), 0 for families without one."""
        return 0

    def _workspace(self, B, dev, slot=0):
        """The executor's buffers for batch size B.  slot > 0: a further, independent set (its own workspace, outputs and captured graphs) for
        a forward that is in flight beside another one (forward_async)."""
        wkey = B if slot == 0 else (B, slot)
        ws = self._ws.get(wkey)
        if ws is None:
            pk = self._packed
            nbytes = _lib.load().tr_vit_workspace_bytes(C.byref(pk["cfg"]), B)
            if nbytes == 0:
                raise RuntimeError("tr_vit_workspace_bytes rejected the model configuration (dims must be multiples of 64, "
                                   "classes of 4, head_dim 64)")
            P = self.patch_embed.num_patches
            ws = dict(buf=torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes=nbytes,
                      kept=torch.empty(self.depth * B * (P + 1), dtype=torch.int32, device=dev),
                      compl=torch.empty(self.depth * B * (P + 1), dtype=torch.int32, device=dev), soft=None)
            if self.viz_mode and self._soft_elems(B):
                ws["soft"] = torch.empty(self._soft_elems(B), dtype=torch.float32, device=dev)
            # keep one batch size resident (all of its slots)
            self._ws = {k: v for k, v in self._ws.items() if (k[0] if isinstance(k, tuple) else k) == B}
            self._ws[wkey] = ws
        return ws

    # ---- training state (flat gradient buffer, tape, workspaces): training.py -----------------------
    def _train_state(self):
        st = getattr(self, "_tstate", None)
        if st is None or st.flat.device != self.pos_embed.device or len(st.order) != len(list(self.parameters())):
            from . import training
            st = self._tstate = training.TrainState(self)
        return st

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        if self.training:
            # engine.py:50-51 `output = model(samples)` in train mode: logits with the HIP backward behind them (training.py)
            from . import training
            return training.train_forward(self, x)
        return self._forward_eval(x, 0)

    def forward_async(self, x: torch.Tensor):
        """Eval forward that may run BESIDE the previous one: the call enqueues the forward on one of `pipeline_depth` (default 2) side streams
        -- each with its own workspace and captured hipGraph -- behind everything already enqueued on the caller's current stream, and
        returns a handle at once; `handle.result()` makes the caller's current stream wait for that forward and returns what `model(x)`
        would.  Forwards on different side streams are independent launches sequences, so the device fills the tail of one forward's launches
        (partial last rounds of the persistent GEMM-class kernels, the short launches of the last stage) with the other's: the headline
        forward runs 2.65 -> 2.44 ms per batch of 256 with two in flight (tools/lab/two_stream_full.py).  A data loop uses it with one batch
        of lookahead (harness.evaluate_multiclass does): launch batch k + 1, then consume batch k.  Same kernels, same bits as `model(x)`.
        Train mode, viz_mode and ATS's dynamic width run synchronously (the handle is already complete)."""
        if self.training or self.viz_mode or getattr(self, "dynamic_width", False) or not x.is_cuda:
            return _Pending(self(x), None)
        depth = max(1, int(getattr(self, "pipeline_depth", 2)))
        st = self.__dict__.setdefault("_pipe_streams", {})
        key = x.device
        if key not in st or len(st[key]) != depth:
            st[key] = [torch.cuda.Stream(device=x.device) for _ in range(depth)]
        slot = self.__dict__.get("_pipe_next", 0) % depth
        self._pipe_next = slot + 1
        side = st[key][slot]
        cur = torch.cuda.current_stream(x.device)
        self._pack()                                   # (re)pack on the caller's stream, where the optimizer / loader wrote
        if getattr(self, "_mlp_pk_stale", False):
            self._refresh_mlp_pack(x.device)
        side.wait_stream(cur)                          # inputs and weights are ready where the forward runs
        with torch.cuda.stream(side):
            out = self._forward_eval(x, slot + 1)
            done = torch.cuda.Event()
            done.record(side)
        x.record_stream(side)
        return _Pending(out, done)

    def _forward_eval(self, x: torch.Tensor, slot: int):
        if not x.is_cuda:
            raise RuntimeError(f"input is on {x.device}: tokenreduction_amd has no CPU path (HIP kernels only)")
        lib = _lib.load()
        pk = self._pack()
        if getattr(self, "_mlp_pk_stale", False):
            self._refresh_mlp_pack(x.device)
        cfg = pk["cfg"]
        B, Cc, Hh, Ww = x.shape
        if (Cc, Hh, Ww) != (cfg.in_chans, cfg.img_size, cfg.img_size):
            raise ValueError(f"expected [B,{cfg.in_chans},{cfg.img_size},{cfg.img_size}], got {tuple(x.shape)}")
        x = x.detach().to(torch.float32).contiguous()
        ws = self._workspace(B, x.device, slot)
        if self.viz_mode and self._soft_elems(B) and ws.get("soft") is None:      # viz_mode switched on after the first call
            ws["soft"] = torch.empty(self._soft_elems(B), dtype=torch.float32, device=x.device)
        want_feat = self.viz_mode or getattr(self, "_always_features", False)
        if want_feat and ws.get("feat") is None:
            # viz_data["Features"]: the residual stream after every block (upper bound depth * B * N0 * D fp32)
            n0 = self.patch_embed.num_patches + 1
            ws["feat"] = torch.empty(self.depth * B * n0 * self.embed_dim, dtype=torch.float32, device=x.device)
        self._noise_slot = slot                      # (a static buffer per workspace slot: two forwards in flight must not share one)
        noise_ptr = self._noise_ptr(B, x.device)
        self._kmed_draws = None
        self._per_forward_config(cfg)
        cfg.concurrent = 1 if slot > 0 else 0        # forward_async: other forwards run beside this one (a scheduling hint, same bits)

        def launch(out):
            tokens = (C.c_int * self.depth)()
            rc = lib.tr_vit_forward(C.byref(cfg), C.byref(pk["W"]), x.data_ptr(), out.data_ptr(), ws["buf"].data_ptr(),
                                    ws["nbytes"], ws["kept"].data_ptr(), ws["compl"].data_ptr(),
                                    None if ws.get("soft") is None else ws["soft"].data_ptr(), noise_ptr,
                                    ws["feat"].data_ptr() if want_feat else None, tokens, B,
                                    torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "tr_vit_forward")
            return list(tokens)

        with torch.cuda.device(x.device):
            if (self.use_graph and not ws.get("graph_off") and not torch.cuda.is_current_stream_capturing()
                    and not getattr(self, "dynamic_width", False)):      # (ATS dynamic width: the executor reads a token count back mid-forward)
                # The forward is a fixed sequence of ~90-130 dependent launches with no host decision in between: replay it as one
                # hipGraph (captured once per batch size / input buffer / output set; the workspace and every output are static
                # buffers).  Re-packing the weights or a new batch size drops the workspace and its graphs with it.
                # A capture bakes the input ADDRESS in.  Callers whose batches arrive at a new address every time (a dtype / layout
                # conversion above, a loader without a static buffer, K-Medoids --equal_weight with its per-forward draws) would
                # re-capture on every call -- slower than not using a graph at all: after GRAPH_MISS_LIMIT misses in a row the
                # workspace goes back to plain launches (at batch 256 within 0.5 % of the replay; bench.py ms_per_step_plain_launches).
                key = (x.data_ptr(), bool(want_feat), ws.get("soft") is not None, noise_ptr, self._kmed_draws)
                graphs = ws.setdefault("graphs", {})
                ent = graphs.get(key)
                if ent is None:
                    ws["graph_misses"] = ws.get("graph_misses", 0) + 1
                    if graphs and ws["graph_misses"] >= self.GRAPH_MISS_LIMIT:
                        ws["graph_off"] = True
                        graphs.clear()
                        warnings.warn(f"{type(self).__name__}: {self.GRAPH_MISS_LIMIT} forwards in a row came with a new input address (or new "
                                      "per-forward draws); hipGraph replay is off for this batch size -- keep the input in one static "
                                      "buffer to get it back (model.use_graph = False silences this)", RuntimeWarning, stacklevel=3)
                        logits = torch.empty(B, self._classes_padded, dtype=torch.float32, device=x.device)
                        tokens = launch(logits)
                        if self._classes_padded != self.num_classes:
                            logits = logits[:, :self.num_classes].contiguous()
                        ent = False
                    else:
                        out = torch.empty(B, self._classes_padded, dtype=torch.float32, device=x.device)
                        if not ws.get("warm"):
                            launch(out)                                   # eager once per workspace: first touch, lazy module load
                            ws["warm"] = True
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g):
                            toks = launch(out)
                        if len(graphs) >= self.GRAPH_CACHE:
                            graphs.pop(next(iter(graphs)))
                        ent = graphs[key] = (g, out, toks)
                else:
                    ws["graph_misses"] = 0
                if ent:
                    g, out, toks = ent
                    g.replay()
                    logits, tokens = out[:, :self.num_classes].clone(), toks
            else:
                logits = torch.empty(B, self._classes_padded, dtype=torch.float32, device=x.device)
                tokens = launch(logits)
                if self._classes_padded != self.num_classes:
                    logits = logits[:, :self.num_classes].contiguous()
        cfg.concurrent = 0                           # (the packed configuration is compared byte-wise on a repack: leave no per-call state in it)
        self._last_tokens = list(tokens)
        self._last_ws = ws
        if self.viz_mode:
            viz = self._viz_data(ws, B, list(tokens))
            viz["Features"] = self._features(ws, B, list(tokens))
            return logits, viz
        return logits

    def check_status(self):
        """Status check of the eval forwards since the last call (tr_vit_forward_status): waits for the stream and raises RuntimeError if
        the device recorded a failure that no launch status can show -- the fused Mlp's stream-K hand-over poll running out (the logits
        of that forward are invalid).  The forward itself never synchronises; call this where the logits are consumed
        (harness.evaluate_multiclass / validate and bench.py do)."""
        if self._packed is None:
            return
        for wkey, ws in list(self._ws.items()):
            if "buf" not in ws:
                continue
            B = wkey[0] if isinstance(wkey, tuple) else wkey
            with torch.cuda.device(ws["buf"].device):
                torch.cuda.synchronize()                   # every stream a forward may have run on (forward_async's side streams)
                _lib.check(_lib.load().tr_vit_forward_status(C.byref(self._packed["cfg"]), ws["buf"].data_ptr(), ws["nbytes"], B,
                                                             torch.cuda.current_stream().cuda_stream), "tr_vit_forward_status")

    def _viz_data(self, ws, B, tokens):
        return {}

    _features_every_block = True

    def _feature_blocks(self, tokens):
        """Blocks whose output the reference records in viz_data["Features"]: every block (deit_viz.py:199, sit.py:130), or --
        families that reduce inside a block -- the blocks that reduced plus the last one (topk.py:195-200 and alike)."""
        if self._features_every_block:
            return list(range(self.depth))
        n_in, out = self.patch_embed.num_patches + 1, {self.depth - 1}
        for blk, n in enumerate(tokens):
            k = self._keep[blk]
            if self._family in (_lib.TR_FAMILY_TOPK, _lib.TR_FAMILY_EVIT):
                reduced = k > 0 and k != n_in - 1                      # topk.py:57: left_tokens == N-1 returns idx None
            elif self._family == _lib.TR_FAMILY_TOME:
                reduced = min(k, (n_in - 1) // 2) > 0
            else:
                reduced = k > 0
            if reduced:
                out.add(blk)
            n_in = n
        return sorted(out)

    def _features(self, ws, B, tokens):
        D = self.embed_dim
        want = set(self._feature_blocks(tokens))
        feats, off = {}, 0
        for blk in range(self.depth):
            n = B * tokens[blk] * D
            if blk in want:
                feats[blk] = ws["feat"][off: off + n].reshape(B, tokens[blk], D).cpu().numpy()
            off += n
        return feats

    def _stage_indices(self, ws, B, tokens):
        """Per reduction block: (blk, N_in, K) from the static per-stage shapes."""
        P = self.patch_embed.num_patches
        out = []
        n_in = P + 1
        for i in range(self.depth):
            K = self._keep[i]
            if K and K != n_in - 1:
                out.append((i, n_in, K))
            n_in = tokens[i]
        return out


class _TopKBase(VisionTransformer):
    """Shared ctor logic of topk.py:108-171 / evit.py:138-201."""
    _blocks_last = True

    _features_every_block = False

    def __init__(self, *a, args=None, dyvit_distillation=False, **kw):
        super().__init__(*a, args=args, **kw)
        token_ratio = list(args.keep_rate)
        pruning_loc = list(args.reduction_loc)
        if len(token_ratio) == 1:
            token_ratio = [token_ratio[0] ** (idx + 1) for idx in range(len(pruning_loc))]   # topk.py:141-142
        assert len(token_ratio) == len(pruning_loc), \
            f"Mismatch between the pruning location ({pruning_loc}) and token ratios ({token_ratio})"
        self.num_patches = self.patch_embed.num_patches
        self.deit_distillation = False
        self.pruning_loc = pruning_loc
        self.token_ratio = token_ratio
        for r, loc in zip(token_ratio, pruning_loc):
            assert 0 < r <= 1, "keep_rate must > 0 and <= 1, got {0}".format(r)   # topk.py:39
            self.blocks[loc].attn.keep_rate = r
            if r < 1:
                self._set_keep(loc, int(r * 196), f"int({r:.4g} * 196) (topk.py:56, 196 hard-coded)")
            else:
                self._keep[loc] = 0
        self._check_static_shapes()

    def _check_static_shapes(self):
        n = self.num_patches + 1
        for i in range(self.depth):
            K = self._keep[i]
            if K and K != n - 1:
                if not 1 <= K < n - 1:
                    raise ValueError(f"block {i}: cannot keep {K} of {n - 1} patch tokens")
                n = self._tokens_after(K)

    def get_reduction_count(self):
        return self.pruning_loc


class TopKVisionTransformer(_TopKBase):
    """models/topk.py:102-212."""
    _family = _lib.TR_FAMILY_TOPK

    def _tokens_after(self, K):
        return K + 1

    def _viz_data(self, ws, B, tokens):
        P = self.patch_embed.num_patches
        kept = ws["kept"].cpu().numpy()
        decisions = {}
        for blk, n_in, K in self._stage_indices(ws, B, tokens):
            decisions[blk] = kept[blk * B * (P + 1): blk * B * (P + 1) + B * K].reshape(B, K).astype(np.int64)
        return {"Kept_Tokens": decisions, "Features": {}}


class EfficientVisionTransformer(_TopKBase):
    """models/evit.py:132-244."""
    _family = _lib.TR_FAMILY_EVIT

    def _tokens_after(self, K):
        return K + 2

    def _viz_data(self, ws, B, tokens):
        P = self.patch_embed.num_patches
        kept = ws["kept"].cpu().numpy()
        compl = ws["compl"].cpu().numpy()
        decisions, fusion = {}, {}
        for blk, n_in, K in self._stage_indices(ws, B, tokens):
            idx = kept[blk * B * (P + 1): blk * B * (P + 1) + B * K].reshape(B, K).astype(np.int64)
            decisions[blk] = np.concatenate([idx, -np.ones((B, 1), dtype=np.int64)], axis=1)   # evit.py:123
            nc = n_in - 1 - K
            fusion[blk] = compl[blk * B * (P + 1): blk * B * (P + 1) + B * nc].reshape(B, nc).astype(np.int64)
        return {"Kept_Tokens": decisions, "Fusion_Assign": fusion, "Features": {}}


class ToMeVisionTransformer(VisionTransformer):
    """models/tome.py:107-223: bipartite soft matching + size-weighted merge between attention and MLP, proportional attention."""
    _blocks_last = True
    _family = _lib.TR_FAMILY_TOME

    _features_every_block = False

    def __init__(self, *a, args=None, **kw):
        super().__init__(*a, args=args, **kw)
        token_ratio = list(args.keep_rate)
        pruning_loc = list(args.reduction_loc)
        P0 = self.patch_embed.num_patches
        if len(token_ratio) == 1:
            token_ratio = [int(P0 * token_ratio[0] ** (idx + 1)) for idx in range(len(pruning_loc))]      # tome.py:145-146
        assert len(token_ratio) == len(pruning_loc), \
            f"Mismatch between the pruning location ({pruning_loc}) and token ratios ({token_ratio})"
        # explicit values are ABSOLUTE patch-token counts; the CLI delivers floats, which crash the reference's slicing
        #  -- cast to int here
        token_ratio = [int(t) for t in token_ratio]
        prev = P0
        for t, loc in zip(token_ratio, pruning_loc):                                                        # tome.py:152-155
            self._keep[loc] = prev - t
            self.blocks[loc].r = prev - t
            prev = t
        if any(k < 0 for k in self._keep):
            raise ValueError(f"ToMe token counts must be non-increasing, got {token_ratio}")
        self.num_patches = P0
        self.deit_distillation = False
        self.pruning_loc = pruning_loc
        self.token_ratio = token_ratio
        self.prop_attn = True

    def get_reduction_count(self):
        return self.pruning_loc

    def _viz_data(self, ws, B, tokens):
        """Assignment_Maps[blk] (tome.py:91-99): for every input patch token, the index of its output token minus 1 -- derived
        from the device's (unm, src, dst) instead of pushing a B*N*N identity through the merge like the reference does."""
        N0 = self.patch_embed.num_patches + 1
        slab = ws["kept"].cpu().numpy()
        maps = {}
        n_in = N0
        for blk in range(self.depth):
            r = min(self._keep[blk], (n_in - 1) // 2)
            if r > 0 and blk in self.pruning_loc:
                na, nb = (n_in + 1) // 2, n_in // 2
                base = blk * B * N0
                unm = slab[base: base + B * (na - r)].reshape(B, na - r).astype(np.int64)
                src = slab[base + B * (na - r): base + B * na].reshape(B, r).astype(np.int64)
                dst = slab[base + B * na: base + B * (na + r)].reshape(B, r).astype(np.int64)
                pos_a = np.zeros((B, na), dtype=np.int64)
                np.put_along_axis(pos_a, unm, np.broadcast_to(np.arange(na - r), (B, na - r)), axis=1)
                np.put_along_axis(pos_a, src, (na - r) + dst, axis=1)
                out = np.empty((B, n_in), dtype=np.int64)
                out[:, 0::2] = pos_a
                out[:, 1::2] = (na - r) + np.arange(nb)
                maps[blk] = (out - 1)[:, 1:]
            n_in = tokens[blk]
        return {"Assignment_Maps": maps, "Features": {}}


def _pad_rows(w, rows):
    """[r, c] -> [rows, c] with zero rows appended (detached fp32 copy)."""
    out = torch.zeros(rows, w.shape[1], dtype=torch.float32, device=w.device)
    out[:w.shape[0]] = w.detach()
    return out


def _pad_cols(w, cols):
    out = torch.zeros(w.shape[0], cols, dtype=torch.float32, device=w.device)
    out[:, :w.shape[1]] = w.detach()
    return out


def _pad_vec(b, n):
    out = torch.zeros(n, dtype=torch.float32, device=b.device)
    out[:b.shape[0]] = b.detach()
    return out


def _ln_default(dim):
    return nn.LayerNorm(dim)          # eps 1e-5: the reduction modules use nn.LayerNorm's default (dyvit.py:97, sit.py:30)


class PredictorLG(nn.Module):
    """Parameter holder with the reference's module tree (dyvit.py:90-110); evaluated by the HIP executor."""

    def __init__(self, embed_dim=384, eps=1e-6):
        super().__init__()
        self.in_conv = nn.Sequential(_ln_default(embed_dim), nn.Linear(embed_dim, embed_dim), nn.GELU())
        self.out_conv = nn.Sequential(nn.Linear(embed_dim, embed_dim // 2), nn.GELU(), nn.Linear(embed_dim // 2, embed_dim // 4),
                                      nn.GELU(), nn.Linear(embed_dim // 4, 2), nn.LogSoftmax(dim=-1))
        self.eps = eps


class DynamicVisionTransformer(VisionTransformer):
    """models/dyvit.py:122-263.  Eval: per pruning block, PredictorLG scores the patch tokens, the best int(P0*ratio) are gathered
    (argsort order) BEFORE the block runs.  Training (dyvit.py:221-229): no token is removed; a straight-through Gumbel-softmax
    sample of the scores becomes the keep policy every later block attends under, and forward returns
    (logits, out_pred_prob) or, with dyvit_distillation, (logits, features, prev_decision, out_pred_prob) (dyvit.py:257-261)."""
    _blocks_last = True
    _family = _lib.TR_FAMILY_DYVIT

    _features_every_block = False

    def __init__(self, *a, args=None, dyvit_distillation=False, **kw):
        distilled = kw.pop("distilled", False)
        super().__init__(*a, args=args, **kw)
        assert (dyvit_distillation & distilled) is False, "Cannot have both DeiT Distillation token and DyViT Distillation scheme"
        if distilled:
            raise NotImplementedError("the distillation token is not built")
        token_ratio = list(args.keep_rate)
        pruning_loc = list(args.reduction_loc)
        if len(token_ratio) == 1:
            token_ratio = [token_ratio[0] ** (idx + 1) for idx in range(len(pruning_loc))]                 # dyvit.py:175-176
        assert len(token_ratio) == len(pruning_loc), \
            f"Mismatch between the pruning location ({pruning_loc}) and token ratios ({token_ratio})"
        self.num_patches = self.patch_embed.num_patches
        self.score_predictor = nn.ModuleList([PredictorLG(self.embed_dim) for _ in range(len(pruning_loc))])
        for m in self.score_predictor.modules():
            _init_vit_weights(m)
        for ratio, loc in zip(token_ratio, pruning_loc):
            self._set_keep(loc, int(self.num_patches * ratio), f"int({self.num_patches} * {ratio:.4g}) (dyvit.py:232)")
        self.deit_distillation = False
        self.dyvit_distillation = dyvit_distillation
        self.pruning_loc = pruning_loc
        self.token_ratio = token_ratio

    def get_new_module_names(self):
        return ["score_predictor"]

    # ---- training (dyvit.py:221-229, 257-261) ------------------------------------------------------------------
    gumbel_noise = None      # tests: {stage: tensor [B,P,2]} = the reference's -log(Exp(1)) draws; None = drawn on the device

    def _gumbel_ptr(self, B, dev):
        P, n_st = self.num_patches, len(self.pruning_loc)
        buf = getattr(self, "_gumbel_buf", None)
        if buf is None or buf.numel() != n_st * B * P * 2 or buf.device != dev:
            buf = self._gumbel_buf = torch.empty(n_st * B * P * 2, dtype=torch.float32, device=dev)
        if self.gumbel_noise is not None:
            buf.copy_(torch.cat([self.gumbel_noise[j].to(device=dev, dtype=torch.float32).reshape(-1) for j in range(n_st)]))
        else:
            buf.exponential_().log_().neg_()          # F.gumbel_softmax: -empty_like(logits).exponential_().log()
        return buf.data_ptr()

    def _grad_stage_ptrs(self, G, ptr):
        for j, loc in enumerate(self.pruning_loc):
            g, pre = G.stage[loc], f"score_predictor.{j}."
            g.ln_g, g.ln_b = ptr(pre + "in_conv.0.weight"), ptr(pre + "in_conv.0.bias")
            g.w0, g.b0 = ptr(pre + "in_conv.1.weight"), ptr(pre + "in_conv.1.bias")
            g.w1, g.b1 = ptr(pre + "out_conv.0.weight"), ptr(pre + "out_conv.0.bias")
            g.w2, g.b2 = ptr(pre + "out_conv.2.weight"), ptr(pre + "out_conv.2.bias")
            g.w3, g.b3 = ptr(pre + "out_conv.4.weight"), ptr(pre + "out_conv.4.bias")

    def _transposed_stage_weights(self, WT, t16):
        hh, qq = (self.embed_dim // 2 + 63) // 64 * 64, (self.embed_dim // 4 + 63) // 64 * 64      # hidden widths as packed (_pack_stages)
        for j, loc in enumerate(self.pruning_loc):
            sp, st = self.score_predictor[j], WT.stage[loc]
            st.w0 = t16(sp.in_conv[1].weight)                                              # [D, D]^T
            st.w1 = t16(_pad_rows(sp.out_conv[0].weight, hh))                              # [Hh, D]^T -> [D, Hh]
            st.w2 = t16(_pad_rows(_pad_cols(sp.out_conv[2].weight, hh), qq))               # [Q, Hh]^T -> [Hh, Q]

    def get_reduction_count(self):
        return self.pruning_loc

    def _pack_stages(self, W, w16, f32, keep_alive):
        for j, loc in enumerate(self.pruning_loc):
            sp, st = self.score_predictor[j], W.stage[loc]
            st.ln_g, st.ln_b = f32(sp.in_conv[0].weight), f32(sp.in_conv[0].bias)
            st.w0, st.b0 = w16(sp.in_conv[1].weight), f32(sp.in_conv[1].bias)
            hh = (self.embed_dim // 2 + 63) // 64 * 64          # D/2 padded with zero weights: K %% 64 for the bf16 GEMM (DeiT-T: 96 -> 128)
            st.w1, st.b1 = w16(_pad_rows(sp.out_conv[0].weight, hh)), f32(_pad_vec(sp.out_conv[0].bias, hh))
            qq = (self.embed_dim // 4 + 63) // 64 * 64          # D/4 rows padded likewise: the training path's GEMMs reduce over them
            st.w2, st.b2 = w16(_pad_rows(_pad_cols(sp.out_conv[2].weight, hh), qq)), f32(_pad_vec(sp.out_conv[2].bias, qq))
            st.h_pad = hh
            st.reserved_ = qq
            st.w3, st.b3 = f32(sp.out_conv[4].weight), f32(sp.out_conv[4].bias)

    def _viz_data(self, ws, B, tokens):
        P1 = self.patch_embed.num_patches + 1
        kept = ws["kept"].cpu().numpy()
        decisions = {}
        for blk in self.pruning_loc:
            K = self._keep[blk]
            decisions[blk] = kept[blk * B * P1: blk * B * P1 + B * K].reshape(B, K).astype(np.int64)
        return {"Kept_Tokens": decisions, "Features": {}}


class TokenSlimmingModule(nn.Module):
    """Parameter holder (sit.py:25-34)."""

    def __init__(self, embed_dim, cluster_centers, ratio=0.5):
        super().__init__()
        hidden_dim = int(embed_dim * ratio)
        self.weight = nn.Sequential(_ln_default(embed_dim), nn.Linear(embed_dim, hidden_dim), nn.GELU(),
                                    nn.Linear(hidden_dim, cluster_centers))
        self.scale = nn.Parameter(torch.ones(1, 1, 1))


class SelfSlimmedVisionTransformer(VisionTransformer):
    """models/sit.py:43-150: a TokenSlimmingModule softly assigns the patch tokens to K outputs before each block in
    reduction_loc."""
    _family = _lib.TR_FAMILY_SIT

    def __init__(self, *a, args=None, **kw):
        super().__init__(*a, args=args, **kw)
        self.cluster_loc = list(args.reduction_loc)
        self.cluster_count = list(args.keep_rate)
        P0 = self.patch_embed.num_patches
        if len(self.cluster_count) == 1:
            self.cluster_count = [int(P0 * (args.keep_rate[0] ** (idx + 1))) for idx in range(len(self.cluster_loc))]   # sit.py:80-81
        assert len(self.cluster_count) == len(self.cluster_loc), \
            f"Mismatch between the cluster location ({self.cluster_loc}) and cluster centers ({self.cluster_count})"
        self.cluster_count = [int(c) for c in self.cluster_count]
        self.cluster_layers = nn.ModuleList([TokenSlimmingModule(self.embed_dim, c) for c in self.cluster_count])
        for m in self.cluster_layers.modules():
            _init_vit_weights(m)
        for c, loc in zip(self.cluster_count, self.cluster_loc):
            self._set_keep(loc, c, "the cluster schedule")

    def get_new_module_names(self):
        return ["cluster_layers"]

    def get_reduction_count(self):
        return self.cluster_loc

    def _pack_stages(self, W, w16, f32, keep_alive):
        for j, loc in enumerate(self.cluster_loc):
            m, st = self.cluster_layers[j], W.stage[loc]
            K = self.cluster_count[j]
            n_pad = (K + 7) // 8 * 8
            w1 = torch.zeros(n_pad, m.weight[3].in_features, dtype=torch.float32, device=m.weight[3].weight.device)
            w1[:K] = m.weight[3].weight.detach()
            b1 = torch.zeros(n_pad, dtype=torch.float32, device=w1.device)
            b1[:K] = m.weight[3].bias.detach()
            st.ln_g, st.ln_b = f32(m.weight[0].weight), f32(m.weight[0].bias)
            hh = (m.weight[1].out_features + 63) // 64 * 64     # hidden width padded with zero weights (DeiT-T: 96 -> 128)
            st.w0, st.b0 = w16(_pad_rows(m.weight[1].weight, hh)), f32(_pad_vec(m.weight[1].bias, hh))
            st.w1, st.b1 = w16(_pad_cols(w1, hh)), f32(b1)
            st.scale = float(m.scale.detach().reshape(-1)[0])
            st.n_pad = n_pad
            st.h_pad = hh

    def _soft_pad(self, K):
        return (K + 7) // 8 * 8, (K + 63) // 64 * 64

    def _grad_slot_numel(self, name, p):
        # the last Linear's rows are padded to a multiple of 8 by the weight-gradient kernel (rows >= K receive zeros)
        for j, K in enumerate(self.cluster_count):
            if name == f"cluster_layers.{j}.weight.3.weight":
                return self._soft_pad(K)[0] * p.shape[1]
            if name == f"cluster_layers.{j}.weight.3.bias":
                return self._soft_pad(K)[0]
        return super()._grad_slot_numel(name, p)

    def _grad_stage_ptrs(self, G, ptr):
        for j, loc in enumerate(self.cluster_loc):
            g, pre = G.stage[loc], f"cluster_layers.{j}."
            g.ln_g, g.ln_b = ptr(pre + "weight.0.weight"), ptr(pre + "weight.0.bias")
            g.w0, g.b0 = ptr(pre + "weight.1.weight"), ptr(pre + "weight.1.bias")
            g.w1, g.b1 = ptr(pre + "weight.3.weight"), ptr(pre + "weight.3.bias")
            g.b2 = ptr(pre + "scale")                                    # d scale (sit.py:34), fp32[1]

    def _transposed_stage_weights(self, WT, t16):
        for j, loc in enumerate(self.cluster_loc):
            m, st = self.cluster_layers[j], WT.stage[loc]
            hh = (m.weight[1].out_features + 63) // 64 * 64                           # hidden width as packed (DeiT-T: 96 -> 128)
            st.w0 = t16(_pad_rows(m.weight[1].weight, hh))                            # [Hh, D]^T
            st.w1 = t16(_pad_rows(_pad_cols(m.weight[3].weight, hh), self._soft_pad(self.cluster_count[j])[1]))   # [K -> ld64, Hh]^T

    def _stage_shapes(self):
        """[(blk, K, P_in)] per slimming stage."""
        out, p_in = [], self.patch_embed.num_patches
        for K, loc in sorted(zip(self.cluster_count, self.cluster_loc), key=lambda t: t[1]):
            out.append((loc, K, p_in))
            p_in = K
        return out

    def _noise_ptr(self, B, dev):
        """Device pointer of the per-stage random inputs (DPC-KNN density noise), None for deterministic families."""
        return None

    def _soft_elems(self, B):
        return sum(B * K * P for _, K, P in self._stage_shapes())

    def _viz_data(self, ws, B, tokens):
        soft = ws["soft"].cpu().numpy()
        assignments, hard = {}, {}
        off = 0
        for blk, K, P in self._stage_shapes():
            a = soft[off: off + B * K * P].reshape(B, K, P)
            off += B * K * P
            assignments[blk] = a
            hard[blk] = np.argmax(a, axis=-2).astype(np.int64)                       # sit.py:122
        return {"Assignment_Maps": hard, "Soft_Assignment_Maps": assignments, "Features": {}}


class CTM(nn.Module):
    """Parameter holder (dpcknn.py:143-151)."""

    def __init__(self, embed_dim, cluster_num, k=5, equal_weight=False):
        super().__init__()
        self.cluster_num, self.equal_weight, self.k = cluster_num, equal_weight, k
        if not self.equal_weight:
            self.score = nn.Linear(embed_dim, 1)


class DPCKNNVisionTransformer(VisionTransformer):
    """models/dpcknn.py:175-290: before each block in reduction_loc the patch tokens are clustered by DPC-KNN and every cluster
    is replaced by the exp(score)-weighted mean of its tokens.

    The reference adds `torch.rand * 1e-6` to the densities (dpcknn.py:71-72).  Here the draws come from torch.rand on the
    device each forward, or from `self.density_noise = {blk: tensor[B,P_in]}` when set (tests feed the reference's draws)."""
    _family = _lib.TR_FAMILY_DPCKNN

    def __init__(self, *a, args=None, **kw):
        super().__init__(*a, args=args, **kw)
        self.cluster_loc = list(args.reduction_loc)
        self.cluster_count = list(args.keep_rate)
        self.k_neighbors = int(args.k_neighbors)
        self.equal_weight = bool(args.equal_weight)
        P0 = self.patch_embed.num_patches
        if len(self.cluster_count) == 1:
            self.cluster_count = [int(P0 * (args.keep_rate[0] ** (idx + 1))) for idx in range(len(self.cluster_loc))]   # dpcknn.py:214-215
        assert len(self.cluster_count) == len(self.cluster_loc), \
            f"Mismatch between the cluster location ({self.cluster_loc}) and cluster centers ({self.cluster_count})"
        self.cluster_count = [int(c) for c in self.cluster_count]
        self.cluster_layers = nn.ModuleList([CTM(self.embed_dim, c, self.k_neighbors, self.equal_weight) for c in self.cluster_count])
        for m in self.cluster_layers.modules():
            _init_vit_weights(m)
        for c, loc in zip(self.cluster_count, self.cluster_loc):
            self._set_keep(loc, c, "the cluster schedule")
        self.density_noise = None
        self._noise_buf = None

    def get_new_module_names(self):
        return ["cluster_layers"]

    def get_reduction_count(self):
        return self.cluster_loc

    def _pack_stages(self, W, w16, f32, keep_alive):
        if self.equal_weight:
            return
        for j, loc in enumerate(self.cluster_loc):
            st = W.stage[loc]
            st.w3, st.b3 = f32(self.cluster_layers[j].score.weight), f32(self.cluster_layers[j].score.bias)

    def _grad_stage_ptrs(self, G, ptr):
        if self.equal_weight:
            return
        for j, loc in enumerate(self.cluster_loc):
            G.stage[loc].w3, G.stage[loc].b3 = ptr(f"cluster_layers.{j}.score.weight"), ptr(f"cluster_layers.{j}.score.bias")

    def _stage_shapes(self):
        out, p_in = [], self.patch_embed.num_patches
        for K, loc in sorted(zip(self.cluster_count, self.cluster_loc), key=lambda t: t[1]):
            out.append((loc, K, p_in))
            p_in = K
        return out

    def _noise_ptr(self, B, dev):
        shapes = self._stage_shapes()
        n = sum(B * P for _, _, P in shapes)
        slot = self.__dict__.get("_noise_slot", 0)                                  # forward_async: one buffer per forward in flight
        buf = self._noise_buf if slot == 0 else (self.__dict__.get("_noise_bufs") or {}).get(slot)
        if buf is None or buf.numel() != n or buf.device != dev:
            buf = torch.empty(n, dtype=torch.float32, device=dev)                    # static: a captured forward reads this address
            if slot == 0:
                self._noise_buf = buf
            else:
                self.__dict__.setdefault("_noise_bufs", {})
                if self._noise_bufs is None:
                    self._noise_bufs = {}
                self._noise_bufs[slot] = buf
        if self.density_noise is not None:
            parts = [self.density_noise[blk].to(device=dev, dtype=torch.float32).reshape(B, P) for blk, _, P in shapes]
            buf.copy_(torch.cat([t.reshape(-1) for t in parts]))
        else:
            buf.uniform_()                                                           # torch.rand: [0, 1)
        return buf.data_ptr()

    def _viz_data(self, ws, B, tokens):
        P1 = self.patch_embed.num_patches + 1
        kept, assign = ws["kept"].cpu().numpy(), ws["compl"].cpu().numpy()
        decisions, assignments = {}, {}
        for blk, K, P in self._stage_shapes():
            decisions[blk] = kept[blk * B * P1: blk * B * P1 + B * K].reshape(B, K).astype(np.int64)
            assignments[blk] = assign[blk * B * P1: blk * B * P1 + B * P].reshape(B, P).astype(np.int64)
        return {"Kept_Tokens": decisions, "Assignment_Maps": assignments, "Center_Feats": {}, "Features": {}}


class ATSVisionTransformer(VisionTransformer):
    """models/ats.py:166-271: blocks in reduction_loc sample tokens by inverse-transform sampling on the CLS attention weighted
    by the value norms.  The reference's token count after such a block is data dependent (batch maximum of unique samples,
    ats.py:78); here it is the static bound K = sample_count, the surplus rows being masked keys (zero attention weight), which
    leaves every valid token and the logits unchanged.  Kept_Tokens is trimmed to the reference's batch-maximum width.

    `model.dynamic_width = True` (eval only, opt-in) makes the executor do what the reference does: after every sampling block it reads the
    batch maximum of unique ids back (one int, a stream synchronisation; no hipGraph replay) and runs the rest of the network on that many
    tokens -- fewer rows in every later block, the same valid tokens (tr_vit_config.ats_dynamic)."""
    _family = _lib.TR_FAMILY_ATS

    _features_every_block = False
    dynamic_width = False

    def _per_forward_config(self, cfg):
        super()._per_forward_config(cfg)
        cfg.ats_dynamic = 1 if (self.dynamic_width and not self.training) else 0

    def __init__(self, *a, args=None, **kw):
        super().__init__(*a, args=args, **kw)
        self.sample_loc = list(args.reduction_loc)
        sample_count = list(args.keep_rate)
        if len(sample_count) == 1:
            sample_count = [int(args.keep_rate[0] ** (idx + 1) * self.patch_embed.num_patches) + 1
                            for idx in range(len(self.sample_loc))]                                        # ats.py:204-205
        assert len(sample_count) == len(self.sample_loc), \
            f"Mismatch between the sample location ({self.sample_loc}) and sample centers ({sample_count})"
        cnt = 0
        self.sample_count = [0] * self.depth
        for idx in range(self.depth):
            if idx in self.sample_loc:
                self.sample_count[idx] = int(sample_count[cnt])
                if self.sample_count[idx] < 1:
                    raise ValueError(f"block {idx}: the sampling schedule keeps {self.sample_count[idx]} tokens -- schedules that leave no patch token are not built")
                cnt += 1
        # static token bound of a sampling block: one token per grid point + CLS.  The grid (ats.py:48) is a float arange with an
        # exclusive end; for 42 sample counts up to 197 (7, 12, 14, ..., 126, ...) rounding admits the end point: K points, bound K + 1
        self._keep = [int(self.sample_steps(k).numel()) + 1 if k else 0 for k in self.sample_count]

    def get_reduction_count(self):
        return self.sample_loc

    @staticmethod
    def sample_steps(sample_count):
        """ats.py:48, verbatim."""
        return torch.arange(1 / (2 * sample_count), (2 * sample_count - 1) / (2 * sample_count), 2 / (2 * sample_count))

    def _pack_stages(self, W, w16, f32, keep_alive):
        for blk, K in enumerate(self.sample_count):
            if K:
                steps = self.sample_steps(K).to(self.pos_embed.device)
                W.stage[blk].w3 = f32(steps)
                W.stage[blk].n_pad = steps.numel()

    def _viz_data(self, ws, B, tokens):
        P1 = self.patch_embed.num_patches + 1
        kept = ws["kept"].cpu().numpy()
        decisions = {}
        for blk, K in enumerate(self._keep):
            if K:
                ids = kept[blk * B * P1: blk * B * P1 + B * K].reshape(B, K).astype(np.int64)
                width = int((ids[:, 1:] != 0).sum(axis=1).max())                      # pad_sequence to the batch maximum, ats.py:78
                decisions[blk] = ids[:, 1:1 + width] - 1                              # ats.py:253
        return {"Kept_Tokens": decisions, "Features": {}}


class Sinkhorn(nn.Module):
    """Parameter holder (sinkhorn.py:59-64): cluster centres v ~ N(0, 1)."""

    def __init__(self, embed_dim, cluster_centers, eps, iters):
        super().__init__()
        self.v = nn.Parameter(torch.randn(cluster_centers, embed_dim))
        self.eps, self.iters = eps, iters


class SinkhornVisionTransformer(SelfSlimmedVisionTransformer):
    """models/sinkhorn.py:89-200: before each block in reduction_loc the unit-norm patch tokens are softly assigned to K
    unit-norm learned centres by `cluster_iters` log-domain Sinkhorn iterations; outputs are assignment-weighted sums of the
    unit-norm tokens.  (The reference re-normalises `v` in place every forward, sinkhorn.py:73-76; here the normalised copy is
    made when the weights are packed and the parameter is left untouched.)"""
    _family = _lib.TR_FAMILY_SINKHORN

    def __init__(self, *a, args=None, **kw):
        VisionTransformer.__init__(self, *a, args=args, **kw)
        self.cluster_loc = list(args.reduction_loc)
        self.cluster_count = list(args.keep_rate)
        self.sinkhorn_eps = float(args.sinkhorn_eps)
        self.sinkhorn_iters = int(args.cluster_iters)
        P0 = self.patch_embed.num_patches
        if len(self.cluster_count) == 1:
            self.cluster_count = [int(P0 * (args.keep_rate[0] ** (idx + 1))) for idx in range(len(self.cluster_loc))]   # sinkhorn.py:128-129
        assert len(self.cluster_count) == len(self.cluster_loc), \
            f"Mismatch between the cluster location ({self.cluster_loc}) and cluster centers ({self.cluster_count})"
        self.cluster_count = [int(c) for c in self.cluster_count]
        self.cluster_layers = nn.ModuleList([Sinkhorn(self.embed_dim, c, self.sinkhorn_eps, self.sinkhorn_iters)
                                             for c in self.cluster_count])
        for c, loc in zip(self.cluster_count, self.cluster_loc):
            self._set_keep(loc, c, "the cluster schedule")

    def _pre_pack(self):
        if self.training:                       # sinkhorn.py:72-76: the centres are re-normalised IN PLACE (no grad) at every forward
            with torch.no_grad():
                for m in self.cluster_layers:
                    m.v.copy_(torch.nn.functional.normalize(m.v, p=2, dim=-1))

    def _grad_slot_numel(self, name, p):
        for j, K in enumerate(self.cluster_count):
            if name == f"cluster_layers.{j}.v":
                return self._soft_pad(K)[0] * p.shape[1]
        return VisionTransformer._grad_slot_numel(self, name, p)

    def _grad_stage_ptrs(self, G, ptr):
        for j, loc in enumerate(self.cluster_loc):
            G.stage[loc].w1 = ptr(f"cluster_layers.{j}.v")               # the gradient reaches v as if it were the unit vector (sinkhorn.py:76-77)

    def _transposed_stage_weights(self, WT, t16):
        for j, loc in enumerate(self.cluster_loc):
            v = torch.nn.functional.normalize(self.cluster_layers[j].v.detach().float(), p=2, dim=-1)
            WT.stage[loc].w1 = t16(_pad_rows(v, self._soft_pad(self.cluster_count[j])[1]))

    def _pack_stages(self, W, w16, f32, keep_alive):
        for j, loc in enumerate(self.cluster_loc):
            v, st = self.cluster_layers[j].v.detach(), W.stage[loc]
            K = self.cluster_count[j]
            n_pad = (K + 7) // 8 * 8
            w1 = torch.zeros(n_pad, v.shape[1], dtype=torch.float32, device=v.device)
            w1[:K] = torch.nn.functional.normalize(v.float(), p=2, dim=-1)
            st.w1, st.b1 = w16(w1), f32(torch.zeros(n_pad, dtype=torch.float32, device=v.device))
            st.n_pad = n_pad

    def _viz_data(self, ws, B, tokens):
        out = super()._viz_data(ws, B, tokens)
        out["Center_Feats"] = {}
        return out


class KMedoids(nn.Module):
    """kmedoids.py:135-149 (no parameters)."""

    def __init__(self, num_clusters, iters, equal_weights=False):
        super().__init__()
        self.cluster_count, self.iters, self.equal_weights = num_clusters, iters, equal_weights


class KMedoidsVisionTransformer(VisionTransformer):
    """models/kmedoids.py:152-272: before each block in reduction_loc the patch tokens are clustered by weighted K-Medoids
    (weights = column sums of the previous block's attention) and replaced by the medoid tokens themselves."""
    _blocks_last = True
    _family = _lib.TR_FAMILY_KMEDOIDS

    def __init__(self, *a, args=None, **kw):
        super().__init__(*a, args=args, **kw)
        self.num_patches = self.patch_embed.num_patches
        self.cluster_loc = list(args.reduction_loc)
        self.cluster_count = list(args.keep_rate)
        self.sinkhorn_iters = self.cluster_iters = int(args.cluster_iters)      # cfg.cluster_iters carries it to the executor
        self.equal_weight = bool(args.equal_weight)
        if len(self.cluster_count) == 1:
            self.cluster_count = [int(self.num_patches * (args.keep_rate[0] ** (idx + 1))) for idx in range(len(self.cluster_loc))]
        assert len(self.cluster_count) == len(self.cluster_loc), \
            f"Mismatch between the cluster location ({self.cluster_loc}) and cluster centers ({self.cluster_count})"
        self.cluster_count = [int(c) for c in self.cluster_count]
        if 0 in self.cluster_loc:
            raise ValueError("kmedoids cannot reduce before block 0: there is no previous attention (kmedoids.py:240)")
        self.cluster_layers = nn.ModuleList([KMedoids(c, self.cluster_iters, self.equal_weight) for c in self.cluster_count])
        for c, loc in zip(self.cluster_count, self.cluster_loc):
            self._set_keep(loc, c, "the cluster schedule")

    def _per_forward_config(self, cfg):
        """args.equal_weight (kmedoids.py:43-47): every k_medoids_fit call draws its first medoid with
        `np.random.choice(np.arange(N), 1)` from numpy's GLOBAL generator -- drawn here, stage by stage in forward order, so a
        seeded run consumes the stream exactly like the reference; the executor takes the ids as inputs (and a captured graph is
        keyed on them through the noise key below)."""
        if not self.equal_weight:
            return
        p_in, draws = self.num_patches, []
        for K, loc in sorted(zip(self.cluster_count, self.cluster_loc), key=lambda t: t[1]):
            first = int(np.random.choice(np.arange(p_in), 1)[0])
            cfg.kmed_init[loc] = first + 1
            draws.append(first)
            p_in = K
        self._kmed_draws = tuple(draws)

    def _noise_ptr(self, B, dev):
        return None

    def get_new_module_names(self):
        return ["cluster_layers"]

    def get_reduction_count(self):
        return self.cluster_loc

    _stage_shapes = DPCKNNVisionTransformer._stage_shapes

    def _viz_data(self, ws, B, tokens):
        P1 = self.patch_embed.num_patches + 1
        kept, assign = ws["kept"].cpu().numpy(), ws["compl"].cpu().numpy()
        decisions, assignments = {}, {}
        for blk, K, P in self._stage_shapes():
            decisions[blk] = kept[blk * B * P1: blk * B * P1 + B * K].reshape(B, K).astype(np.int64)
            assignments[blk] = assign[blk * B * P1: blk * B * P1 + B * P].reshape(B, P).astype(np.int64)
        return {"Kept_Tokens": decisions, "Assignment_Maps": assignments, "Center_Feats": {}, "Features": {}}


class PatchMerger(nn.Module):
    """Parameter holder (patchmerger.py:24-33)."""

    def __init__(self, embed_dim, cluster_centers, scaled_attention=False):
        super().__init__()
        self.scale = embed_dim ** -0.5 if scaled_attention else 1.
        self.norm = _ln_default(embed_dim)
        self.queries = nn.Parameter(torch.randn(cluster_centers, embed_dim))


class PatchMergerVisionTransformer(SelfSlimmedVisionTransformer):
    """models/patchmerger.py:42-150: before each block in reduction_loc, K learned queries attend over the LayerNorm-ed patch
    tokens (softmax over the tokens) and the outputs are the attention-weighted sums of the normalised tokens."""
    _family = _lib.TR_FAMILY_PATCHMERGER

    def __init__(self, *a, args=None, **kw):
        VisionTransformer.__init__(self, *a, args=args, **kw)
        self.cluster_loc = list(args.reduction_loc)
        self.cluster_count = list(args.keep_rate)
        P0 = self.patch_embed.num_patches
        if len(self.cluster_count) == 1:
            self.cluster_count = [int(P0 * (args.keep_rate[0] ** (idx + 1))) for idx in range(len(self.cluster_loc))]   # patchmerger.py:78-79
        assert len(self.cluster_count) == len(self.cluster_loc), \
            f"Mismatch between the cluster location ({self.cluster_loc}) and cluster centers ({self.cluster_count})"
        self.cluster_count = [int(c) for c in self.cluster_count]
        self.cluster_layers = nn.ModuleList([PatchMerger(self.embed_dim, c) for c in self.cluster_count])
        for m in self.cluster_layers.modules():
            _init_vit_weights(m)
        for c, loc in zip(self.cluster_count, self.cluster_loc):
            self._set_keep(loc, c, "the cluster schedule")

    def _grad_slot_numel(self, name, p):
        for j, K in enumerate(self.cluster_count):
            if name == f"cluster_layers.{j}.queries":
                return self._soft_pad(K)[0] * p.shape[1]
        return VisionTransformer._grad_slot_numel(self, name, p)

    def _grad_stage_ptrs(self, G, ptr):
        for j, loc in enumerate(self.cluster_loc):
            g, pre = G.stage[loc], f"cluster_layers.{j}."
            g.ln_g, g.ln_b = ptr(pre + "norm.weight"), ptr(pre + "norm.bias")
            g.w1 = ptr(pre + "queries")

    def _transposed_stage_weights(self, WT, t16):
        for j, loc in enumerate(self.cluster_loc):
            WT.stage[loc].w1 = t16(_pad_rows(self.cluster_layers[j].queries, self._soft_pad(self.cluster_count[j])[1]))

    def _pack_stages(self, W, w16, f32, keep_alive):
        for j, loc in enumerate(self.cluster_loc):
            m, st = self.cluster_layers[j], W.stage[loc]
            K = self.cluster_count[j]
            n_pad = (K + 7) // 8 * 8
            w1 = torch.zeros(n_pad, self.embed_dim, dtype=torch.float32, device=m.queries.device)
            w1[:K] = m.queries.detach()
            st.ln_g, st.ln_b = f32(m.norm.weight), f32(m.norm.bias)
            st.w1, st.b1 = w16(w1), f32(torch.zeros(n_pad, dtype=torch.float32, device=w1.device))
            st.scale = float(m.scale)
            st.n_pad = n_pad

    def _viz_data(self, ws, B, tokens):
        out = super()._viz_data(ws, B, tokens)
        out["Center_Feats"] = {}
        return out


class HeuristicVisionTransformer(VisionTransformer):
    """models/heuristic.py:88-277: image-independent spatial pruning.  From every block in the reduction range on, the patch
    tokens farther than that block's radius (L1 / L2 / Linf distance from the grid centre) are masked as attention keys; no
    token is removed.  The radii are constructor-time geometry (prep_pattern / prep_pattern_stage_subset), computed on the
    host exactly as the reference does.  (The reference also masks those tokens as QUERIES, which turns their rows into the
    mean of V; here they attend like everyone else -- rows nothing ever reads: they stay masked as keys to the end.)"""
    _family = _lib.TR_FAMILY_HEURISTIC
    _features_every_block = False

    def __init__(self, *a, args=None, **kw):
        super().__init__(*a, args=args, **kw)
        self.heuristic_pattern = args.heuristic_pattern
        if args.not_contiguous:
            self.reduction_loc = list(args.reduction_loc)
            self.keep_rate = list(args.keep_rate)
            if len(self.keep_rate) != 1:
                raise ValueError("heuristic with --not_contiguous defines its token targets for a single keep_rate only "
                                 "(heuristic.py:127-128)")
            num_tokens = [int(self.patch_embed.num_patches * self.keep_rate[0] ** (idx + 1)) for idx in range(len(self.reduction_loc))]
            self.distances, self.threshold, self.P = self.prep_pattern_stage_subset(num_tokens)
        else:
            self.min_radius = args.min_radius
            self.start_stage = int(min(args.reduction_loc))
            self.end_stage = int(max(args.reduction_loc))
            self.reduction_loc = [idx for idx in range(self.start_stage, self.end_stage + 1)]
            self.distances, self.threshold, self.P = self.prep_pattern()

    # ---- constructor-time geometry: one radius per block, a patch is visible while its distance to the grid centre is within it
    _NORMS = {"l1": lambda u, v: u.abs() + v.abs(), "l2": lambda u, v: (u * u + v * v).sqrt(), "linf": lambda u, v: torch.maximum(u.abs(), v.abs())}

    def _patch_radii(self):
        """[g, g] distance of every patch to the centre of the patch grid, in the norm named by --heuristic_pattern.  The grid
        coordinates are the reference's (g evenly spaced values from floor(-g/2) to floor(g/2), heuristic.py:158-162) so the
        radii -- and with them every mask -- are the same floats."""
        norm = self._NORMS.get(str(self.heuristic_pattern).lower())
        if norm is None:
            raise ValueError(f"heuristic_pattern {self.heuristic_pattern!r}: expected l1 | l2 | linf")
        g = int(self.patch_embed.num_patches ** 0.5)
        axis = torch.linspace((-g) // 2, g // 2, steps=g)
        return norm(axis[:, None].expand(g, g), axis[None, :].expand(g, g)), g

    def prep_pattern(self):
        """Contiguous range (heuristic.py:157-181): the radius shrinks linearly from the corner distance (everything visible) one
        block before the range to `min_radius` one block after it, and stays constant outside."""
        radii, g = self._patch_radii()
        if self.min_radius is None or self.min_radius <= 0:
            self.min_radius = radii[g // 2, g // 2]
        corner = float(radii[0, 0])
        n_stage = self.end_stage - self.start_stage + 1
        ramp = torch.linspace(corner, float(self.min_radius), n_stage + 2)            # ramp[0] = all visible, ramp[-1] = min_radius
        # the ramp starts one block BEFORE the range -- except for a range that starts at block 0, where the reference's left padding is
        # empty (F.pad(..., max(start_stage - 1, 0)), heuristic.py:178) and block 0 itself still sees everything
        pos = (torch.arange(self.depth + 2) - max(self.start_stage - 1, 0)).clamp(0, n_stage + 1)
        return radii, ramp[pos], g

    def prep_pattern_stage_subset(self, num_tokens):
        """Listed blocks only (heuristic.py:184-224): each stage takes, among the distinct radii of the grid, the one whose disc
        covers a patch count closest to its target (the smaller radius on a tie); a block uses the radius of the last stage at
        or before it, and everything is visible before the first one."""
        radii, g = self._patch_radii()
        levels = torch.unique(radii)                                                   # ascending
        covered = (radii.reshape(1, -1) <= levels.reshape(-1, 1)).sum(dim=1)            # patches inside each candidate disc
        targets = torch.as_tensor(list(num_tokens), dtype=covered.dtype)
        pick = (covered.reshape(1, -1) - targets.reshape(-1, 1)).abs().argmin(dim=1)    # first minimum = smallest radius
        table = torch.cat([levels[-1:], levels[pick]])
        is_stage = torch.zeros(self.depth, dtype=torch.long)
        is_stage[torch.as_tensor(sorted(self.reduction_loc), dtype=torch.long)] = 1
        return radii, table[torch.cumsum(is_stage, dim=0)], g

    def get_reduction_count(self):
        return self.reduction_loc

    def _block_mask(self, idx):
        return (self.distances <= self.threshold[idx]).reshape(self.P * self.P)

    def _pack_stages(self, W, w16, f32, keep_alive):
        dev = self.pos_embed.device
        for idx in self.reduction_loc:
            m = torch.cat([torch.ones(1), self._block_mask(idx).float()]).to(dev)
            W.stage[idx].w3 = f32(m)
            W.stage[idx].n_pad = m.numel()

    def _feature_blocks(self, tokens):
        return sorted(set(int(b) for b in self.reduction_loc) | {self.depth - 1})

    def _viz_data(self, ws, B, tokens):
        decisions = {}
        for idx in self.reduction_loc:
            ind = self._block_mask(idx).nonzero(as_tuple=True)[0]
            decisions[idx] = ind.unsqueeze(0).expand(B, -1).numpy().astype(np.int64)
        return {"Kept_Tokens_Abs": decisions}


class VisionTransformerTeacher(VisionTransformer):
    """models/dyvit.py:267-334: the DyViT distillation teacher -- a plain DeiT whose forward returns
    (head(norm(x)[:, 0]), norm(x)[:, 1:]): the logits and the final-norm patch-token features, always as a tuple.  The teacher is
    only ever run under `torch.no_grad()` (losses.py:122-123), so its forward is the inference executor whatever `self.training` says;
    its outputs carry no gradient."""
    _always_features = True

    def forward(self, x):
        viz, self.viz_mode = self.viz_mode, False
        was_training, self.training = self.training, False          # the module flag only: the teacher has no train-mode behaviour
        try:
            logits = super().forward(x)
        finally:
            self.viz_mode = viz
            self.training = was_training
        from . import ops
        B, N, D = x.shape[0], self._last_tokens[-1], self.embed_dim
        off = sum(B * n * D for n in self._last_tokens[:-1])
        x_final = self._last_ws["feat"][off: off + B * N * D]                    # residual stream after the last block
        feature = ops.layernorm_f32(x_final.view(B * N, D), self.norm.weight.detach().float().contiguous(),
                                    self.norm.bias.detach().float().contiguous(), float(self.norm.eps)).view(B, N, D)
        return logits, feature[:, 1:]
