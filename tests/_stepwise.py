"""TEST INFRASTRUCTURE: kernel-by-kernel forward through the per-op C-ABI entry points.

Same launch sequence as the native executor (csrc/tr_vit.hip -> tr_vit_forward), driven from Python so the parity tests can
look at every intermediate (op-boundary checks against the oracle) and assert the executor's result bit for bit.  Not part of
the product and not used for timing: bench.py times the executor itself through the library's launch profiler.
HIP kernels only -- no CPU path.  Results are bit-identical to tr_vit_forward (same kernels, same order).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch

from tokenreduction_amd import _lib, ops
from tokenreduction_amd.models import VisionTransformer, _pad_cols, _pad_rows, _pad_vec


class Trace:
    """Collects (kernel name, flops, bytes, elapsed_ms) per launch when timing is enabled, and named intermediates."""

    def __init__(self, timing: bool = False, keep: bool = False):
        self.timing, self.keep = timing, keep
        self.launches: List[dict] = []
        self.tensors: Dict[str, torch.Tensor] = {}
        self._pending = []
        self._bufs = {}

    def run(self, name: str, flops: float, nbytes: float, fn: Callable):
        if not self.timing:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self._pending.append((name, flops, nbytes, e0, e1))
        return out

    def finish(self):
        torch.cuda.synchronize()
        for name, flops, nbytes, e0, e1 in self._pending:
            self.launches.append(dict(kernel=name, flops=flops, bytes=nbytes, ms=e0.elapsed_time(e1)))
        self._pending = []

    def buf(self, name: str, shape, dtype, device):
        """Reusable output buffer: keeps torch's allocator (and its hipMalloc/hipFree stalls) out of the timed launches."""
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.empty(shape, dtype=dtype, device=device)
        return t

    def save(self, name: str, t: torch.Tensor):
        if self.keep:
            self.tensors[name] = t.clone()


_EPI_NAME = {ops.TR_EPI_BF16: "gemm_bf16_pc<EPI_BF16>", ops.TR_EPI_GELU_BF16: "gemm_bf16_pc<EPI_GELU_BF16>",
             ops.TR_EPI_RESID_F32: "gemm_bf16_persistent<EPI_RESID_F32>", ops.TR_EPI_F32: "gemm_bf16_persistent<EPI_F32>",
             ops.TR_EPI_PATCH_F32: "gemm_bf16_persistent<EPI_PATCH_F32>"}


def _gemm(tr: Trace, a, w, b, epi, out=None, aux=None, aux_i=0, tag=""):
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = tr.buf("gemm" + tag, (M, N), torch.bfloat16 if epi in (ops.TR_EPI_BF16, ops.TR_EPI_GELU_BF16) else torch.float32,
                     a.device)
    obytes = 2 if epi in (ops.TR_EPI_BF16, ops.TR_EPI_GELU_BF16) else (8 if epi == ops.TR_EPI_RESID_F32 else 4)
    nbytes = 2.0 * M * K + 2.0 * N * K + float(obytes) * M * N
    return tr.run(_EPI_NAME[epi], 2.0 * M * N * K, nbytes, lambda: ops.gemm(a, w, b, epi, out=out, aux=aux, aux_i=aux_i))


@torch.no_grad()
def forward_stepwise(model: VisionTransformer, x: torch.Tensor, trace: Optional[Trace] = None, forced_ids: Optional[dict] = None):
    """Returns (logits, info) with info = dict(kept={blk: idx}, compl={blk: compl}, scores={blk: scores}, tokens=[...])."""
    tr = trace or Trace()
    pk = model._pack()
    cfg = pk["cfg"]
    B = x.shape[0]
    D, H, P = cfg.embed_dim, cfg.num_heads, model.patch_embed.num_patches
    N = P + 1
    dev = x.device
    bf = lambda t: t.detach().to(torch.bfloat16).contiguous()
    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    x = f32(x)
    pos = f32(model.pos_embed.reshape(-1, D))
    w16, pb, cls = bf(model.patch_embed.proj.weight.reshape(D, -1)), f32(model.patch_embed.proj.bias), f32(model.cls_token.reshape(-1))
    if _lib.load().tr_patch_embed_supported(x.shape[1], x.shape[2], cfg.patch, D):          # the eval executor's choice (tr_vit.hip)
        h = tr.run("patch_embed_kernel", 2.0 * B * P * D * w16.shape[1], 4.0 * x.numel() + 4.0 * B * N * D,
                   lambda: ops.patch_embed(x, w16, pb, cls, pos, cfg.patch)).view(B * N, D)
    else:
        cols = tr.run("im2col_kernel", 0.0, 4.0 * x.numel() + 2.0 * x.numel(), lambda: ops.im2col(x, cfg.patch))
        h = torch.empty(B * N, D, dtype=torch.float32, device=dev)
        _gemm(tr, cols, w16, pb, ops.TR_EPI_PATCH_F32, out=h, aux=pos, aux_i=P)
        tr.run("cls_pos_kernel", 0.0, 12.0 * B * D, lambda: ops.cls_pos_rows(cls, pos, h, B, N, D))
    tr.save("embed", h.view(B, N, D))
    info = dict(kept={}, compl={}, scores={}, tokens=[])
    eps = float(model.norm.eps)
    fuse = cfg.family == 2
    pending = None     # bf16 residual not yet added to h (previous block's fc2 output)
    tome = cfg.family == 3
    size = None        # ToMe token sizes [B, N] (None until the first merge)
    info["tome"] = {}
    info["soft"] = {}
    colsum = None      # K-Medoids: per-wave column sums of the previous block's attention
    noise_parts = None
    if cfg.family == 6:       # DPC-KNN: the same density noise the executor would use (model.density_noise or fresh draws)
        model._noise_ptr(B, dev)
        noise_parts, off = [], 0
        for _, _, P_in in model._stage_shapes():
            noise_parts.append(model._noise_buf[off: off + B * P_in].reshape(B, P_in))
            off += B * P_in
    for i, blk in enumerate(model.blocks):
        xn = None
        if cfg.family == 6 and int(cfg.keep[i]) > 0:
            # DPC-KNN: cluster + merge BEFORE the block (dpcknn.py:258-262), merge fused with norm1
            Kc, M = int(cfg.keep[i]), B * N
            j = model.cluster_loc.index(i)
            if pending is not None:
                tr.run("layernorm_kernel", 0.0, 12.0 * M * D,
                       lambda: ops.layernorm(h, f32(blk.norm1.weight), f32(blk.norm1.bias), eps, delta=pending))
                pending = None
            noise = None if noise_parts is None else noise_parts[j]
            centers, assign, score = tr.run("dpcknn_cluster", 2.0 * B * (N - 1) * (N - 1) * D, 16.0 * B * (N - 1) * (N - 1),
                                            lambda: ops.dpcknn_cluster(h.view(B, N, D), Kc, noise, model.k_neighbors, fast_dist=True))
            info["kept"][i], info["compl"][i], info["scores"][i] = centers, assign, score
            ctm = model.cluster_layers[j]
            sw = None if model.equal_weight else f32(ctm.score.weight)
            sb = None if model.equal_weight else f32(ctm.score.bias)
            h3_, xn = tr.run("cluster_merge_layernorm_kernel", 0.0, 4.0 * M * D + 6.0 * B * (Kc + 1) * D,
                             lambda: ops.cluster_merge_layernorm(h.view(B, N, D), assign, Kc, f32(blk.norm1.weight),
                                                                 f32(blk.norm1.bias), eps, sw, sb))
            N = Kc + 1
            h, xn = h3_.view(B * N, D), xn.view(B * N, D)
        if cfg.family == 9 and int(cfg.keep[i]) > 0:
            # K-Medoids: medoids of the patch tokens replace them BEFORE the block (kmedoids.py:238-248)
            Kc, M = int(cfg.keep[i]), B * N
            if pending is not None:
                tr.run("layernorm_kernel", 0.0, 12.0 * M * D,
                       lambda: ops.layernorm(h, f32(blk.norm1.weight), f32(blk.norm1.bias), eps, delta=pending))
                pending = None
            if getattr(model, "equal_weight", False):          # the first-medoid draws of the model's last forward (same stage order)
                first = model._kmed_draws[sorted(model.cluster_loc).index(i)]
                centers, assign = tr.run("kmedoids", 2.0 * B * (N - 1) * (N - 1) * D, 8.0 * B * (N - 1) * (N - 1),
                                         lambda: ops.kmedoids_equal(h.view(B, N, D), first, Kc, model.cluster_iters, fast_dist=True))
                info["kept"][i], info["compl"][i], info["scores"][i] = centers, assign, None
            else:
                centers, assign = tr.run("kmedoids", 2.0 * B * (N - 1) * (N - 1) * D, 8.0 * B * (N - 1) * (N - 1),
                                         lambda: ops.kmedoids(h.view(B, N, D), colsum, Kc, model.cluster_iters, fast_dist=True))
                info["kept"][i], info["compl"][i], info["scores"][i] = centers, assign, colsum.sum(dim=(1, 2))[:, 1:]
            h3_, xn = tr.run("gather_layernorm_kernel", 0.0, 10.0 * B * (Kc + 1) * D,
                             lambda: ops.gather_layernorm(h.view(B, N, D), centers, None, None, f32(blk.norm1.weight),
                                                          f32(blk.norm1.bias), eps))
            N = Kc + 1
            h, xn = h3_.view(B * N, D), xn.view(B * N, D)
        if cfg.family == 11 and i in model.reduction_loc:
            # Heuristic: a new spatial key mask from this block on (heuristic.py:247-258)
            m = torch.cat([torch.ones(1), model._block_mask(i).float()]).to(dev)
            size = m.unsqueeze(0).expand(B, -1).contiguous()
        if cfg.family == 10 and int(cfg.keep[i]) > 0:
            # PatchMerger: K learned queries attend over the normalised tokens BEFORE the block (patchmerger.py:35-39)
            Kc, M = int(cfg.keep[i]), B * N
            j = model.cluster_loc.index(i)
            m = model.cluster_layers[j]
            y = tr.run("layernorm_kernel", 0.0, (6.0 if pending is None else 12.0) * M * D,
                       lambda: ops.layernorm(h, f32(m.norm.weight), f32(m.norm.bias), 1e-5, delta=pending))
            pending = None
            xh = tr.run("layernorm_kernel", 0.0, 8.0 * M * D, lambda: ops.layernorm_f32(h, f32(m.norm.weight), f32(m.norm.bias), 1e-5))
            n_pad = (Kc + 7) // 8 * 8
            w1 = torch.zeros(n_pad, D, dtype=torch.float32, device=dev)
            w1[:Kc] = m.queries.detach()
            lg = _gemm(tr, y, bf(w1), torch.zeros(n_pad, dtype=torch.float32, device=dev), ops.TR_EPI_F32, tag="pm")
            h3_, soft = tr.run("sit_merge_kernel", 2.0 * B * Kc * (N - 1) * D, 4.0 * B * (N + Kc) * D,
                               lambda: ops.softassign_merge_fast(lg.view(B, N, n_pad), float(m.scale), h.view(B, N, D), Kc, True, tr.keep,
                                                                 src=xh.view(B, N, D))
                               if Kc <= 192 else ops.sit_merge(lg.view(B, N, n_pad), float(m.scale), h.view(B, N, D), Kc,
                                                               want_soft=tr.keep, src=xh.view(B, N, D)))
            info["soft"][i] = soft
            N = Kc + 1
            h = h3_.view(B * N, D)
        if cfg.family == 8 and int(cfg.keep[i]) > 0:
            # Sinkhorn: soft assignment to unit-norm centres BEFORE the block (sinkhorn.py:66-86)
            Kc, M = int(cfg.keep[i]), B * N
            j = model.cluster_loc.index(i)
            if pending is not None:
                tr.run("layernorm_kernel", 0.0, 12.0 * M * D,
                       lambda: ops.layernorm(h, f32(blk.norm1.weight), f32(blk.norm1.bias), eps, delta=pending))
                pending = None
            xh, xlp = tr.run("rownorm_kernel", 0.0, 10.0 * M * D, lambda: ops.rownorm(h))
            n_pad = (Kc + 7) // 8 * 8
            w1 = torch.zeros(n_pad, D, dtype=torch.float32, device=dev)
            w1[:Kc] = torch.nn.functional.normalize(model.cluster_layers[j].v.detach().float(), p=2, dim=-1)
            sc = _gemm(tr, xlp, bf(w1), torch.zeros(n_pad, dtype=torch.float32, device=dev), ops.TR_EPI_F32, tag="sk")
            wt, soft = tr.run("sinkhorn_kernel", 0.0, 8.0 * M * n_pad,
                              lambda: ops.sinkhorn(sc.view(B, N, n_pad), Kc, model.sinkhorn_eps, model.sinkhorn_iters, want_soft=tr.keep))
            info["soft"][i] = soft
            h3_ = tr.run("sit_merge_kernel", 2.0 * B * Kc * (N - 1) * D, 4.0 * B * (N + Kc) * D,
                         lambda: ops.softassign_merge_fast(wt, 1.0, h.view(B, N, D), Kc, False, src=xh.view(B, N, D))[0]
                         if Kc <= 192 else ops.weighted_merge(wt, h.view(B, N, D), xh.view(B, N, D), Kc))
            N = Kc + 1
            h = h3_.view(B * N, D)
        if cfg.family in (4, 5) and int(cfg.keep[i]) > 0:
            # DyViT / SiT: the reduction module runs on x BEFORE the block (dyvit.py:218-239, sit.py:116-119)
            Kc, M = int(cfg.keep[i]), B * N
            j = (model.pruning_loc if cfg.family == 4 else model.cluster_loc).index(i)
            if cfg.family == 4:
                sp = model.score_predictor[j]
                y = tr.run("layernorm_kernel", 0.0, (6.0 if pending is None else 12.0) * M * D,
                           lambda: ops.layernorm(h, f32(sp.in_conv[0].weight), f32(sp.in_conv[0].bias), 1e-5, delta=pending))
                pending = None
                h1 = _gemm(tr, y, bf(sp.in_conv[1].weight), f32(sp.in_conv[1].bias), ops.TR_EPI_GELU_BF16, tag="p0")
                tr.run("pool_broadcast_kernel", 0.0, 2.0 * M * D, lambda: ops.pool_broadcast(h1, B, N))
                hh = (D // 2 + 63) // 64 * 64       # hidden width padded with zero weights, as models.py packs it
                h2 = _gemm(tr, h1, bf(_pad_rows(sp.out_conv[0].weight, hh)), _pad_vec(sp.out_conv[0].bias, hh), ops.TR_EPI_GELU_BF16,
                           tag="p1")
                h3 = _gemm(tr, h2, bf(_pad_cols(sp.out_conv[2].weight, hh)), f32(sp.out_conv[2].bias), ops.TR_EPI_GELU_BF16, tag="p2")
                sc = tr.run("dyvit_score_kernel", 0.0, 0.5 * M * D,
                            lambda: ops.dyvit_score(h3, f32(sp.out_conv[4].weight), f32(sp.out_conv[4].bias)))
                idx, _, scores = tr.run("cls_topk_kernel", 0.0, 8.0 * M, lambda: ops.cls_topk(sc.view(B, 1, N), Kc))
                info["kept"][i], info["compl"][i], info["scores"][i] = idx, None, scores
                h3_, xn = tr.run("gather_layernorm_kernel", 0.0, 10.0 * B * (Kc + 1) * D,
                                 lambda: ops.gather_layernorm(h.view(B, N, D), idx, None, None, f32(blk.norm1.weight),
                                                              f32(blk.norm1.bias), eps))
                N = Kc + 1
                h, xn = h3_.view(B * N, D), xn.view(B * N, D)
            else:
                m = model.cluster_layers[j]
                y = tr.run("layernorm_kernel", 0.0, (6.0 if pending is None else 12.0) * M * D,
                           lambda: ops.layernorm(h, f32(m.weight[0].weight), f32(m.weight[0].bias), 1e-5, delta=pending))
                pending = None
                hh = (m.weight[1].out_features + 63) // 64 * 64
                h1 = _gemm(tr, y, bf(_pad_rows(m.weight[1].weight, hh)), _pad_vec(m.weight[1].bias, hh), ops.TR_EPI_GELU_BF16, tag="s0")
                n_pad = (Kc + 7) // 8 * 8
                w1 = torch.zeros(n_pad, hh, dtype=torch.float32, device=dev)
                w1[:Kc, :m.weight[3].in_features] = m.weight[3].weight.detach()
                b1 = torch.zeros(n_pad, dtype=torch.float32, device=dev)
                b1[:Kc] = m.weight[3].bias.detach()
                lg = _gemm(tr, h1, bf(w1), b1, ops.TR_EPI_F32, tag="s1")
                scale = float(m.scale.detach().reshape(-1)[0])
                h3_, soft = tr.run("sit_merge_kernel", 2.0 * B * Kc * (N - 1) * D, 4.0 * B * (N + Kc) * D,
                                   lambda: ops.softassign_merge_fast(lg.view(B, N, n_pad), scale, h.view(B, N, D), Kc, True, tr.keep)
                                   if Kc <= 192 else ops.sit_merge(lg.view(B, N, n_pad), scale, h.view(B, N, D), Kc, want_soft=tr.keep))
                info["soft"][i] = soft
                N = Kc + 1
                h = h3_.view(B * N, D)
        K = int(cfg.keep[i]) if cfg.family in (1, 2) else 0
        Ks = int(cfg.keep[i]) if cfg.family == 7 else 0          # ATS sample_count
        if K == N - 1:
            K = 0
        r = min(int(cfg.keep[i]), (N - 1) // 2) if tome else 0
        M = B * N
        if xn is None:
            xn = tr.run("layernorm_kernel", 0.0, (6.0 if pending is None else 12.0) * M * D,
                        lambda: ops.layernorm(h, f32(blk.norm1.weight), f32(blk.norm1.bias), eps, delta=pending))
        qkv = _gemm(tr, xn, bf(blk.attn.qkv.weight), f32(blk.attn.qkv.bias), ops.TR_EPI_BF16, tag="qkv")
        colsum = None
        if cfg.family == 9 and i + 1 < len(model.blocks) and int(cfg.keep[i + 1]) > 0:
            colsum = torch.empty(B, H, 4, N, dtype=torch.float32, device=dev)      # feeds the NEXT block's K-Medoids
        ao, cls_rows = tr.run("attention_kernel", 4.0 * B * H * N * N * 64, 2.0 * M * 4 * D,
                              lambda: ops.attention(qkv, B, N, H, want_cls=K > 0 or Ks > 0, size=size, colsum_part=colsum))
        tr.save(f"attn_out_{i}", ao)
        if Ks > 0:
            # ATS: sample ids on CLS attention x |v|, keep those rows of x and of attn @ v (ats.py:52-89,157)
            steps = model.sample_steps(Ks).to(dev)
            ids, new_mask, cdf = tr.run("ats_sample_kernel", 0.0, 2.0 * M * D, lambda: ops.ats_sample(cls_rows, qkv, size, steps, Ks, want_cdf=tr.keep))
            if forced_ids is not None and i in forced_ids:        # tests: teacher forcing (ids [B,Ks] incl. CLS 0 and 0 padding)
                ids = forced_ids[i].to(device=dev, dtype=torch.int32).contiguous()
                new_mask = (ids != 0).float()
                new_mask[:, 0] = 1.0
            info["kept"][i], info["compl"][i], info["scores"][i] = ids, None, cdf
            h3_, ao = tr.run("ats_gather_kernel", 0.0, 12.0 * B * Ks * D, lambda: ops.ats_gather(h.view(B, N, D), ao, ids))
            size = new_mask
            N = Ks
            M = B * N
            h = h3_.view(M, D)
        d1 = _gemm(tr, ao, bf(blk.attn.proj.weight), f32(blk.attn.proj.bias), ops.TR_EPI_BF16, tag="d1")
        if K > 0:
            idx, compl, scores = tr.run("cls_topk_kernel", 0.0, 4.0 * B * (H * N + N),
                                        lambda: ops.cls_topk(cls_rows, K, want_compl=fuse))
            info["kept"][i], info["compl"][i], info["scores"][i] = idx, compl, scores
            Nn = K + 1 + (1 if fuse else 0)
            h3, xn = tr.run("gather_layernorm_kernel", 0.0, 12.0 * B * Nn * D,
                            lambda: ops.gather_layernorm(h.view(B, N, D), idx, compl, scores if fuse else None,
                                                         f32(blk.norm2.weight), f32(blk.norm2.bias), eps, delta=d1.view(B, N, D)))
            h = h3.view(B * Nn, D)
            xn = xn.view(B * Nn, D)
            N = Nn
        elif r > 0:
            unm, src, dst = tr.run("tome_match_kernel", 2.0 * B * (N // 2) * ((N + 1) // 2) * 64, 2.0 * M * D,
                                   lambda: ops.tome_match(qkv, B, N, H, r))
            info["tome"][i] = (unm, src, dst)
            tr.save(f"qkv_{i}", qkv)
            h3, size, xn = tr.run("tome_merge_layernorm_kernel", 0.0, 6.0 * M * D + 6.0 * B * (N - r) * D,
                                  lambda: ops.tome_merge_layernorm(h.view(B, N, D), d1.view(B, N, D), size, unm, src, dst,
                                                                   f32(blk.norm2.weight), f32(blk.norm2.bias), eps))
            N = N - r
            h = h3.view(B * N, D)
            xn = xn.view(B * N, D)
        else:
            xn = tr.run("layernorm_kernel", 0.0, 12.0 * M * D,
                        lambda: ops.layernorm(h, f32(blk.norm2.weight), f32(blk.norm2.bias), eps, delta=d1))
        hid = _gemm(tr, xn, bf(blk.mlp.fc1.weight), f32(blk.mlp.fc1.bias), ops.TR_EPI_GELU_BF16, tag="hid")
        pending = _gemm(tr, hid, bf(blk.mlp.fc2.weight), f32(blk.mlp.fc2.bias), ops.TR_EPI_BF16, tag="d2")
        info["tokens"].append(N)
        tr.save(f"block_{i}", h.view(B, N, D))
    xc = tr.run("layernorm_kernel", 0.0, 12.0 * B * D,
                lambda: ops.layernorm(h, f32(model.norm.weight), f32(model.norm.bias), eps, rows=B, ldx=N * D, delta=pending,
                                      ldd=N * D))
    logits = _gemm(tr, xc, bf(model.head.weight), f32(model.head.bias), ops.TR_EPI_F32)
    if tr.timing:
        tr.finish()
    return logits, info
