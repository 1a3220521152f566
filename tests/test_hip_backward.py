"""GPU parity tests of the backward kernels (csrc/tr_backward.hip, csrc/tr_attention_bwd.hip) through the C ABI.

The reference's backward is torch.autograd over its eager forward (engine.py:50-76), so the checker here is exactly that:
torch.autograd in fp32 (on the GPU box's torch) over a plain PyTorch restatement of the op, fed the SAME bf16-rounded
operands.  Tolerances: fp32 accumulations differ by summation order only (1e-4 relative to the result's scale); bf16 outputs
add one rounding (2^-8 relative).  Gradients through bf16-rounded intermediates (attention: P and dS are rounded to bf16 for the
MFMA products, exactly like P in the forward) are compared by relative L2 with the bound written in the test.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tokenreduction_amd import ops as _ops
    return _ops


def _randn(seed, *shape, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def rel_l2(got, want):
    got, want = got.double().cpu(), want.double().cpu()
    return float((got - want).norm() / want.norm().clamp_min(1e-30))


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (100, 8, 8), (1000, 384, 1152), (50432, 1536, 384), (3000, 1000, 384), (777, 192, 576),
                                   (1, 16, 24), (4096, 768, 3072)])
def test_wgrad(ops, M, N, K):
    dy = _randn(1, M, N, dtype=torch.bfloat16)
    x = _randn(2, M, K, dtype=torch.bfloat16)
    got = ops.wgrad(dy, x)
    want = dy.float().t() @ x.float()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-4 * scale + 1e-6, f"max err {(got - want).abs().max():.3e} scale {scale:.3e}"
    # accumulate: adds to the destination
    acc = ops.wgrad(dy, x, out=got.clone(), accumulate=True)
    assert float((acc - 2 * want).abs().max()) <= 4e-4 * scale + 1e-6
    # deterministic: bitwise identical run to run
    assert torch.equal(ops.wgrad(dy, x), got)


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (1000, 384, 1152), (50432, 1536, 384), (777, 96, 192), (3000, 1000, 384)])
def test_linear_bwd_params(ops, M, N, K):
    """Fused weight + bias gradient == the separate kernels, bit for bit (same partial sums in the same order for dW; the bias sums
    are a different, also fixed, order -> compared with the torch value)."""
    dy = _randn(50, M, N, dtype=torch.bfloat16)
    x = _randn(51, M, K, dtype=torch.bfloat16)
    dw, db = ops.linear_bwd_params(dy, x)
    assert torch.equal(dw, ops.wgrad(dy, x))
    want = dy.float().sum(0)
    assert float((db - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-5
    dw2, db2 = ops.linear_bwd_params(dy, x, accumulate=True, dw=dw.clone(), db=db.clone())
    assert torch.allclose(dw2, 2 * dw, rtol=1e-6) and torch.allclose(db2, 2 * db, rtol=1e-6)
    assert torch.equal(ops.linear_bwd_params(dy, x)[1], db)


def test_wgrad_exact_integers(ops):
    """Small integers are exact in bf16 and fp32: any operand-layout mistake shows as a wrong integer."""
    M, N, K = 200, 144, 136
    g = torch.Generator().manual_seed(0)
    dy = torch.randint(-3, 4, (M, N), generator=g).to(torch.bfloat16).cuda()
    x = torch.randint(-3, 4, (M, K), generator=g).to(torch.bfloat16).cuda()
    assert torch.equal(ops.wgrad(dy, x), dy.float().t() @ x.float())


@pytest.mark.parametrize("M,N,K", [(200, 192, 384), (64, 384, 192), (65, 192, 192), (1000, 576, 192), (12608, 1152, 384), (50432, 384, 1536)])
def test_wgrad_exact_integers_producer_consumer_kernel(ops, M, N, K):
    """N and K multiples of 192 take the LDS-DMA weight-gradient kernel (tr_wgrad_pc.hip): exact integers through its swizzled images and
    transposed reads, ragged last slabs (rows past M must contribute zero), one slab, many units per workgroup; bias sums from the
    all-ones MFMA exact as well."""
    g = torch.Generator().manual_seed(M + N)
    dy = torch.randint(-2, 3, (M, N), generator=g).to(torch.bfloat16).cuda()
    x = torch.randint(-2, 3, (M, K), generator=g).to(torch.bfloat16).cuda()
    want = dy.float().t() @ x.float()
    assert torch.equal(ops.wgrad(dy, x), want)
    dw, db = ops.linear_bwd_params(dy, x)
    assert torch.equal(dw, want)
    assert torch.equal(db, dy.float().sum(0))
    # strided operands (column slices of wider tensors, as the executor passes them)
    wide_y = torch.randint(-2, 3, (M, N + 64), generator=g).to(torch.bfloat16).cuda()
    wide_x = torch.randint(-2, 3, (M, K + 192), generator=g).to(torch.bfloat16).cuda()
    dw2, db2 = ops.linear_bwd_params(wide_y[:, :N], wide_x[:, 192:])
    assert torch.equal(dw2, wide_y[:, :N].float().t() @ wide_x[:, 192:].float())
    assert torch.equal(db2, wide_y[:, :N].float().sum(0))


def test_wgrad_patch_rows(ops):
    """yskip: dY rows are the patch rows of a [B, P+1, D] tensor (PatchEmbed weight gradient)."""
    B, P, D, Kc = 5, 36, 128, 768
    g = _randn(3, B, P + 1, D, dtype=torch.bfloat16)
    cols = _randn(4, B * P, Kc, dtype=torch.bfloat16)
    got = ops.wgrad(g.view(-1, D), cols, yskip=P, rows=B * P)
    want = g[:, 1:].reshape(-1, D).float().t() @ cols.float()
    assert float((got - want).abs().max()) <= 2e-4 * float(want.abs().max())
    gotb = ops.colsum(g.view(-1, D), yskip=P, rows=B * P)
    wantb = g[:, 1:].reshape(-1, D).float().sum(0)
    assert float((gotb - wantb).abs().max()) <= 2e-4 * float(wantb.abs().max())


@pytest.mark.parametrize("B,P,D,Kc", [(5, 36, 192, 768), (3, 196, 384, 768), (128, 196, 768, 768)])
def test_wgrad_patch_rows_producer_consumer_kernel_exact_integers(ops, B, P, D, Kc):
    """The PatchEmbed weight gradient at widths the LDS-DMA kernel takes (D, 3 * 16 * 16 multiples of 192): the CLS rows of [B, P + 1, D]
    are stepped over by its loader (row m -> m + m / P + 1), weight and bias gradient exact on small integers up to the DeiT-B size."""
    g = torch.Generator().manual_seed(B + P)
    dy = torch.randint(-2, 3, (B, P + 1, D), generator=g).to(torch.bfloat16).cuda()
    dy[:, 0] = 7.0                                     # a CLS row that leaks into the sums shows at once
    cols = torch.randint(-2, 3, (B * P, Kc), generator=g).to(torch.bfloat16).cuda()
    want = dy[:, 1:].reshape(-1, D).float().t() @ cols.float()
    assert torch.equal(ops.wgrad(dy.view(-1, D), cols, yskip=P, rows=B * P), want)
    dw, db = ops.linear_bwd_params(dy.view(-1, D), cols, yskip=P)
    assert torch.equal(dw, want)
    assert torch.equal(db, dy[:, 1:].reshape(-1, D).float().sum(0))


@pytest.mark.parametrize("M,N", [(7, 8), (128, 1000), (50432, 1536), (1000, 1000), (513, 384)])
def test_colsum(ops, M, N):
    dy = _randn(5, M, N, dtype=torch.bfloat16)
    got = ops.colsum(dy)
    want = dy.float().sum(0)
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-5
    assert torch.equal(ops.colsum(dy), got)


def test_gelu_forward_matches_fused_epilogue_and_backward_matches_autograd(ops):
    M, K, N = 300, 128, 256
    a = _randn(6, M, K, dtype=torch.bfloat16)
    w = _randn(7, N, K, scale=0.2, dtype=torch.bfloat16)
    bias = _randn(8, N)
    pre = ops.gemm(a, w, bias, ops.TR_EPI_BF16)
    assert torch.equal(ops.gelu(pre), ops.gemm(a, w, bias, ops.TR_EPI_GELU_BF16)) or \
        float((ops.gelu(pre).float() - ops.gemm(a, w, bias, ops.TR_EPI_GELU_BF16).float()).abs().max()) <= 2.0 ** -7 * 4
    x = _randn(9, 1000, 64, scale=2.0, dtype=torch.bfloat16)
    dh = _randn(10, 1000, 64, dtype=torch.bfloat16)
    xf = x.float().requires_grad_(True)
    torch.nn.functional.gelu(xf).backward(dh.float())
    got = ops.gelu_bwd(x, dh.clone()).float()
    assert float((got - xf.grad).abs().max()) <= 2.0 ** -8 * float(xf.grad.abs().max()) + 1e-6


@pytest.mark.parametrize("M,D", [(50, 384), (1001, 768), (4100, 192), (300, 128), (64, 1024)])
def test_layernorm_bwd(ops, M, D):
    x = _randn(11, M, D, scale=2.0) + 0.5
    gamma = (_randn(12, D) * 0.3 + 1.0).contiguous()
    dy = _randn(13, M, D, dtype=torch.bfloat16)
    g_in = _randn(14, M, D)
    eps = 1e-6
    xf = x.clone().requires_grad_(True)
    gf = gamma.clone().requires_grad_(True)
    bf = torch.zeros(D, device="cuda", requires_grad=True)
    torch.nn.functional.layer_norm(xf, (D,), gf, bf, eps).backward(dy.float())
    g, gb, dgamma, dbeta = ops.layernorm_bwd(dy, x, gamma, eps, g_in=g_in)
    want = g_in + xf.grad
    assert float((g - want).abs().max()) <= 1e-4 * float(want.abs().max())
    assert float((gb.float() - want).abs().max()) <= 2.0 ** -8 * float(want.abs().max())
    assert float((dgamma - gf.grad).abs().max()) <= 2e-4 * float(gf.grad.abs().max())
    assert float((dbeta - bf.grad).abs().max()) <= 2e-4 * float(bf.grad.abs().max()) + 1e-5
    g2, _, _, _ = ops.layernorm_bwd(dy, x, gamma, eps)          # no incoming gradient
    assert float((g2 - xf.grad).abs().max()) <= 1e-4 * float(xf.grad.abs().max())


@pytest.mark.parametrize("fused", [False, True])
def test_layernorm_bwd_scatter(ops, fused):
    """Backward of gather -> LayerNorm (topk.py:89-95): rows land at 1 + idx, dropped rows stay zero, EViT's fused row goes aside."""
    B, N, K, D = 3, 50, 20, 384
    n_in = K + 1 + (1 if fused else 0)
    g0 = torch.Generator().manual_seed(1)
    idx = torch.stack([torch.randperm(N - 1, generator=g0)[:K] for _ in range(B)]).to(torch.int32).cuda()
    x = _randn(15, B * n_in, D)
    gamma = (_randn(16, D) * 0.2 + 1.0).contiguous()
    dy = _randn(17, B * n_in, D, dtype=torch.bfloat16)
    g_in = _randn(18, B * n_in, D)
    xf = x.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xf, (D,), gamma, None, 1e-6).backward(dy.float())
    rows = (g_in + xf.grad).view(B, n_in, D)
    want = torch.zeros(B, N, D, device="cuda")
    want[:, 0] = rows[:, 0]
    for b in range(B):
        want[b, 1 + idx[b].long()] = rows[b, 1:K + 1]
    out = ops.layernorm_bwd(dy, x, gamma, 1e-6, g_in=g_in, idx=idx, n_out=N, fused=fused)
    g = out[0].view(B, N, D)
    assert float((g - want).abs().max()) <= 1e-4 * float(want.abs().max())
    assert float((out[1].float().view(B, N, D) - want).abs().max()) <= 2.0 ** -8 * float(want.abs().max())
    if fused:
        assert float((out[4] - rows[:, K + 1]).abs().max()) <= 1e-4 * float(want.abs().max())


def _attn_ref(qkv, B, N, H, size=None):
    q, k, v = qkv.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    if size is not None:
        s = s + size.log()[:, None, None, :]
    p = s.softmax(-1)
    return (p @ v).transpose(1, 2).reshape(B * N, H * 64), p


@pytest.mark.parametrize("B,N,H", [(2, 197, 6), (3, 50, 3), (1, 224, 2), (2, 17, 1), (2, 139, 12), (1, 64, 1), (4, 98, 3)])
@pytest.mark.parametrize("bias", [False, True])
def test_attention_bwd(ops, B, N, H, bias):
    qkv = _randn(20, B * N, 3 * H * 64, dtype=torch.bfloat16)
    dout = _randn(21, B * N, H * 64, dtype=torch.bfloat16)
    size = None
    if bias:
        size = (torch.rand(B, N, generator=torch.Generator().manual_seed(3)) * 3 + 1).floor().cuda()
        size[:, -1] = 0.0 if N > 20 else 1.0          # a masked key (ATS / heuristic masks are log 0)
    qf = qkv.float().requires_grad_(True)
    out, _ = _attn_ref(qf, B, N, H, size)
    out.backward(dout.float())
    got = ops.attention_bwd(qkv, dout, B, N, H, size=size)
    want = qf.grad.view(B * N, 3, H * 64)
    gv = got.float().view(B * N, 3, H * 64)
    for i, nm in enumerate("qkv"):
        r = rel_l2(gv[:, i], want[:, i])
        # P and dS pass through bf16 (2^-9 relative rounding each) before the three products; the result is rounded to bf16
        assert r <= 1.2e-2, f"d{nm}: rel L2 {r:.3e}"


def test_attention_bwd_cls_gradient(ops):
    """EViT: extra gradient on the head-mean CLS attention row (evit.py:117-120)."""
    B, N, H = 2, 139, 6
    qkv = _randn(22, B * N, 3 * H * 64, dtype=torch.bfloat16)
    dout = _randn(23, B * N, H * 64, scale=0.1, dtype=torch.bfloat16)
    dcls = _randn(24, B, N)
    dcls[:, 0] = 0
    qf = qkv.float().requires_grad_(True)
    out, p = _attn_ref(qf, B, N, H)
    cls_attn = p[:, :, 0, :].mean(1)
    ((out * dout.float()).sum() + (cls_attn * dcls).sum()).backward()
    got = ops.attention_bwd(qkv, dout, B, N, H, dcls=dcls).float()
    assert rel_l2(got, qf.grad) <= 1.2e-2


@pytest.mark.parametrize("B,N,H", [(2, 577, 3), (1, 290, 2), (2, 197, 6), (1, 65, 1), (3, 64, 2), (1, 640, 1), (2, 17, 1)])
@pytest.mark.parametrize("bias", [False, True])
def test_attention_bwd_long(ops, B, N, H, bias):
    """The key-blocked backward (any N; the executor's choice beyond 224 tokens) against torch.autograd, masked keys and the EViT
    d cls_attn path included; at N <= 224 also against the register-resident kernel it replaces there."""
    qkv = _randn(30, B * N, 3 * H * 64, dtype=torch.bfloat16)
    dout = _randn(31, B * N, H * 64, dtype=torch.bfloat16)
    dcls = _randn(32, B, N, scale=0.5)
    dcls[:, 0] = 0
    size = None
    if bias:
        size = (torch.rand(B, N, generator=torch.Generator().manual_seed(3)) * 3 + 1).floor().cuda()
        size[:, -1] = 0.0 if N > 20 else 1.0
    qf = qkv.float().requires_grad_(True)
    out, p = _attn_ref(qf, B, N, H, size)
    ((out * dout.float()).sum() + (p[:, :, 0, :].mean(1) * dcls).sum()).backward()
    got = ops.attention_bwd_long(qkv, dout, B, N, H, size=size, dcls=dcls)
    want = qf.grad.view(B * N, 3, H * 64)
    gv = got.float().view(B * N, 3, H * 64)
    for i, nm in enumerate("qkv"):
        r = rel_l2(gv[:, i], want[:, i])
        assert r <= 1.2e-2, f"d{nm}: rel L2 {r:.3e}"
    if N <= 224:
        short = ops.attention_bwd(qkv, dout, B, N, H, size=size, dcls=dcls)
        assert rel_l2(got.float(), short.float()) <= 8e-3          # two roundings of P / dS apart


@pytest.mark.parametrize("B,N,H", [(2, 577, 2), (1, 300, 3), (2, 197, 6), (2, 138, 2), (1, 40, 1), (1, 640, 1)])
def test_attention_policy_bwd(ops, B, N, H):
    """Backward of DyViT's training-time attention (Policy_Attention.softmax_with_policy, dyvit.py:39-67) against torch.autograd over
    the oracle's restatement: d qkv and the per-key policy gradient (the straight-through Gumbel sample upstream makes the policy
    differentiable, dyvit.py:223-224).  N <= 224: the register-resident kernel; beyond (384 x 384 inputs): the key-blocked one."""
    import oracle
    qkv = _randn(40, B * N, 3 * H * 64, dtype=torch.bfloat16)
    dout = _randn(41, B * N, H * 64, dtype=torch.bfloat16)
    policy = (torch.rand(B, N, generator=torch.Generator().manual_seed(5)) > 0.4).float()
    policy[:, 0] = 1.0                                               # CLS is always kept (dyvit.py:226)
    qf = qkv.float().cpu().requires_grad_(True)
    pol = policy.clone().requires_grad_(True)
    q, k, v = qf.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    attn = oracle.dyvit_softmax_with_policy((q @ k.transpose(-2, -1)) * 0.125, pol.unsqueeze(-1))
    out = (attn @ v).transpose(1, 2).reshape(B * N, H * 64)
    out.backward(dout.float().cpu())
    got, dpart = ops.attention_policy_bwd(qkv, dout, policy.cuda(), B, N, H)
    want = qf.grad.view(B * N, 3, H * 64)
    gv = got.float().cpu().view(B * N, 3, H * 64)
    for i, nm in enumerate("qkv"):
        r = rel_l2(gv[:, i], want[:, i])
        assert r <= 1.2e-2, f"d{nm}: rel L2 {r:.3e}"
    r = rel_l2(dpart.sum(1).cpu(), pol.grad)
    assert r <= 1.2e-2, f"d policy: rel L2 {r:.3e}"
    # the forward the executor pairs with it
    fwd = ops.attention_policy(qkv, policy.cuda(), B, N, H)
    torch.testing.assert_close(fwd.float().cpu(), out.detach(), atol=3e-2, rtol=2e-2)


def test_head_and_embed_bwd(ops):
    B, Cc, D, N = 32, 1000, 384, 68
    dl = _randn(25, B, Cc, scale=0.01)
    w = _randn(26, Cc, D, scale=0.02, dtype=torch.bfloat16)
    xn = _randn(27, B, D, dtype=torch.bfloat16)
    dxn, dw, db = ops.head_bwd(dl, w, xn)
    assert float((dxn.float() - dl @ w.float()).abs().max()) <= 2.0 ** -8 * float((dl @ w.float()).abs().max())
    dl16 = dl.to(torch.bfloat16).float()              # the weight / bias gradients take dlogits as a bf16 GEMM operand
    assert float((dw - dl16.t() @ xn.float()).abs().max()) <= 2e-4 * float((dl16.t() @ xn.float()).abs().max())
    assert float((db - dl16.sum(0)).abs().max()) <= 1e-5
    g = _randn(28, B, N, D)
    dpos, dcls = ops.embed_bwd(g)
    assert float((dpos - g.sum(0)).abs().max()) <= 1e-4 * float(g.sum(0).abs().max())
    assert float((dcls - g[:, 0].sum(0)).abs().max()) <= 1e-4 * float(g.sum(0).abs().max())


def test_evit_fuse_bwd(ops):
    B, N, K, D = 3, 60, 25, 384
    P = N - 1
    x = _randn(30, B, N, D)
    delta = _randn(31, B, N, D, dtype=torch.bfloat16)
    scores = torch.rand(B, P, generator=torch.Generator().manual_seed(5)).cuda()
    g0 = torch.Generator().manual_seed(2)
    perm = torch.stack([torch.randperm(P, generator=g0) for _ in range(B)])
    compl = perm[:, K:].sort(dim=1).values.to(torch.int32).cuda()
    g_fused = _randn(32, B, D)
    xm = (x + delta.float()).requires_grad_(True)
    sc = scores.clone().requires_grad_(True)
    rows = torch.gather(xm[:, 1:], 1, compl.long()[..., None].expand(-1, -1, D))
    wts = torch.gather(sc, 1, compl.long())
    extra = (rows * wts[..., None]).sum(1)
    (extra * g_fused).sum().backward()
    g_out = torch.zeros(B, N, D, device="cuda")
    gb_out = torch.zeros(B, N, D, dtype=torch.bfloat16, device="cuda")
    dscore = ops.evit_fuse_bwd(x, delta, compl, scores, g_fused, g_out, gb_out)
    assert float((g_out - xm.grad).abs().max()) <= 1e-5 * float(xm.grad.abs().max()) + 1e-7
    assert float((dscore[:, 1:] - sc.grad).abs().max()) <= 1e-4 * float(sc.grad.abs().max())


def test_tome_merge_bwd(ops):
    B, N, r, D = 2, 51, 9, 384
    na, nb = (N + 1) // 2, N // 2
    g0 = torch.Generator().manual_seed(7)
    x = _randn(33, B, N, D)
    size_in = (torch.rand(B, N, generator=g0) * 3 + 1).floor().cuda()
    unm, src, dst = [], [], []
    for b in range(B):
        perm = torch.randperm(na - 1, generator=g0) + 1           # A-tokens 1.. (0 = CLS stays unmerged)
        src.append(perm[:r])
        unm.append(torch.cat([torch.zeros(1, dtype=torch.long), perm[r:]]).sort().values)
        dst.append(torch.randint(0, nb, (r,), generator=g0))
    unm, src, dst = (torch.stack(t).to(torch.int32).cuda() for t in (unm, src, dst))
    xf = x.clone().requires_grad_(True)
    xs = xf * size_in[..., None]
    a_x, b_x = xs[:, 0::2], xs[:, 1::2]
    a_s, b_s = size_in[:, 0::2], size_in[:, 1::2]
    outs, sizes = [], []
    for b in range(B):
        bx = b_x[b].index_add(0, dst[b].long(), a_x[b, src[b].long()])
        bs = b_s[b].index_add(0, dst[b].long(), a_s[b, src[b].long()])
        outs.append(torch.cat([a_x[b, unm[b].long()], bx]))
        sizes.append(torch.cat([a_s[b, unm[b].long()], bs]))
    size_out = torch.stack(sizes)
    x_out = torch.stack(outs) / size_out[..., None]
    gm = _randn(34, B, N - r, D)
    (x_out * gm).sum().backward()
    g, gb = ops.tome_merge_bwd(gm, size_in, size_out.contiguous(), unm, src, dst, N)
    assert float((g - xf.grad).abs().max()) <= 1e-5 * float(xf.grad.abs().max())
    assert float((gb.float() - xf.grad).abs().max()) <= 2.0 ** -8 * float(xf.grad.abs().max())


@pytest.mark.parametrize("weighted", [True, False])
@pytest.mark.parametrize("D", [128, 768])
def test_cluster_merge_bwd(ops, weighted, D):
    """DPC-KNN CTM (merge_tokens dpcknn.py:103-132 + score Linear): backward vs torch.autograd of the same expression."""
    B, N, K = 3, 99, 49
    P = N - 1
    x0 = _randn(40, B, N, D)
    sw = _randn(41, D, scale=0.05) if weighted else None
    sb = _randn(42, 1, scale=0.02) if weighted else None
    g0 = torch.Generator().manual_seed(9)
    assign = torch.stack([torch.cat([torch.arange(K), torch.randint(0, K, (P - K,), generator=g0)])[torch.randperm(P, generator=g0)]
                          for _ in range(B)]).to(torch.int32).cuda()
    xf = x0.clone().requires_grad_(True)
    swf = sw.clone().requires_grad_(True) if weighted else None
    sbf = sb.clone().requires_grad_(True) if weighted else None
    xs = xf[:, 1:]
    w = (xs @ swf + sbf).exp() if weighted else torch.ones(B, P, device="cuda")
    onehot = torch.nn.functional.one_hot(assign.long(), K).float()            # [B,P,K]
    W = torch.einsum("bpk,bp->bk", onehot, w) + 1e-6
    merged = torch.einsum("bpk,bp,bpd->bkd", onehot, w, xs) / W[..., None]
    x1 = torch.cat([xf[:, :1], merged], dim=1)
    g_in = _randn(43, B, K + 1, D)
    (x1 * g_in).sum().backward()
    g, gb, dsw, dsb = ops.cluster_merge_bwd(g_in, x0, x1.detach().contiguous(), w.detach().contiguous() if weighted else None, assign, sw)
    assert float((g - xf.grad).abs().max()) <= 2e-4 * float(xf.grad.abs().max()), float((g - xf.grad).abs().max())
    assert float((gb.float() - xf.grad).abs().max()) <= 2.0 ** -8 * float(xf.grad.abs().max()) + 2e-4 * float(xf.grad.abs().max())
    if weighted:
        assert float((dsw - swf.grad).abs().max()) <= 2e-4 * float(swf.grad.abs().max())
        # d sb = sum_i dlog_i cancels within every cluster (sum_i w_i (x_i - x_c) = x_c * 1e-6): what is left is rounding residue of
        # terms of magnitude |dlog_i|, so the bound is relative to those, not to the (near-zero) result
        assert float((dsb - sbf.grad).abs().max()) <= 1e-3 * float(swf.grad.abs().max())


def test_ats_scatter(ops):
    B, N, Ks, D = 2, 50, 12, 384
    g = _randn(44, B, Ks, D)
    dao = _randn(45, B, Ks, D, dtype=torch.bfloat16)
    ids = torch.tensor([[0, 3, 7, 8, 20, 21, 30, 41, 49, 0, 0, 0], [0, 1, 2, 5, 9, 10, 11, 12, 13, 14, 15, 48]], dtype=torch.int32).cuda()
    gf, df = ops.ats_scatter(g, dao, ids, N)
    want = torch.zeros(B, N, D, device="cuda")
    wantd = torch.zeros(B, N, D, device="cuda")
    for b in range(B):
        for t in range(Ks):
            if t == 0 or ids[b, t] != 0:
                want[b, ids[b, t]] = g[b, t]
                wantd[b, ids[b, t]] = dao[b, t].float()
    assert torch.equal(gf, want) and torch.equal(df.float(), wantd)


# ------------------------------------------------------------------------------------------ soft-assignment reducers (tr_soft_bwd.hip)
def _soft_case(seed, B, N, K, D):
    rng = np.random.default_rng(seed)
    ld = (K + 7) // 8 * 8
    src = torch.from_numpy(rng.standard_normal((B, N, D)).astype(np.float32))
    logits = torch.zeros(B, N, ld)
    logits[:, :, :K] = torch.from_numpy(rng.standard_normal((B, N, K)).astype(np.float32))
    g = torch.from_numpy(rng.standard_normal((B, K + 1, D)).astype(np.float32)) * 0.1
    return src, logits, g, ld


@pytest.mark.parametrize("B,N,K,D", [(2, 197, 137, 128), (3, 138, 96, 384), (1, 20, 5, 64)])
def test_soft_merge_and_token_softmax_bwd(ops, B, N, K, D):
    """SiT / PatchMerger: out_k = sum_p softmax_p(scale * logits[p,k]) src[p] against float64 autograd, d scale included."""
    src, logits, g, ld = _soft_case(3, B, N, K, D)
    scale = 0.7
    s64 = src.double().requires_grad_(True)
    l64 = logits[:, 1:, :K].double().requires_grad_(True)
    sc = torch.tensor(scale, dtype=torch.float64, requires_grad=True)
    w = torch.softmax(l64 * sc, dim=1)                                   # over the tokens
    out = torch.einsum("bpk,bpd->bkd", w, s64[:, 1:])
    out.backward(g[:, 1:].double())
    wt = torch.zeros(B, N, ld)
    wt[:, 1:, :K] = w.detach().float()
    dwt, dsrc = ops.soft_merge_bwd(g.cuda(), wt.cuda(), src.cuda())
    want_dw = torch.einsum("bkd,bpd->bpk", g[:, 1:].double(), src[:, 1:].double())
    torch.testing.assert_close(dwt[:, 1:, :K].cpu().double(), want_dw, atol=1e-4, rtol=1e-4)
    assert float(dwt[:, 0].abs().max()) == 0.0
    torch.testing.assert_close(dsrc[:, 1:].cpu().double(), s64.grad[:, 1:], atol=1e-5, rtol=1e-4)
    assert float(dsrc[:, 0].abs().max()) == 0.0
    ds, dscale = ops.token_softmax_bwd(wt.cuda(), dwt, logits.cuda(), scale, K, want_dscale=True)
    got = ds[:, 1:, :K].float().cpu().double()
    rel = float((got - l64.grad).norm() / l64.grad.norm())
    assert rel < 4e-3, rel                                               # bf16 output
    assert float(ds[:, 0].float().abs().max()) == 0.0 and float(ds[:, :, K:].float().abs().max()) == 0.0
    assert abs(float(dscale) - float(sc.grad)) <= 1e-4 * max(1.0, abs(float(sc.grad))), (float(dscale), float(sc.grad))


@pytest.mark.parametrize("B,N,K,iters,eps", [(2, 197, 137, 3, 1.0), (2, 138, 96, 3, 0.5), (1, 30, 7, 5, 1.0), (2, 577, 144, 3, 1.0), (1, 401, 190, 2, 0.7)])
def test_sinkhorn_bwd(ops, B, N, K, iters, eps):
    """log_optimal_transport backwards against autograd through the oracle's restatement of sinkhorn.py:25-56 (float64)."""
    import oracle
    rng = np.random.default_rng(11)
    ld = (K + 7) // 8 * 8
    scores = torch.zeros(B, N, ld)
    scores[:, :, :K] = torch.from_numpy((rng.standard_normal((B, N, K)) * 0.3).astype(np.float32))
    dplan = torch.zeros(B, N, ld)
    dplan[:, 1:, :K] = torch.from_numpy(rng.standard_normal((B, N - 1, K)).astype(np.float32))
    s64 = scores[:, 1:, :K].double().requires_grad_(True)
    plan = oracle.sinkhorn_transport(s64.transpose(1, 2), eps, iters).transpose(1, 2)          # [B,P,K]
    plan.backward(dplan[:, 1:, :K].double())
    ds = ops.sinkhorn_bwd(scores.cuda(), dplan.cuda(), K, eps, iters)
    got = ds[:, 1:, :K].float().cpu().double()
    rel = float((got - s64.grad).norm() / s64.grad.norm())
    assert rel < 4e-3, rel
    assert float(ds[:, 0].float().abs().max()) == 0.0


def test_rownorm_bwd_and_add_into_bf16(ops):
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((77, 384)).astype(np.float32)) * 3
    da = torch.from_numpy(rng.standard_normal((77, 384)).astype(np.float32))
    db = torch.from_numpy(rng.standard_normal((77, 384)).astype(np.float32)).bfloat16()
    x64 = x.double().requires_grad_(True)
    torch.nn.functional.normalize(x64, p=2, dim=-1).backward(da.double() + db.double())
    dx = ops.rownorm_bwd(x.cuda(), da.cuda(), db.cuda())
    torch.testing.assert_close(dx.cpu().double(), x64.grad, atol=1e-6, rtol=1e-4)
    y = db.clone().cuda()
    ops.add_into_bf16(da.cuda(), y)
    assert torch.equal(y.cpu(), (da + db.float()).bfloat16())


@pytest.mark.parametrize("M,N,K", [(50432, 1536, 384), (300, 1536, 384), (30000, 384, 384), (777, 192, 576), (5, 16, 128)])
def test_gemm_gelu_keep_is_bitwise_the_two_launch_path(ops, M, N, K):
    """fc1 of the training forward in one launch: the pre-activation and its GELU must be bit for bit what tr_gemm_bf16(TR_EPI_BF16)
    followed by tr_gelu_bf16 produce (the backward differentiates the stored pre-activation)."""
    a = _randn(40, M, K, dtype=torch.bfloat16)
    w = _randn(41, N, K, scale=0.05, dtype=torch.bfloat16)
    b = _randn(42, N, scale=0.1)
    pre, h = ops.gemm_gelu_keep(a, w, b)
    pre2 = ops.gemm(a, w, b, ops.TR_EPI_BF16)
    assert torch.equal(pre, pre2)
    assert torch.equal(h, ops.gelu(pre2))


@pytest.mark.parametrize("M,N,K", [(50432, 1536, 384), (300, 1536, 384), (25216, 3072, 768), (777, 192, 576), (5, 16, 128)])
def test_gemm_dgelu_is_bitwise_the_two_launch_path(ops, M, N, K):
    """fc2's data gradient with the GELU backward folded into the epilogue: bit for bit tr_gemm_bf16(TR_EPI_BF16) followed by
    tr_gelu_bwd_bf16 (both multiply the ROUNDED data gradient by the same derivative), full tiles, half-tile tails and ragged edges."""
    a = _randn(50, M, K, dtype=torch.bfloat16)
    w = _randn(51, N, K, scale=0.05, dtype=torch.bfloat16)
    b = torch.zeros(N, device="cuda")
    pre = _randn(52, M, N, scale=1.5, dtype=torch.bfloat16)
    got = ops.gemm_dgelu(a, w, pre)
    want = ops.gelu_bwd(pre, ops.gemm(a, w, b, ops.TR_EPI_BF16))
    assert torch.equal(got, want)


@pytest.mark.parametrize("shapes", [((50432, 384, 1536), (50432, 1536, 384)), ((24832, 384, 384), (35328, 1152, 384)), ((200, 192, 192), (70, 576, 192)),
                                    ((12608, 768, 3072), (12608, 3072, 768)), ((300, 128, 256), (300, 384, 192))])
def test_linear_bwd_params2_pairs_two_layers_in_one_launch(ops, shapes):
    """Two Linear layers' parameter gradients from one weight-gradient launch: exact on small integers (every unit -> problem -> tile -> token
    range mapping), different M per layer, accumulate, run-to-run bitwise; a pair the 192-tile kernel does not take falls back to two calls."""
    (M0, N0, K0), (M1, N1, K1) = shapes
    g = torch.Generator().manual_seed(M0 + N1)
    mk = lambda m, n: torch.randint(-2, 3, (m, n), generator=g).to(torch.bfloat16).cuda()
    dy0, x0, dy1, x1 = mk(M0, N0), mk(M0, K0), mk(M1, N1), mk(M1, K1)
    (dw0, db0), (dw1, db1) = ops.linear_bwd_params2(dy0, x0, dy1, x1)
    assert torch.equal(dw0, dy0.float().t() @ x0.float()) and torch.equal(db0, dy0.float().sum(0))
    assert torch.equal(dw1, dy1.float().t() @ x1.float()) and torch.equal(db1, dy1.float().sum(0))
    outs = (dw0.clone(), db0.clone(), dw1.clone(), db1.clone())
    ops.linear_bwd_params2(dy0, x0, dy1, x1, accumulate=True, outs=outs)
    assert torch.equal(outs[0], 2 * dw0) and torch.equal(outs[3], 2 * db1)
    # random data: within fp32 summation-order noise of the separate calls, and bitwise reproducible
    dy0, x0, dy1, x1 = (_randn(60 + i, *t.shape, dtype=torch.bfloat16) for i, t in enumerate((dy0, x0, dy1, x1)))
    (a0, b0), (a1, b1) = ops.linear_bwd_params2(dy0, x0, dy1, x1)
    (c0, d0), (c1, d1) = ops.linear_bwd_params(dy0, x0), ops.linear_bwd_params(dy1, x1)
    for got, want in ((a0, c0), (b0, d0), (a1, c1), (b1, d1)):
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-6
    again = ops.linear_bwd_params2(dy0, x0, dy1, x1)
    assert torch.equal(again[0][0], a0) and torch.equal(again[1][0], a1) and torch.equal(again[1][1], b1)


@pytest.mark.parametrize("shapes", [[(50432, 384, 1536), (50432, 1536, 384), (50432, 384, 384), (50432, 1152, 384)],
                                    [(24832, 384, 1536), (24832, 1536, 384), (35328, 384, 384), (35328, 1152, 384)],
                                    [(200, 192, 192), (70, 576, 192), (64, 192, 384)],
                                    [(12672, 768, 3072), (12672, 3072, 768), (12672, 768, 768), (25216, 2304, 768)],
                                    [(3200, 768, 3072), (3200, 3072, 768), (3200, 768, 768), (3200, 2304, 768)],
                                    [(300, 128, 256), (300, 384, 192), (300, 192, 192)]])
def test_linear_bwd_group_up_to_four_layers_in_one_launch(ops, shapes):
    """The block's four parameter-gradient products from one weight-gradient launch: exact on small integers (unit -> layer -> tile -> token
    range), layers with different token counts, three layers, a group with a layer the 192-tile kernel does not take (separate calls),
    accumulate, run-to-run bitwise.  The short groups (3200 and 200 rows: one token range per layer) store their results in place
    without a reduce launch when they overwrite, and go through partials + reduce when they accumulate: both asserted here."""
    g = torch.Generator().manual_seed(shapes[0][0] + len(shapes))
    mk = lambda m, n: torch.randint(-2, 3, (m, n), generator=g).to(torch.bfloat16).cuda()
    layers = [(mk(M, N), mk(M, K)) for M, N, K in shapes]
    outs = ops.linear_bwd_group(layers)
    for (dy, x), (dw, db) in zip(layers, outs):
        assert torch.equal(dw, dy.float().t() @ x.float()) and torch.equal(db, dy.float().sum(0))
    acc = [(dw.clone(), db.clone()) for dw, db in outs]
    ops.linear_bwd_group(layers, accumulate=True, outs=acc)
    for (dw, db), (aw, ab) in zip(outs, acc):
        assert torch.equal(aw, 2 * dw) and torch.equal(ab, 2 * db)
    layers = [(_randn(70 + 2 * i, *dy.shape, dtype=torch.bfloat16), _randn(71 + 2 * i, *x.shape, dtype=torch.bfloat16)) for i, (dy, x) in enumerate(layers)]
    a = ops.linear_bwd_group(layers)
    for (dy, x), (dw, db) in zip(layers, a):
        cw, cb = ops.linear_bwd_params(dy, x)
        assert float((dw - cw).abs().max()) <= 1e-5 * float(cw.abs().max()) + 1e-6
        assert float((db - cb).abs().max()) <= 1e-5 * float(cb.abs().max()) + 1e-6
    b = ops.linear_bwd_group(layers)
    assert all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(a, b))
