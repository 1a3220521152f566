"""torch.library registration of the C-ABI entry points (tokenreduction_amd/torch_ops.py): schemas, fake (meta) shapes, loud
failure without the GPU -- on CPU; on the GPU the ops against the ctypes wrappers, their autograd formulas against torch.autograd,
and a torch.compile trace without graph breaks."""
import numpy as np
import pytest
import torch

import tokenreduction_amd.torch_ops as T

OPS = torch.ops.tokenreduction_amd


def test_every_op_is_registered_with_a_schema_and_a_fake_kernel():
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in T.OPS:
        op = getattr(OPS, name)
        assert op.default._schema.name == f"tokenreduction_amd::{name}"
    with FakeTensorMode():
        a = torch.empty(394, 384, dtype=torch.bfloat16, device="cuda")
        w = torch.empty(1152, 384, dtype=torch.bfloat16, device="cuda")
        b = torch.empty(1152, dtype=torch.float32, device="cuda")
        qkv = OPS.linear(a, w, b, 0)
        assert qkv.shape == (394, 1152) and qkv.dtype == torch.bfloat16
        out, cls = OPS.attention(qkv, 2, 197, 6, True, None)
        assert out.shape == (394, 384) and cls.shape == (2, 6, 197) and cls.dtype == torch.float32
        idx, compl, scores = OPS.cls_topk(cls, 137, True)
        assert idx.shape == (2, 137) and compl.shape == (2, 59) and idx.dtype == torch.int32 and scores.shape == (2, 196)
        x = torch.empty(2, 197, 384, device="cuda")
        g = torch.empty(384, device="cuda")
        xo, y = OPS.gather_layernorm(x, idx, compl, scores, g, g, 1e-6, None)
        assert xo.shape == (2, 139, 384) and y.dtype == torch.bfloat16
        unm, src, dst = OPS.tome_match(qkv, 2, 197, 6, 16)
        assert unm.shape == (2, 99 - 16) and src.shape == dst.shape == (2, 16)
        da, dw, db = OPS.linear_bwd(qkv, a, w)
        assert da.shape == a.shape and dw.shape == w.shape and dw.dtype == torch.float32 and db.shape == (1152,)


def test_cpu_tensors_fail_loudly():
    a, w, b = torch.zeros(8, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16), torch.zeros(8)
    with pytest.raises(NotImplementedError):
        OPS.linear(a, w, b, 0)


def _randn(seed, *shape, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


@pytest.mark.gpu
def test_ops_match_the_ctypes_wrappers_and_differentiate():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tokenreduction_amd import ops
    B, N, H, D = 2, 197, 6, 384
    x = _randn(1, B * N, D).cuda()
    g, be = (1 + _randn(2, D, scale=0.1)).cuda(), _randn(3, D, scale=0.1).cuda()
    w = _randn(4, 3 * D, D, scale=0.05).cuda().bfloat16()
    b = _randn(5, 3 * D, scale=0.1).cuda()
    xn = OPS.layernorm(x, g, be, 1e-6)
    assert torch.equal(xn, ops.layernorm(x.clone(), g, be, 1e-6))
    qkv = OPS.linear(xn, w, b, ops.TR_EPI_BF16)
    assert torch.equal(qkv, ops.gemm(xn, w, b, ops.TR_EPI_BF16))
    out, cls = OPS.attention(qkv, B, N, H, True, None)
    o2, c2 = ops.attention(qkv, B, N, H, want_cls=True)
    assert torch.equal(out, o2) and torch.equal(cls, c2)
    idx, compl, scores = OPS.cls_topk(cls, 137, False)
    assert torch.equal(idx, ops.cls_topk(cls, 137)[0]) and compl.numel() == 0

    # autograd through the registered formulas against torch.autograd on the same bf16-rounded operands
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    gr, br, bb = g.clone().requires_grad_(True), be.clone().requires_grad_(True), b.clone().requires_grad_(True)
    h = OPS.linear(OPS.layernorm(xr, gr, br, 1e-6), wr, bb, ops.TR_EPI_BF16)
    o, _ = OPS.attention(h, B, N, H, False, None)
    o = OPS.gelu(o)
    tgt = _randn(9, B * N, D).cuda()
    (o.float() * tgt).sum().backward()
    x2 = x.clone().requires_grad_(True)
    w2 = w.float().clone().requires_grad_(True)
    g2, b2, bb2 = g.clone().requires_grad_(True), be.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ln = torch.nn.functional.layer_norm(x2, (D,), g2, b2, 1e-6).bfloat16().float()
    h2 = (ln @ w2.t() + bb2).bfloat16().float()
    q, k, v = h2.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    o2 = ((q @ k.transpose(-2, -1) * 0.125).softmax(-1) @ v).transpose(1, 2).reshape(B * N, D)
    o2 = torch.nn.functional.gelu(o2.bfloat16().float())
    (o2 * tgt).sum().backward()

    def rel(a_, b_):
        return float((a_.double() - b_.double()).norm() / b_.double().norm())
    assert rel(xr.grad, x2.grad) < 3e-2, rel(xr.grad, x2.grad)
    assert rel(wr.grad.float(), w2.grad) < 3e-2
    assert rel(gr.grad, g2.grad) < 3e-2 and rel(br.grad, b2.grad) < 3e-2 and rel(bb.grad, bb2.grad) < 3e-2


@pytest.mark.gpu
def test_ops_trace_under_torch_compile_without_graph_breaks():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tokenreduction_amd import ops
    B, N, H, D = 2, 197, 6, 384
    x = _randn(1, B * N, D).cuda()
    g, be = (1 + _randn(2, D, scale=0.1)).cuda(), _randn(3, D, scale=0.1).cuda()
    w = _randn(4, 3 * D, D, scale=0.05).cuda().bfloat16()
    b = _randn(5, 3 * D, scale=0.1).cuda()

    def f(x_, w_, b_, g_, be_):
        q = OPS.linear(OPS.layernorm(x_, g_, be_, 1e-6), w_, b_, 0)
        o, c = OPS.attention(q, B, N, H, True, None)
        i, _, _ = OPS.cls_topk(c, 137, False)
        return o, i

    o_e, i_e = f(x, w, b, g, be)
    o_c, i_c = torch.compile(f, backend="aot_eager", fullgraph=True)(x, w, b, g, be)
    assert torch.equal(o_e, o_c) and torch.equal(i_e, i_c)
