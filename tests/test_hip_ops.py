"""GPU parity tests, op level: every HIP kernel (through the C ABI via tokenreduction_amd.ops) against the
oracle on the same inputs.  Integer outputs are bit-exact; floating-point tolerances are written per test.

bf16 kernels take bf16 operands, so the oracle is fed the SAME bf16-rounded operands (as fp32) and accumulates
in fp32 on the CPU; what remains is accumulation order (<= 1e-5 relative) and, for bf16 outputs, one final
rounding (1 bf16 ulp = 2^-8 relative).
"""
import math
import os

import numpy as np
import pytest
import torch

import oracle
from tests._params import make_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

BF16_ULP = 2.0 ** -8


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tokenreduction_amd import ops as _ops
    return _ops


def _rng(seed):
    return np.random.default_rng(seed)


def _randn(rng, *shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def _bf(t):
    return oracle.round_bf16(t)


def assert_close_bf16(got, want, what, ulps=1.0, abs_floor=1e-5):
    got, want = got.float().cpu(), want.float().cpu()
    tol = ulps * BF16_ULP * want.abs() + abs_floor
    bad = (got - want).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max abs err {(got - want).abs().max():.3e}"


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 1152, 384), (197 * 3, 384, 1536), (256, 1000, 384), (5, 16, 128),
                                   (1000, 1536, 384), (70000, 384, 384), (513, 192, 192), (50432, 384, 128), (1, 64, 64)])
@pytest.mark.parametrize("epi", ["bf16", "gelu", "resid", "f32"])
def test_gemm(ops, M, N, K, epi):
    if epi == "resid" and N % 64:
        # the in-place residual epilogue is only ever used with N = embed_dim (a multiple of 64); the C ABI rejects others
        a = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
        w = torch.zeros(N, K, dtype=torch.bfloat16, device="cuda")
        with pytest.raises(RuntimeError, match="N % 64"):
            ops.gemm(a, w, torch.zeros(N, device="cuda"), ops.TR_EPI_RESID_F32, out=torch.zeros(M, N, device="cuda"))
        return
    rng = _rng(M * 7 + N * 3 + K)
    a, w, b = _bf(_randn(rng, M, K)), _bf(_randn(rng, N, K, scale=0.05)), _randn(rng, N, scale=0.1)
    ref = a.double() @ w.double().t() + b.double()
    ad, wd, bd = a.cuda().bfloat16(), w.cuda().bfloat16(), b.cuda()
    if epi == "bf16":
        out = ops.gemm(ad, wd, bd, ops.TR_EPI_BF16)
        assert_close_bf16(out, ref.float(), "gemm bf16", ulps=1.01, abs_floor=2e-4)
    elif epi == "gelu":
        out = ops.gemm(ad, wd, bd, ops.TR_EPI_GELU_BF16)
        assert_close_bf16(out, oracle.gelu_erf(ref).float(), "gemm gelu", ulps=1.01, abs_floor=3e-4)
    elif epi == "resid":
        r = _randn(rng, M, N)
        out = r.clone().cuda()
        ops.gemm(ad, wd, bd, ops.TR_EPI_RESID_F32, out=out)
        torch.testing.assert_close(out.cpu(), (ref + r.double()).float(), atol=2e-4, rtol=1e-5)
    else:
        out = ops.gemm(ad, wd, bd, ops.TR_EPI_F32)
        torch.testing.assert_close(out.cpu(), ref.float(), atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize("M,N,K,why", [
    (30000, 384, 384, "354 tiles = 1 round + 98: tail as 196 half tiles; the last M tile has 48 rows, so one half tile has NO valid row"),
    (50432, 384, 1536, "591 tiles: 2 rounds + 79 tail tiles split (fc2 of the headline config)"),
    (24832, 1152, 384, "873 tiles: 3 rounds + 105 split (qkv at 97 tokens)"),
    (12000, 384, 384, "141 tiles < 256 and > 128: ONE round of full tiles, nothing split"),
    (9000, 256, 128, "72 tiles <= 128: every tile runs as two half tiles"),
    (129, 136, 64, "2 tiles, both ragged: second half tile of rows 128..255 holds one valid row, columns 128..135 ragged"),
])
@pytest.mark.parametrize("epi", ["bf16", "gelu"])
def test_gemm_half_tile_tail_round(ops, M, N, K, why, epi):
    """gemm_bf16_pc cuts the tail round into 128-row half tiles when at most half the workgroups would get a tile (tr_gemm.hip): every
    row of the output against the float64 Linear on the bf16-rounded operands, for tile counts on both sides of every scheduling case."""
    rng = np.random.default_rng(M + N)
    a, w, b = _randn(rng, M, K).bfloat16(), _randn(rng, N, K, scale=0.05).bfloat16(), _randn(rng, N, scale=0.1)
    out = ops.gemm(a.cuda(), w.cuda(), b.cuda(), ops.TR_EPI_GELU_BF16 if epi == "gelu" else ops.TR_EPI_BF16)
    rows = torch.cat([torch.arange(0, min(M, 300)), torch.arange(max(0, M - 700), M), torch.arange(0, M, 997)]).unique()
    ref = a[rows].double() @ w.double().t() + b.double()
    if epi == "gelu":
        ref = oracle.gelu_erf(ref)
    torch.testing.assert_close(out[rows.cuda()].cpu().double(), ref, atol=2e-2, rtol=1.2e-2)
    # and nothing outside the [M, N] output was touched / left unwritten: a second call on a poisoned buffer gives the same bits
    out2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm(a.cuda(), w.cuda(), b.cuda(), ops.TR_EPI_GELU_BF16 if epi == "gelu" else ops.TR_EPI_BF16, out=out2)
    assert torch.equal(out, out2) and not torch.isnan(out2.float()).any()


# ------------------------------------------------------------------------------------------ fused eval Mlp: stream-K corner cases
@pytest.mark.parametrize("Hd", [64, 192, 1536])
@pytest.mark.parametrize("nblk,ragged", [(257, 0), (258, 5), (263, 0), (300, 77), (384, 0), (511, 1), (513, 127), (1100, 0)])
def test_mlp_fused_stream_k_ranges(ops, nblk, ragged, Hd):
    """The stream-K schedule at its corners: ranges barely longer than a block (257 blocks on 256 workgroups: head segments and tails of ONE step,
    a head followed directly by the workgroup's own tail), blocks of 2 and 6 steps (shorter than the two-step lag between the fc1 and the
    fc2 waves), several whole blocks between head and tail, ragged last blocks -- each launched four times on a scratch whose slots still hold
    the previous launch's accumulators, every output equal to the two-launch pair bit for bit."""
    D = 384
    M = (nblk - 1) * 128 + (ragged if ragged else 128)
    if Hd == 1536 and nblk > 600:
        pytest.skip("covered by the 70,001-row case")
    rng = _rng(nblk * 7 + Hd)
    x = _randn(rng, M, D).bfloat16().cuda()
    w1, w2 = _randn(rng, Hd, D, scale=0.05).bfloat16().cuda(), _randn(rng, D, Hd, scale=0.05).bfloat16().cuda()
    b1, b2 = _randn(rng, Hd, scale=0.1).cuda(), _randn(rng, D, scale=0.1).cuda()
    want = ops.gemm(ops.gemm(x, w1, b1, ops.TR_EPI_GELU_BF16), w2, b2, ops.TR_EPI_BF16)
    pk = ops.mlp_pack(w1, w2, b2)
    for it in range(4):
        guard = torch.full((M + 3, D), float("nan"), dtype=torch.bfloat16, device="cuda")
        got = ops.mlp_fused(x, pk, b1, out=guard[:M], streamk=True)
        d = got.view(torch.int16) != want.view(torch.int16)
        assert not bool(d.any()), f"launch {it}: {int(d.sum())} elements differ, blocks {(d.any(dim=1).nonzero().flatten() // 128).unique().tolist()[:8]}"
        assert torch.isnan(guard[M:].float()).all(), "rows beyond M were written"


# ------------------------------------------------------------------------------------------ norm2 inside the fused Mlp launch
@pytest.mark.parametrize("M", [1, 77, 129, 1000, 24832, 257 * 128, 35328 + 3, 50432])
def test_mlp_fused_ln_is_bit_identical_to_layernorm_then_mlp(ops, M):
    """tr_mlp_fused_ln_bf16 (topk.py:95 `self.mlp(self.norm2(x))` in one launch: the fc1 waves normalise x + delta in registers) against
    the two launches it replaces -- tr_layernorm2_bf16 without a stream write-back, then tr_mlp_fused_bf16 -- BIT FOR BIT: the row sums
    are formed in the LayerNorm kernel's order.  Rows with a large common offset (|mean| >> std), a constant row and rows beyond M
    (guard) included; the stream and the pending residual must come back unchanged; stream-K and whole-block schedules agree."""
    D, Hd = 384, 1536
    rng = _rng(M * 11 + 3)
    x = (2.0 * _randn(rng, M, D))
    x[::7] += 300.0                      # rows whose mean dwarfs their spread: the two-pass variance must survive
    if M > 3:
        x[3] = 1.25                      # a constant row: variance exactly 0 before the pending residual is added
    delta = _randn(rng, M, D).bfloat16()
    x, delta = x.cuda(), delta.cuda()
    w1, w2 = _randn(rng, Hd, D, scale=0.05).bfloat16().cuda(), _randn(rng, D, Hd, scale=0.05).bfloat16().cuda()
    b1, b2 = _randn(rng, Hd, scale=0.1).cuda(), _randn(rng, D, scale=0.1).cuda()
    g, bt = (1.0 + 0.2 * _randn(rng, D)).cuda(), (0.1 * _randn(rng, D)).cuda()
    pk = ops.mlp_pack(w1, w2, b2)
    x_before, d_before = x.clone(), delta.clone()
    xn = ops.layernorm2(x, g, bt, 1e-6, delta, write_x=False)
    assert torch.equal(x, x_before), "layernorm2(write_x=False) wrote the stream"
    want = ops.mlp_fused(xn, pk, b1)
    for streamk in (True, False):
        guard = torch.full((M + 3, D), float("nan"), dtype=torch.bfloat16, device="cuda")
        got = ops.mlp_fused_ln(x, delta, g, bt, 1e-6, pk, b1, out=guard[:M], streamk=streamk)
        diff = got.view(torch.int16) != want.view(torch.int16)
        assert not bool(diff.any()), (f"streamk={streamk}: {int(diff.sum())} of {got.numel()} elements differ from LayerNorm + fused Mlp, rows "
                                      f"{diff.any(dim=1).nonzero().flatten()[:8].tolist()}")
        assert torch.isnan(guard[M:].float()).all(), "rows beyond M were written"
    assert torch.equal(x, x_before) and torch.equal(delta.view(torch.int16), d_before.view(torch.int16)), "an input was written"
    ops.mlp_fused_status()


def test_mlp_fused_ln_repeated_launches_under_uneven_load(ops):
    """Race screen of the norm-prologue launch (the plain launch's is below): 40 launches at two blocks per workgroup, every third beside a
    copy kernel on a second stream, each compared with LayerNorm + fused Mlp."""
    M, D, Hd = 35328, 384, 1536
    rng = _rng(17)
    x, delta = (2.0 * _randn(rng, M, D)).cuda(), _randn(rng, M, D).bfloat16().cuda()
    w1, w2 = _randn(rng, Hd, D, scale=0.05).bfloat16().cuda(), _randn(rng, D, Hd, scale=0.05).bfloat16().cuda()
    b1, b2 = _randn(rng, Hd, scale=0.1).cuda(), _randn(rng, D, scale=0.1).cuda()
    g, bt = (1.0 + 0.2 * _randn(rng, D)).cuda(), (0.1 * _randn(rng, D)).cuda()
    pk = ops.mlp_pack(w1, w2, b2)
    want = ops.mlp_fused(ops.layernorm2(x, g, bt, 1e-6, delta, write_x=False), pk, b1)
    side, junk = torch.cuda.Stream(), torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    out = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    bad = 0
    for it in range(40):
        out.fill_(float("nan"))
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk.add_(1)
        ops.mlp_fused_ln(x, delta, g, bt, 1e-6, pk, b1, out=out)
        torch.cuda.synchronize()
        bad += 0 if torch.equal(out.view(torch.int16), want.view(torch.int16)) else 1
    assert bad == 0, f"{bad} of 40 launches differ from LayerNorm + fused Mlp"


_GRID_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["TR_ROOT"])
import torch
from tokenreduction_amd import ops, _lib
D, Hd = 384, 1536
g = torch.Generator().manual_seed(3)
for nblk, ragged in ((257, 5), (394, 0), (130, 77), (64, 0)):
    M = (nblk - 1) * 128 + (ragged or 128)
    x = torch.randn(M, D, generator=g).bfloat16().cuda()
    w1, w2 = (0.05 * torch.randn(Hd, D, generator=g)).bfloat16().cuda(), (0.05 * torch.randn(D, Hd, generator=g)).bfloat16().cuda()
    b1, b2 = (0.1 * torch.randn(Hd, generator=g)).cuda(), (0.1 * torch.randn(D, generator=g)).cuda()
    want = ops.gemm(ops.gemm(x, w1, b1, ops.TR_EPI_GELU_BF16), w2, b2, ops.TR_EPI_BF16)
    pk = ops.mlp_pack(w1, w2, b2)
    for it in range(3):
        got = ops.mlp_fused(x, pk, b1)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (nblk, it)
    ops.mlp_fused_status()
mode = os.environ.get("TR_TEST_MODE")
if mode == "poll":
    # a poll bound of ZERO iterations: every consumer of a hand-over gives up at once.  The launch must END (no trap: this process
    # lives), and the status check must turn the device-side record into an error and clear it
    ops.set_mlp_poll_max(0)
    M = 41 * 128                       # 41 blocks on 8 workgroups: every range starts or ends inside a block
    x = torch.randn(M, D, generator=g).bfloat16().cuda()
    ops.mlp_fused(x, pk, b1)
    torch.cuda.synchronize()
    try:
        ops.mlp_fused_status()
    except RuntimeError as e:
        assert "hand-over" in str(e), str(e)
        ops.mlp_fused_status()          # the record was cleared by the read
        ops.set_mlp_poll_max(-1)        # the default bound again: the same launch is clean and correct
        want = ops.gemm(ops.gemm(x, w1, b1, ops.TR_EPI_GELU_BF16), w2, b2, ops.TR_EPI_BF16)
        assert torch.equal(ops.mlp_fused(x, pk, b1).view(torch.int16), want.view(torch.int16))
        ops.mlp_fused_status()
        print("grid ok")
        sys.exit(0)
    print("the status check did not report the abandoned hand-over")
    sys.exit(5)
print("grid ok")
"""


@pytest.mark.parametrize("grid,mode", [(96, ""), (37, ""), (8, "poll")])
def test_mlp_fused_on_a_smaller_grid(ops, tmp_path, grid, mode):
    """The schedule takes its workgroup count from the device (hipDeviceAttributeMultiprocessorCount), not a literal 256: with another grid
    (TR_MLP_FUSED_GRID, read at library load -> a subprocess) -- a partitioned or CU-masked device, the workgroups of the stream-K chain
    running in several rounds -- every output still equals the GEMM pair bit for bit.  mode "poll": with the hand-over poll bound at zero
    iterations every consumer gives up: the launch ends, tr_mlp_fused_status turns the device-side record into an error (no trap: the
    process lives) and clears it."""
    import subprocess
    import sys
    script = tmp_path / "grid_worker.py"
    script.write_text(_GRID_WORKER)
    env = dict(os.environ, TR_ROOT=ROOT, TR_MLP_FUSED_GRID=str(grid), TR_TEST_MODE=mode)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "grid ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


# ------------------------------------------------------------------------------------------ LayerNorm + Linear in one launch (norm1 + qkv)
@pytest.mark.parametrize("M,N,nd", [(1, 1152, 2), (77, 1152, 1), (129, 1152, 0), (1000, 1152, 2), (4321, 384, 2), (5000, 1536, 1), (17408, 1152, 2),
                                    (24832 + 5, 1152, 2), (35328, 1152, 1), (50432, 1152, 2), (50432, 1152, 0), (3000, 128, 2)])
def test_lnlin_is_bit_identical_to_layernorm_then_gemm(ops, M, N, nd):
    """tr_lnlin_bf16 (topk.py:86-87 norm1 + :44 qkv, with the pending residual adds of the previous block) against the launches it
    replaces -- tr_layernorm_bf16 / tr_layernorm2_bf16, then tr_gemm_bf16 -- BIT FOR BIT, outputs AND the rewritten stream: 0, 1 and 2
    pending residuals; one row to the headline's stage sizes (ranges of steps that start and end inside blocks, blocks shared by two
    workgroups, first blocks with a single step in a range); ragged last blocks; rows with |mean| >> std and a constant row; guard rows
    behind every output; inputs unchanged."""
    D = 384
    rng = _rng(M * 13 + N + nd)
    x = 2.0 * _randn(rng, M, D)
    x[::7] += 300.0
    if M > 3:
        x[3] = 1.25
    x = x.cuda()
    d1 = _randn(rng, M, D).bfloat16().cuda() if nd >= 1 else None
    d2 = _randn(rng, M, D).bfloat16().cuda() if nd >= 2 else None
    w = _randn(rng, N, D, scale=0.05).bfloat16().cuda()
    bias = _randn(rng, N, scale=0.1).cuda()
    g, bt = (1.0 + 0.2 * _randn(rng, D)).cuda(), (0.1 * _randn(rng, D)).cuda()
    x_before = x.clone()
    xr = x.clone()
    if nd == 0:
        xn = ops.layernorm(xr, g, bt, 1e-6)
    elif nd == 1:
        xn = ops.layernorm(xr, g, bt, 1e-6, delta=d1)
    else:
        xn = ops.layernorm2(xr, g, bt, 1e-6, d1, d2)
    want = ops.gemm(xn, w, bias, ops.TR_EPI_BF16)
    pk = ops.lnlin_pack(w)
    for it in range(2):
        guard = torch.full((M + 3, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        got, x_out = ops.lnlin(x, g, bt, 1e-6, pk, bias, d1=d1, d2=d2, out=guard[:M])
        diff = got.view(torch.int16) != want.view(torch.int16)
        assert not bool(diff.any()), (f"launch {it}: {int(diff.sum())} of {got.numel()} outputs differ from LayerNorm + GEMM; rows "
                                      f"{diff.any(dim=1).nonzero().flatten()[:8].tolist()}, columns {diff.any(dim=0).nonzero().flatten()[:8].tolist()}")
        assert torch.isnan(guard[M:].float()).all(), "rows beyond M were written"
        if nd:
            assert torch.equal(x_out, xr), f"launch {it}: the rewritten stream differs in {int((x_out != xr).sum())} elements"
        else:
            assert x_out is None
        assert torch.equal(x, x_before), "the input stream was written"


def test_lnlin_repeated_launches_under_uneven_load(ops):
    """Race screen (the launch hands rows from its LN waves to its MFMA waves through L2, refills registers in place and runs a 3-slot
    LDS ring behind counted waits): 40 launches at the headline's first-stage shape, every third beside a copy kernel on a second
    stream, outputs and stream compared with LayerNorm + GEMM each time."""
    M, D, N = 50432, 384, 1152
    rng = _rng(23)
    x = (2.0 * _randn(rng, M, D)).cuda()
    d1, d2 = _randn(rng, M, D).bfloat16().cuda(), _randn(rng, M, D).bfloat16().cuda()
    w, bias = _randn(rng, N, D, scale=0.05).bfloat16().cuda(), _randn(rng, N, scale=0.1).cuda()
    g, bt = (1.0 + 0.2 * _randn(rng, D)).cuda(), (0.1 * _randn(rng, D)).cuda()
    xr = x.clone()
    want = ops.gemm(ops.layernorm2(xr, g, bt, 1e-6, d1, d2), w, bias, ops.TR_EPI_BF16)
    pk = ops.lnlin_pack(w)
    side, junk = torch.cuda.Stream(), torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    bad = 0
    for it in range(40):
        out.fill_(float("nan"))
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk.add_(1)
        _, x_out = ops.lnlin(x, g, bt, 1e-6, pk, bias, d1=d1, d2=d2, out=out)
        torch.cuda.synchronize()
        bad += 0 if (torch.equal(out.view(torch.int16), want.view(torch.int16)) and torch.equal(x_out, xr)) else 1
    assert bad == 0, f"{bad} of 40 launches differ from LayerNorm + GEMM"


# ------------------------------------------------------------------------------------------ fused block tail: Mlp + residual + next norm1
@pytest.mark.parametrize("M", [1, 77, 129, 1000, 24832, 257 * 128, 35328 + 3, 50432])
def test_mlp_fused_resid_ln(ops, M):
    """tr_mlp_fused_resid_ln_bf16 (topk.py:95 `x = x + self.mlp(self.norm2(x))` and the next block's :87 norm1 in one launch): the stream row
    must equal x + Mlp(xn) up to the bf16 rounding of the Mlp output that this form does NOT do (|diff| <= half a bf16 ulp of the pair's output),
    sampled rows must match the float64 Mlp on the bf16-rounded operands more closely than the pair does, the norm output must be torch's
    LayerNorm of the kernel's own stream row to one bf16 ulp, rows beyond M stay untouched, and the stream-K schedule gives the same bits as
    whole blocks."""
    D, Hd = 384, 1536
    rng = _rng(M * 5 + 1)
    xn = _randn(rng, M, D).bfloat16().cuda()
    x0 = (2.0 * _randn(rng, M, D)).cuda()
    w1, w2 = _randn(rng, Hd, D, scale=0.05).bfloat16().cuda(), _randn(rng, D, Hd, scale=0.05).bfloat16().cuda()
    b1, b2 = _randn(rng, Hd, scale=0.1).cuda(), _randn(rng, D, scale=0.1).cuda()
    g, bt = (1.0 + 0.2 * _randn(rng, D)).cuda(), (0.1 * _randn(rng, D)).cuda()
    pk = ops.mlp_pack(w1, w2, b2)
    d = ops.mlp_fused(xn, pk, b1)                       # bit-identical to the GEMM pair (test above)
    outs = []
    for streamk in (True, False):
        xg = torch.full((M + 2, D), float("nan"), dtype=torch.float32, device="cuda")
        xg[:M] = x0
        yg = torch.full((M + 2, D), float("nan"), dtype=torch.bfloat16, device="cuda")
        ops.mlp_fused_resid_ln(xn, pk, b1, b2, xg[:M], g, bt, 1e-6, xn_next=yg[:M], streamk=streamk)
        assert torch.isnan(xg[M:]).all() and torch.isnan(yg[M:].float()).all(), "rows beyond M were written"
        outs.append((xg[:M].clone(), yg[:M].clone()))
    (xa, ya), (xb, yb) = outs
    assert torch.equal(xa, xb) and torch.equal(ya.view(torch.int16), yb.view(torch.int16)), "stream-K and whole-block schedules differ"
    # (+ the fp32 roundings of a sum that is accumulated ON the stream value: 48 steps at ulp(|x| <= 16) ~ 1e-6 each)
    ulp_half = d.float().abs() * 2.0 ** -8 + 1e-4
    assert bool(((xa - (x0 + d.float())).abs() <= ulp_half).all()), "stream row is not x + Mlp(xn) to the rounding of the Mlp output"
    ref_y = torch.nn.functional.layer_norm(xa, (D,), g, bt, 1e-6)
    err = (ya.float() - ref_y).abs()
    assert bool((err <= ref_y.abs() * 2.0 ** -7 + 1e-3).all()), float(err.max())
    rows = torch.cat([torch.arange(0, min(M, 130)), torch.arange(max(0, M - 130), M)]).unique()
    hid = oracle.gelu_erf(xn[rows.cuda()].cpu().double() @ w1.cpu().double().t() + b1.cpu().double())
    ref = x0[rows.cuda()].cpu().double() + _bf(hid.float()).double() @ w2.cpu().double().t() + b2.cpu().double()
    e_new = (xa[rows.cuda()].cpu().double() - ref).abs().max()
    e_old = ((x0 + d.float())[rows.cuda()].cpu().double() - ref).abs().max()
    assert e_new <= e_old + 1e-3, (float(e_new), float(e_old))


@pytest.mark.parametrize("M", [257 * 128, 35328 + 3])
def test_mlp_fused_resid_ln_short_tails_race_screen(ops, M):
    """Race screen for the fused tail under the stream-K schedule where a range ends in a one- or two-step tail (257 / 277 blocks on 256
    workgroups): the range's last time step has no barrier of its own, and a C wave that left the hand-over poll early used to write its
    epilogue's vectors over W2 pieces its slower neighbours had yet to read -- 1-4 % of back-to-back launches came out with one 16-column
    fragment of one block off by a step's contribution (tools/lab/resid_ln_soak.py).  300 back-to-back launches must all equal the
    whole-block schedule's bits."""
    D, Hd = 384, 1536
    rng = _rng(M + 11)
    xn = _randn(rng, M, D).bfloat16().cuda()
    x0 = (2.0 * _randn(rng, M, D)).cuda()
    w1, w2 = _randn(rng, Hd, D, scale=0.05).bfloat16().cuda(), _randn(rng, D, Hd, scale=0.05).bfloat16().cuda()
    b1, b2 = _randn(rng, Hd, scale=0.1).cuda(), _randn(rng, D, scale=0.1).cuda()
    g, bt = (1.0 + 0.2 * _randn(rng, D)).cuda(), (0.1 * _randn(rng, D)).cuda()
    pk = ops.mlp_pack(w1, w2, b2)
    xw, yw = x0.clone(), torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    ops.mlp_fused_resid_ln(xn, pk, b1, b2, xw, g, bt, 1e-6, xn_next=yw, streamk=False)
    bad = 0
    for _ in range(300):
        ops.mlp_fused(xn, pk, b1)                      # another stream-K launch through the same scratch in between
        xg, yg = x0.clone(), torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
        ops.mlp_fused_resid_ln(xn, pk, b1, b2, xg, g, bt, 1e-6, xn_next=yg, streamk=True)
        bad += 0 if (torch.equal(xg, xw) and torch.equal(yg.view(torch.int16), yw.view(torch.int16))) else 1
    assert bad == 0, f"{bad} of 300 stream-K launches differ from the whole-block schedule"


# ------------------------------------------------------------------------------------------ fused eval Mlp (fc1 -> GELU -> fc2, one launch)
@pytest.mark.parametrize("M", [1, 77, 128, 129, 1000, 197 * 8, 32768 + 5, 50432, 35328, 257 * 128, 70001])
@pytest.mark.parametrize("Hd", [1536, 64, 192])
def test_mlp_fused_is_bit_identical_to_the_gemm_pair(ops, M, Hd):
    """tr_mlp_fused_bf16 (timm Mlp of the eval forward, models/topk.py:95) keeps the hidden activation on the CU; by construction -- same
    MFMA, same operand maps, accumulators that start at the bias, K in the same 32-deep steps, same GELU fit, hidden rounded to bf16 at the
    same point -- its output equals tr_gemm_bf16(GELU_BF16) -> tr_gemm_bf16(BF16) BIT FOR BIT, for ragged last blocks, several blocks per
    workgroup, the stream-K schedule (257 blocks: every workgroup but the first and the last continues its neighbour's accumulator), and hidden widths of 2, 6 and 48 steps (the pair needs Hd %% 64 == 0).  Sampled rows are also held against the float64 Mlp on the bf16-rounded operands."""
    if M > 2000 and Hd != 1536:
        pytest.skip("large M only at the model's hidden width")
    D = 384
    rng = _rng(M * 3 + Hd)
    x = _randn(rng, M, D).bfloat16()
    w1, w2 = _randn(rng, Hd, D, scale=0.05).bfloat16(), _randn(rng, D, Hd, scale=0.05).bfloat16()
    b1, b2 = _randn(rng, Hd, scale=0.1), _randn(rng, D, scale=0.1)
    xd, w1d, w2d, b1d, b2d = x.cuda(), w1.cuda(), w2.cuda(), b1.cuda(), b2.cuda()
    h = ops.gemm(xd, w1d, b1d, ops.TR_EPI_GELU_BF16)
    want = ops.gemm(h, w2d, b2d, ops.TR_EPI_BF16)
    pk = ops.mlp_pack(w1d, w2d, b2d)
    for streamk in (True, False):       # beyond 256 blocks: steps dealt evenly with the accumulator handed from workgroup to workgroup / whole blocks
        guard = torch.full((M + 3, D), float("nan"), dtype=torch.bfloat16, device="cuda")        # rows M.. must stay untouched
        got = ops.mlp_fused(xd, pk, b1d, out=guard[:M], streamk=streamk)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), \
            f"streamk={streamk}: {int((got.view(torch.int16) != want.view(torch.int16)).sum())} of {got.numel()} elements differ from the two-launch pair"
        assert torch.isnan(guard[M:].float()).all(), "rows beyond M were written"
    rows = torch.cat([torch.arange(0, min(M, 140)), torch.arange(max(0, M - 140), M), torch.arange(0, M, 1009)]).unique()
    hid = oracle.gelu_erf(x[rows].double() @ w1.double().t() + b1.double())
    ref = _bf(hid.float()).double() @ w2.double().t() + b2.double()
    # the hidden layer is rounded to bf16 (a GELU-fit difference of 2.6e-5 can flip that rounding: one bf16 step of one hidden unit, times a weight)
    torch.testing.assert_close(got[rows.cuda()].cpu().double(), ref, atol=2e-2, rtol=1.2e-2)


def test_mlp_fused_repeated_launches_under_uneven_load(ops):
    """Race screen (found one: an SGPR-soffset buffer store followed by a VALU write of its data registers, profiles/r05_mlp_lab.md): 60
    launches at a shape with two blocks per workgroup, every third one beside a copy kernel on a second stream, each compared with the pair."""
    M, D, Hd = 35328, 384, 1536
    rng = _rng(7)
    x = _randn(rng, M, D).bfloat16().cuda()
    w1, w2 = _randn(rng, Hd, D, scale=0.05).bfloat16().cuda(), _randn(rng, D, Hd, scale=0.05).bfloat16().cuda()
    b1, b2 = _randn(rng, Hd, scale=0.1).cuda(), _randn(rng, D, scale=0.1).cuda()
    want = ops.gemm(ops.gemm(x, w1, b1, ops.TR_EPI_GELU_BF16), w2, b2, ops.TR_EPI_BF16)
    pk = ops.mlp_pack(w1, w2, b2)
    side, junk = torch.cuda.Stream(), torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    out = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    bad = 0
    for it in range(60):
        out.fill_(float("nan"))
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk.add_(1)
        ops.mlp_fused(x, pk, b1, out=out)
        torch.cuda.synchronize()
        bad += 0 if torch.equal(out.view(torch.int16), want.view(torch.int16)) else 1
    assert bad == 0, f"{bad} of 60 launches differ from the two-launch pair"


def test_mlp_fused_rejects_other_widths(ops):
    w1, w2 = torch.zeros(3072, 768, dtype=torch.bfloat16, device="cuda"), torch.zeros(768, 3072, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError, match="does not serve"):
        ops.mlp_pack(w1, w2, torch.zeros(768, device="cuda"))


def test_gemm_operand_roles_not_transposed(ops):
    """A = I-like probe with an ASYMMETRIC weight: catches a swapped row/col map in the accumulator write."""
    M = N = K = 128
    a = torch.eye(M, K)
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) / 256.0   # exactly representable in bf16
    out = ops.gemm(a.cuda().bfloat16(), w.cuda().bfloat16(), torch.zeros(N).cuda(), ops.TR_EPI_F32)
    torch.testing.assert_close(out.cpu(), w.t().contiguous(), atol=0, rtol=0)


def test_patch_embed(ops):
    """im2col + PATCH epilogue + cls/pos rows against oracle.patch_embed/embed_tokens (bf16 operands)."""
    rng = _rng(5)
    B, D = 3, 128
    img = _randn(rng, B, 3, 224, 224)
    w, b = _randn(rng, D, 3, 16, 16, scale=0.02), _randn(rng, D, scale=0.02)
    cls, pos = _randn(rng, 1, 1, D, scale=0.02), _randn(rng, 1, 197, D, scale=0.02)
    want = oracle.embed_tokens(oracle.patch_embed(img, w, b, 16, precision="bf16"), cls, pos)
    cols = ops.im2col(img.cuda(), 16)
    x = torch.empty(B * 197, D, device="cuda")
    ops.gemm(cols, w.reshape(D, -1).cuda().bfloat16(), b.cuda(), ops.TR_EPI_PATCH_F32, out=x, aux=pos.reshape(197, D).cuda(), aux_i=196)
    ops.cls_pos_rows(cls.reshape(-1).cuda(), pos.reshape(197, D).cuda(), x, B, 197, D)
    torch.testing.assert_close(x.cpu().view(B, 197, D), want, atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("B,HW,D,C", [(3, 224, 384, 3), (2, 384, 768, 3), (1, 224, 768, 3), (5, 32, 384, 1), (2, 240, 384, 4)])
def test_patch_embed_fused(ops, B, HW, D, C):
    """The eval forward's one-launch patch embedding (unfold on the way into the LDS + GEMM + bias + pos_embed + the CLS row) against
    the oracle (bf16 operands) and against the three-launch path it replaces (same products; pos_embed joins the sum at a different point)."""
    rng = _rng(50 + HW + D)
    P = (HW // 16) ** 2
    img = _randn(rng, B, C, HW, HW)
    w, b = _randn(rng, D, C, 16, 16, scale=0.02), _randn(rng, D, scale=0.02)
    cls, pos = _randn(rng, 1, 1, D, scale=0.02), _randn(rng, 1, P + 1, D, scale=0.02)
    want = oracle.embed_tokens(oracle.patch_embed(img, w, b, 16, precision="bf16"), cls, pos)
    w16 = w.reshape(D, -1).cuda().bfloat16()
    got = ops.patch_embed(img.cuda(), w16, b.cuda(), cls.reshape(-1).cuda(), pos.reshape(P + 1, D).cuda())
    torch.testing.assert_close(got.cpu(), want, atol=1e-4, rtol=1e-5)
    cols = ops.im2col(img.cuda(), 16)
    x = torch.empty(B * (P + 1), D, device="cuda")
    ops.gemm(cols, w16, b.cuda(), ops.TR_EPI_PATCH_F32, out=x, aux=pos.reshape(P + 1, D).cuda(), aux_i=P)
    ops.cls_pos_rows(cls.reshape(-1).cuda(), pos.reshape(P + 1, D).cuda(), x, B, P + 1, D)
    torch.testing.assert_close(got.view(B * (P + 1), D), x, atol=2e-6, rtol=1e-6)


@pytest.mark.parametrize("M,D", [(394, 384), (33, 768), (7, 128), (5, 192)])
def test_layernorm_with_two_pending_residuals(ops, M, D):
    """The eval executor's lazy norm2: norm2 normalises x + d_attn WITHOUT writing it, the next norm1 takes (x + d_attn) + d_mlp and
    writes the stream once -- bit for bit what the two eager in-place calls produce (outputs and stream)."""
    rng = _rng(700 + M + D)
    x = _randn(rng, M, D) * 2 + 0.3
    d1, d2 = _randn(rng, M, D).bfloat16(), _randn(rng, M, D).bfloat16()
    g, b = 1 + _randn(rng, D, scale=0.1), _randn(rng, D, scale=0.1)
    xe = x.clone().cuda()
    y2_eager = ops.layernorm(xe, g.cuda(), b.cuda(), 1e-6, delta=d1.cuda())              # x += d1
    y1_eager = ops.layernorm(xe, g.cuda(), b.cuda(), 1e-6, delta=d2.cuda())              # x += d2
    xl = x.clone().cuda()
    y2_lazy = ops.layernorm2(xl, g.cuda(), b.cuda(), 1e-6, d1.cuda(), write_x=False)
    assert torch.equal(xl.cpu(), x)                                                      # the stream was not touched
    y1_lazy = ops.layernorm2(xl, g.cuda(), b.cuda(), 1e-6, d1.cuda(), d2.cuda())
    assert torch.equal(y2_lazy, y2_eager) and torch.equal(y1_lazy, y1_eager) and torch.equal(xl, xe)


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,D", [(7, 128), (394, 384), (33, 768), (5, 192), (3, 1024)])
def test_layernorm(ops, M, D):
    rng = _rng(M + D)
    x, g, b = _randn(rng, M, D) * 3 + 0.5, 1 + _randn(rng, D, scale=0.1), _randn(rng, D, scale=0.1)
    want = torch.nn.functional.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-6).float()
    got = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-6)
    assert_close_bf16(got, want, "layernorm", ulps=1.01, abs_floor=1e-4)


@pytest.mark.parametrize("M,D", [(394, 384), (9, 768), (5, 128)])
def test_layernorm_with_pending_residual(ops, M, D):
    """x += delta (bf16 Linear output) written back, then LayerNorm: `x = x + drop_path(...)` folded into the next norm."""
    rng = _rng(M * 3 + D)
    x, d = _randn(rng, M, D), _bf(_randn(rng, M, D, scale=0.3))
    g, b = 1 + _randn(rng, D, scale=0.1), _randn(rng, D, scale=0.1)
    xd = x.clone().cuda()
    got = ops.layernorm(xd, g.cuda(), b.cuda(), 1e-6, delta=d.cuda().bfloat16())
    want_x = x + d                                        # exact in fp32: one add
    assert torch.equal(xd.cpu(), want_x)
    want = torch.nn.functional.layer_norm(want_x.double(), (D,), g.double(), b.double(), 1e-6).float()
    assert_close_bf16(got, want, "add+layernorm", ulps=1.01, abs_floor=1e-4)


def test_layernorm_strided_rows(ops):
    rng = _rng(9)
    B, N, D = 4, 69, 384
    x, g, b = _randn(rng, B, N, D), 1 + _randn(rng, D, scale=0.1), _randn(rng, D, scale=0.1)
    want = torch.nn.functional.layer_norm(x[:, 0].double(), (D,), g.double(), b.double(), 1e-6).float()
    got = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-6, rows=B, ldx=N * D)
    assert_close_bf16(got, want, "layernorm cls rows", ulps=1.01, abs_floor=1e-4)
    d = _bf(_randn(rng, B, N, D, scale=0.2))
    xd = x.clone().cuda()
    got = ops.layernorm(xd, g.cuda(), b.cuda(), 1e-6, rows=B, ldx=N * D, delta=d.cuda().bfloat16(), ldd=N * D)
    want = torch.nn.functional.layer_norm((x[:, 0] + d[:, 0]).double(), (D,), g.double(), b.double(), 1e-6).float()
    assert_close_bf16(got, want, "add+layernorm cls rows", ulps=1.01, abs_floor=1e-4)
    assert torch.equal(xd.cpu()[:, 0], x[:, 0] + d[:, 0]) and torch.equal(xd.cpu()[:, 1:], x[:, 1:])


# ------------------------------------------------------------------------------------------ attention
def _attention_ref(qkv, B, N, H):
    """softmax(q k^T / 8) v in fp64 from bf16-valued qkv; P rounded to bf16 like the kernel (oracle bf16 mode)."""
    q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * 0.125
    p = torch.exp(s - s.amax(-1, keepdim=True))
    l = p.sum(-1, keepdim=True)
    o = (oracle.round_bf16(p.float()).double() @ v) / l
    return o.transpose(1, 2).reshape(B * N, H * 64).float(), (p / l)[:, :, 0, :].float()


@pytest.mark.parametrize("B,N,H", [(2, 197, 6), (3, 138, 2), (2, 97, 3), (2, 68, 2), (1, 224, 1), (2, 33, 2), (1, 7, 1),
                                   (2, 139, 6), (2, 160, 1), (1, 192, 2)])
def test_attention(ops, B, N, H):
    rng = _rng(N * 13 + H)
    qkv = _bf(_randn(rng, B * N, 3 * H * 64, scale=1.5))
    want_o, want_cls = _attention_ref(qkv, B, N, H)
    got_o, got_cls = ops.attention(qkv.cuda().bfloat16(), B, N, H, want_cls=True)
    # P is rounded to bf16 at a value that depends on fp32-vs-fp64 exp in the last bit: allow 2 ulps + small floor
    assert_close_bf16(got_o, want_o, "attention out", ulps=2.0, abs_floor=2e-3)
    torch.testing.assert_close(got_cls.cpu(), want_cls, atol=1e-6, rtol=2e-4)
    got_o2, none = ops.attention(qkv.cuda().bfloat16(), B, N, H, want_cls=False)
    assert none is None and torch.equal(got_o2, got_o)


def test_attention_matches_reference_module(ops, golden_dir):
    """Golden vector from the reference's Attention_TopK (fp32): qkv GEMM + attention + proj through the HIP ops."""
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    x = torch.from_numpy(g["att_x"])
    B, N, D = x.shape
    qkv = ops.gemm(x.reshape(B * N, D).cuda().bfloat16(), torch.from_numpy(g["att_qkv_weight"]).cuda().bfloat16(),
                   torch.from_numpy(g["att_qkv_bias"]).cuda(), ops.TR_EPI_BF16)
    ao, cls_rows = ops.attention(qkv, B, N, 2, want_cls=True)
    out = ops.gemm(ao, torch.from_numpy(g["att_proj_weight"]).cuda().bfloat16(), torch.from_numpy(g["att_proj_bias"]).cuda(),
                   ops.TR_EPI_F32)
    # bf16 operands vs the fp32 reference: tolerance 2e-2 abs on outputs of magnitude ~0.3
    torch.testing.assert_close(out.cpu().view(B, N, D), torch.from_numpy(g["att_out"]), atol=2e-2, rtol=0)
    idx, _, scores = ops.cls_topk(cls_rows, g["att_idx"].shape[1])
    torch.testing.assert_close(scores.cpu(), torch.from_numpy(g["att_scores"]), atol=0, rtol=5e-2)
    overlap = np.mean([len(set(idx[b].tolist()) & set(g["att_idx"][b].tolist())) / idx.shape[1] for b in range(B)])
    assert overlap > 0.9, overlap


# ------------------------------------------------------------------------------------------ Top-K / complement
def test_cls_topk_bit_exact_on_golden_scores(ops, golden_dir):
    """Same scores in -> same indices out, bit exact, against the reference's own torch.topk output."""
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    scores = torch.from_numpy(g["att_scores"])                 # [B,P] from the reference
    B, P = scores.shape
    cls_rows = torch.zeros(B, 1, P + 1)
    cls_rows[:, 0, 1:] = scores                                 # H=1: the head-mean is the identity
    K = g["att_idx"].shape[1]
    idx, compl, sc = ops.cls_topk(cls_rows.cuda(), K, want_compl=True)
    assert torch.equal(sc.cpu(), scores)
    np.testing.assert_array_equal(idx.cpu().numpy().astype(np.int64), g["att_idx"])
    np.testing.assert_array_equal(compl.cpu().numpy().astype(np.int64),
                                  oracle.complement_idx(torch.from_numpy(g["att_idx"]), P).numpy())


@pytest.mark.parametrize("B,H,N,K", [(4, 6, 197, 137), (3, 6, 138, 96), (2, 3, 97, 67), (2, 12, 197, 98), (5, 2, 13, 1),
                                     (2, 2, 13, 11), (2, 6, 577, 144), (1, 1, 1025, 500)])
def test_cls_topk_random(ops, B, H, N, K):
    rng = _rng(N + K)
    rows = torch.from_numpy(rng.random((B, H, N)).astype(np.float32))
    idx, compl, sc = ops.cls_topk(rows.cuda(), K, want_compl=True)
    # head mean: sequential fp32 sum over heads then divide, as the kernel documents
    acc = torch.zeros(B, N - 1)
    for h in range(H):
        acc = acc + rows[:, h, 1:]
    want_scores = acc / H
    assert torch.equal(sc.cpu(), want_scores)
    want_idx = oracle.cls_topk_select(want_scores, K)
    np.testing.assert_array_equal(idx.cpu().numpy(), want_idx.numpy())
    np.testing.assert_array_equal(compl.cpu().numpy(), oracle.complement_idx(want_idx, N - 1).numpy())
    # and against torch's mean (what the reference calls): identical or within 1 ulp
    torch.testing.assert_close(sc.cpu(), rows[:, :, 1:].mean(1), atol=0, rtol=2e-7)


def test_cls_topk_ties_lowest_index_first(ops):
    rows = torch.zeros(1, 1, 9)
    rows[0, 0, 1:] = torch.tensor([0.1, 0.5, 0.5, 0.2, 0.5, 0.1, 0.7, 0.2])
    idx, compl, _ = ops.cls_topk(rows.cuda(), 5, want_compl=True)
    assert idx.cpu().tolist() == [[6, 1, 2, 4, 3]]
    assert compl.cpu().tolist() == [[0, 5, 7]]


# ------------------------------------------------------------------------------------------ gather (+fuse) + LN2
@pytest.mark.parametrize("with_delta", [False, True])
@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("B,N,K,D", [(3, 197, 137, 384), (2, 138, 96, 128), (2, 98, 67, 768), (1, 5, 1, 192)])
def test_gather_layernorm(ops, fuse, with_delta, B, N, K, D):
    rng = _rng(N + K + D + fuse)
    x = _randn(rng, B, N, D)
    delta = _bf(_randn(rng, B, N, D, scale=0.3)) if with_delta else None
    x_in = x
    if with_delta:
        x = x + delta          # what the kernel must gather from (exact fp32 add)
    scores = torch.from_numpy(rng.random((B, N - 1)).astype(np.float32))
    g, b = 1 + _randn(rng, D, scale=0.1), _randn(rng, D, scale=0.1)
    idx = oracle.cls_topk_select(scores, K)
    if fuse:
        want_x, compl = oracle.evit_fuse(x, idx, scores)
    else:
        want_x, compl = oracle.gather_compact(x, idx), None
    got_x, got_y = ops.gather_layernorm(x_in.cuda(), idx.int().cuda(), None if compl is None else compl.int().cuda(),
                                        scores.cuda() if fuse else None, g.cuda(), b.cuda(), 1e-6,
                                        delta=None if delta is None else delta.cuda().bfloat16())
    # gathered rows are pure copies: bit exact; the fused row is a (P-K)-term fp32 sum: 1e-5 relative
    assert torch.equal(got_x.cpu()[:, :K + 1], want_x[:, :K + 1])
    if fuse:
        torch.testing.assert_close(got_x.cpu()[:, K + 1], want_x[:, K + 1], atol=1e-5, rtol=1e-5)
    want_y = torch.nn.functional.layer_norm(got_x.cpu().double(), (D,), g.double(), b.double(), 1e-6).float()
    assert_close_bf16(got_y, want_y, "gather+ln", ulps=1.01, abs_floor=1e-4)


def test_evit_block_against_reference_golden(ops, golden_dir):
    """One full EViT block (evit.py:105-129) through the HIP ops vs the reference's Block_EVIT output (fp32 golden)."""
    from types import SimpleNamespace
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    cfgp = SimpleNamespace(embed_dim=128, depth=1, num_heads=2, mlp_ratio=4, num_classes=4, img_size=224, patch_size=16, in_chans=3)
    p = {k: v.cuda() for k, v in make_params(cfgp, 4321, qkv_gain=6.0).items()}
    x = torch.from_numpy(g["evitblk_x"]).cuda()
    B, N, D = x.shape
    h = x.reshape(B * N, D).clone()
    pre = "blocks.0."
    xn = ops.layernorm(h, p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-6)
    qkv = ops.gemm(xn, p[pre + "attn.qkv.weight"].bfloat16(), p[pre + "attn.qkv.bias"], ops.TR_EPI_BF16)
    ao, cls_rows = ops.attention(qkv, B, N, 2, want_cls=True)
    ops.gemm(ao, p[pre + "attn.proj.weight"].bfloat16(), p[pre + "attn.proj.bias"], ops.TR_EPI_RESID_F32, out=h)
    idx, compl, scores = ops.cls_topk(cls_rows, 98, want_compl=True)
    h2, xn2 = ops.gather_layernorm(h.view(B, N, D), idx, compl, scores, p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-6)
    hid = ops.gemm(xn2.view(-1, D), p[pre + "mlp.fc1.weight"].bfloat16(), p[pre + "mlp.fc1.bias"], ops.TR_EPI_GELU_BF16)
    h2 = h2.view(-1, D)
    ops.gemm(hid, p[pre + "mlp.fc2.weight"].bfloat16(), p[pre + "mlp.fc2.bias"], ops.TR_EPI_RESID_F32, out=h2)
    want_idx = g["evitblk_idx"][:, :-1]
    same = (idx.cpu().numpy() == want_idx).all(axis=1)
    # where bf16 scores reproduce the reference's selection exactly, the block output must match to bf16 accuracy
    got = h2.view(B, 100, D).cpu()
    want = torch.from_numpy(g["evitblk_out"])
    for b in range(B):
        overlap = len(set(idx[b].tolist()) & set(want_idx[b].tolist())) / 98
        assert overlap >= 0.95, overlap
        if same[b]:
            torch.testing.assert_close(got[b], want[b], atol=3e-2, rtol=0)
    torch.testing.assert_close(got[:, 0], want[:, 0], atol=3e-2, rtol=0)    # CLS row is order-independent


# ---------------------------------------------------------------------------------------- ToMe (tome.py)
@pytest.mark.parametrize("B,N,H", [(2, 197, 6), (2, 138, 2), (3, 98, 3), (1, 7, 1)])
def test_attention_proportional(ops, B, N, H):
    """tome.py:48-49: + log(size) on every key's logit.  bf16 kernel vs fp64 on the same bf16 operands."""
    rng = _rng(70 + N)
    qkv = _bf(_randn(rng, B * N, 3 * H * 64, scale=1.5))
    size = torch.from_numpy(rng.integers(1, 6, size=(B, N)).astype(np.float32))
    q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    attn = ((q @ k.transpose(-2, -1)) * 0.125 + size.double().log()[:, None, None, :]).softmax(-1)
    want = (attn @ v).transpose(1, 2).reshape(B * N, H * 64).float()
    got, _ = ops.attention(qkv.bfloat16().cuda(), B, N, H, size=size.cuda())
    # P is rounded to bf16 before P.V (as in the size-free kernel): 2^-8 relative on O(1) values
    torch.testing.assert_close(got.float().cpu(), want, atol=3e-2, rtol=2e-2)
    got32, _ = ops.attention_f32(qkv.cuda(), B, N, H, size=size.cuda())
    torch.testing.assert_close(got32.cpu(), want, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("B,N,H,r", [(3, 197, 6, 59), (2, 197, 2, 98), (2, 138, 3, 41), (2, 98, 6, 1), (1, 5, 1, 2), (2, 8, 2, 3),
                                     (2, 577, 12, 144)])
def test_tome_match_bit_exact(ops, f32, B, N, H, r):
    """bipartite_soft_matching: integer outputs, bit-exact against the oracle on the device's own K (bf16-rounded for the
    bf16 path).  Random gaussians: score gaps are far above fp32 accumulation-order noise, checked below."""
    rng = _rng(1000 + N + r)
    qkv = _randn(rng, B * N, 3 * H * 64)
    if not f32:
        qkv = _bf(qkv)
    k = qkv.reshape(B, N, 3, H, 64)[:, :, 1].permute(0, 2, 1, 3)            # [B,H,N,64]
    metric = k.mean(1)
    unm_w, src_w, dst_w = oracle.tome_match(metric, r)
    # tie-freeness of the oracle's decision inputs (else the test would be testing the tie rule, covered separately)
    m = metric / metric.norm(dim=-1, keepdim=True)
    sc = m[:, ::2] @ m[:, 1::2].transpose(-1, -2)
    top2 = sc[:, 1:].topk(min(2, sc.shape[-1]), dim=-1).values
    if top2.shape[-1] == 2:
        assert (top2[..., 0] - top2[..., 1]).min() > 2e-6
    unm, src, dst = ops.tome_match((qkv if f32 else qkv.bfloat16()).cuda(), B, N, H, r)
    np.testing.assert_array_equal(unm.cpu().numpy(), unm_w.numpy())
    np.testing.assert_array_equal(src.cpu().numpy(), src_w.numpy())
    np.testing.assert_array_equal(dst.cpu().numpy(), dst_w.numpy())


def test_tome_match_ties(ops):
    """Identical tokens: every score ties.  Row argmax -> first index; rank -> lowest index first; CLS never merged."""
    B, N, H, r = 1, 9, 1, 2
    qkv = torch.ones(B * N, 3 * 64)
    unm, src, dst = ops.tome_match(qkv.cuda(), B, N, H, r)
    assert src.cpu().tolist() == [[1, 2]] and dst.cpu().tolist() == [[0, 0]] and unm.cpu().tolist() == [[0, 3, 4]]


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("with_size", [False, True])
@pytest.mark.parametrize("B,N,r,D", [(3, 197, 59, 384), (2, 138, 41, 128), (2, 8, 3, 64), (1, 197, 98, 768)])
def test_tome_merge_layernorm(ops, f32, with_size, B, N, r, D):
    rng = _rng(2000 + N + D)
    x = _randn(rng, B, N, D)
    delta = _randn(rng, B, N, D, scale=0.3)
    if not f32:
        delta = _bf(delta)
    size = torch.from_numpy(rng.integers(1, 5, size=(B, N, 1)).astype(np.float32)) if with_size else None
    na, nb = (N + 1) // 2, N // 2
    perm = np.stack([1 + rng.permutation(na - 1) for _ in range(B)])           # even-set indices 1..na-1, CLS excluded
    src = torch.from_numpy(perm[:, :r].astype(np.int64))
    unm = torch.from_numpy(np.sort(np.concatenate([np.zeros((B, 1), dtype=np.int64), perm[:, r:]], axis=1), axis=1))
    dst = torch.from_numpy(rng.integers(0, max(1, nb // 3), size=(B, r)).astype(np.int64))   # many sources share a destination
    g, b = 1 + _randn(rng, D, scale=0.1), _randn(rng, D, scale=0.05)
    xw, sw = oracle.tome_merge(x + delta, size, unm, src, dst)
    yw = oracle.layer_norm(xw, g, b, 1e-6)
    xo, so, y = ops.tome_merge_layernorm(x.cuda(), (delta if f32 else delta.bfloat16()).cuda(), None if size is None else size[..., 0].cuda(),
                                         unm.int().cuda(), src.int().cuda(), dst.int().cuda(), g.cuda(), b.cuda(), 1e-6, f32=f32)
    assert torch.equal(so.cpu(), sw[..., 0])                                  # sizes are small integers: exact
    # same fp32 operations in the same (edge) order as torch's scatter_add; only fma contraction may differ
    torch.testing.assert_close(xo.cpu(), xw, atol=2e-6, rtol=2e-6)
    if f32:
        torch.testing.assert_close(y.cpu(), yw, atol=1e-5, rtol=1e-5)
    else:
        torch.testing.assert_close(y.float().cpu(), yw, atol=2 * BF16_ULP, rtol=BF16_ULP)


# ---------------------------------------------------------------------------------------- DyViT / SiT pieces
@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("B,N,C", [(3, 197, 384), (2, 138, 128), (2, 5, 64), (1, 97, 768)])
def test_pool_broadcast(ops, f32, B, N, C):
    rng = _rng(300 + N)
    h = _randn(rng, B, N, C)
    if not f32:
        h = _bf(h)
    want = h.clone()
    g = h[:, 1:, C // 2:].double().mean(dim=1, keepdim=True).float() + 1e-6          # patch rows only, eps outside (dyvit.py:117)
    want[:, :, C // 2:] = g if f32 else _bf(g)
    got = ops.pool_broadcast((h if f32 else h.bfloat16()).cuda().reshape(B * N, C), B, N).float().cpu().reshape(B, N, C)
    assert torch.equal(got[:, :, :C // 2], want[:, :, :C // 2])                      # local half untouched
    torch.testing.assert_close(got[:, :, C // 2:], want[:, :, C // 2:], atol=(1e-6 if f32 else BF16_ULP * 0.1), rtol=(1e-5 if f32 else BF16_ULP))


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("M,C", [(3 * 197, 96), (2 * 138, 32), (7, 192), (1000, 48)])
def test_dyvit_score(ops, f32, M, C):
    rng = _rng(400 + C)
    h = _randn(rng, M, C)
    if not f32:
        h = _bf(h)
    w, b = _randn(rng, 2, C, scale=0.3), _randn(rng, 2, scale=0.1)
    want = torch.log_softmax(h.double() @ w.double().t() + b.double(), dim=-1)[:, 0].float()
    got = ops.dyvit_score((h if f32 else h.bfloat16()).cuda(), w.cuda(), b.cuda()).cpu()
    torch.testing.assert_close(got, want, atol=2e-6, rtol=2e-6)


@pytest.mark.parametrize("B,N,K,D,ldl", [(3, 197, 137, 384, 144), (2, 138, 96, 128, 96), (2, 97, 67, 768, 72), (1, 9, 3, 64, 8),
                                         (2, 197, 176, 192, 176)])
def test_sit_merge(ops, B, N, K, D, ldl):
    rng = _rng(500 + N + K)
    logits = _randn(rng, B, N, ldl, scale=2.0)
    x = _randn(rng, B, N, D)
    scale = 1.7
    w = torch.softmax(logits[:, 1:, :K].double() * scale, dim=1).transpose(2, 1)      # sit.py:38
    want = torch.cat([x[:, :1].double(), torch.bmm(w, x[:, 1:].double())], dim=1).float()
    got, soft = ops.sit_merge(logits.cuda(), scale, x.cuda(), K, want_soft=True)
    torch.testing.assert_close(soft.cpu(), w.float(), atol=1e-7, rtol=2e-5)
    torch.testing.assert_close(got.cpu(), want, atol=2e-5, rtol=2e-5)
    assert torch.equal(got[:, 0].cpu(), x[:, 0])
    got2, none = ops.sit_merge(logits.cuda(), scale, x.cuda(), K)
    assert none is None and torch.equal(got2, got)


# ---------------------------------------------------------------------------------------- DPC-KNN
@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("B,N,D,K,k", [(3, 197, 384, 137, 5), (2, 138, 128, 96, 5), (2, 50, 64, 7, 3), (1, 197, 768, 98, 5),
                                       (2, 9, 64, 8, 2)])
def test_dpcknn_cluster(ops, fast, B, N, D, K, k):
    """Integer outputs of a floating-point pipeline: checked against the oracle's fp32 scores/distances up to fp noise --
    centres must be a valid descending top-K of the oracle's scores, assignments the nearest centre wherever the two
    nearest centres differ by more than the noise."""
    rng = _rng(600 + N + D)
    x = _randn(rng, B, N, D)
    noise = torch.from_numpy(rng.random((B, N - 1)).astype(np.float32))
    dist = oracle.dpcknn_distances(x[:, 1:])
    _, _, score = oracle.dpcknn_scores(dist, noise, k)
    # fast = Gram product on MFMA with hi/lo-split bf16 operands (relative error ~2^-16 instead of 2^-24): 20x the tolerances
    tm = 20.0 if fast else 1.0
    centers, assign, score_dev = ops.dpcknn_cluster(x.cuda(), K, noise.cuda(), k, fast_dist=fast)
    centers, assign = centers.cpu().long(), assign.cpu().long()
    scale = float(score.abs().max())
    torch.testing.assert_close(score_dev.cpu(), score, atol=2e-5 * tm * scale, rtol=2e-5 * tm)
    from tests._params import assert_valid_ranking
    assert_valid_ranking(centers.numpy(), score.numpy(), tol=4e-5 * tm * scale)
    want = oracle.dpcknn_assign(dist, centers)                       # the oracle's assignment for the device's centres
    d = torch.gather(dist, 1, centers[:, :, None].expand(B, K, N - 1))
    top2 = d.topk(min(2, K), dim=1, largest=False).values
    decided = (top2[:, -1] - top2[:, 0] > 1e-5 * tm) if K > 1 else torch.ones(B, N - 1, dtype=torch.bool)
    is_center = torch.zeros(B, N - 1, dtype=torch.bool).scatter_(1, centers, True)
    ok = (assign == want) | (~decided & ~is_center)
    assert ok.all(), f"{(~ok).sum().item()} assignments differ beyond fp noise"
    np.testing.assert_array_equal(torch.gather(assign, 1, centers).numpy(), np.broadcast_to(np.arange(K), (B, K)))


@pytest.mark.parametrize("B,N,D,K,k", [(5, 197, 384, 137, 5), (3, 138, 384, 96, 5), (3, 97, 384, 67, 5), (2, 197, 768, 98, 5),
                                       (2, 99, 768, 49, 5), (3, 50, 64, 7, 3), (2, 209, 128, 100, 4), (4, 27, 64, 13, 1)])
def test_dpcknn_one_launch_equals_the_staged_launches(ops, B, N, D, K, k):
    """tr_dpcknn_cluster_fused (distance matrix in LDS, one launch) against the staged launches it replaces (fast_dist = 2), same
    split-bf16 arithmetic: centres, assignments and scores must agree -- bit for bit in the scores of every token whose row needs no
    below-diagonal element... in practice everywhere: the staged matrix is symmetric up to the last bit, so the test allows a score to
    move by one part in 10^6 and a decision to differ only where the staged pipeline's own margin is below that."""
    rng = _rng(640 + N + D)
    x = _randn(rng, B, N, D)
    noise = torch.from_numpy(rng.random((B, N - 1)).astype(np.float32))
    c2, a2, s2 = ops.dpcknn_cluster(x.cuda(), K, noise.cuda(), k, fast_dist=2)
    c1, a1, s1 = ops.dpcknn_cluster(x.cuda(), K, noise.cuda(), k, fast_dist=1)
    torch.testing.assert_close(s1, s2, atol=0, rtol=2e-6)
    if not torch.equal(c1, c2):
        sc = s2.cpu()
        srt = sc.sort(dim=1, descending=True).values
        gap = (srt[:, :-1] - srt[:, 1:]).abs() / srt[:, :-1].abs().clamp_min(1e-30)
        assert float(gap.min()) < 4e-6, "centres differ although no two scores are within rounding of each other"
    else:
        diff = (a1 != a2)
        if diff.any():
            dist = oracle.dpcknn_distances(x[:, 1:])
            d = torch.gather(dist, 1, c2.cpu().long()[:, :, None].expand(B, K, N - 1))
            top2 = d.topk(2, dim=1, largest=False).values
            assert ((top2[:, 1] - top2[:, 0])[diff.cpu()] < 1e-5).all(), "assignments differ beyond a last-bit tie"
    assert lib_supported(N, D, k) == (26 <= N - 1 <= 208 and D % 32 == 0 and k <= 5)


def lib_supported(N, D, k):
    from tokenreduction_amd import _lib
    return bool(_lib.load().tr_dpcknn_fused_supported(N, D, k))


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("B,N,K,D", [(3, 197, 137, 384), (2, 138, 40, 128), (2, 9, 3, 64), (1, 197, 98, 768)])
def test_cluster_merge_layernorm(ops, f32, weighted, B, N, K, D):
    rng = _rng(700 + N + K)
    x = _randn(rng, B, N, D)
    P = N - 1
    assign = np.stack([np.concatenate([np.arange(K), rng.integers(0, K, size=P - K)])[rng.permutation(P)] for _ in range(B)])
    assign = torch.from_numpy(assign.astype(np.int64))                # every cluster non-empty
    sw, sb = _randn(rng, 1, D, scale=0.05), _randn(rng, 1, scale=0.1)
    tw = (x[:, 1:] @ sw.t() + sb).exp() if weighted else None
    xm = torch.cat([x[:, :1], oracle.dpcknn_merge(x[:, 1:], assign, K, tw)], dim=1)
    g, b = 1 + _randn(rng, D, scale=0.1), _randn(rng, D, scale=0.05)
    yw = oracle.layer_norm(xm, g, b, 1e-6)
    xo, y = ops.cluster_merge_layernorm(x.cuda(), assign.int().cuda(), K, g.cuda(), b.cuda(), 1e-6,
                                        sw.cuda() if weighted else None, sb.cuda() if weighted else None, f32=f32)
    torch.testing.assert_close(xo.cpu(), xm, atol=3e-6, rtol=3e-6)
    if f32:
        torch.testing.assert_close(y.cpu(), yw, atol=2e-5, rtol=2e-5)
    else:
        torch.testing.assert_close(y.float().cpu(), yw, atol=2 * BF16_ULP, rtol=BF16_ULP)


# ---------------------------------------------------------------------------------------- ATS
@pytest.mark.parametrize("B,N,H", [(2, 197, 6), (3, 138, 2), (1, 40, 3)])
def test_attention_key_mask(ops, B, N, H):
    """ats.py:117-120: masked keys get exactly zero weight (mask passed as the 1/0 `size` of the attention kernels)."""
    rng = _rng(800 + N)
    qkv = _bf(_randn(rng, B * N, 3 * H * 64, scale=1.5))
    mask = torch.ones(B, N)
    for b in range(B):
        mask[b, N - 1 - rng.integers(3, N // 2):] = 0
    q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    dots = (q @ k.transpose(-2, -1)) * 0.125
    dots = dots.masked_fill(~(mask.bool()[:, None, None, :]), -torch.finfo(torch.float32).max)
    attn = dots.softmax(-1)
    want = (attn @ v).transpose(1, 2).reshape(B * N, H * 64).float()
    got, cls = ops.attention(qkv.bfloat16().cuda(), B, N, H, want_cls=True, size=mask.cuda())
    torch.testing.assert_close(got.float().cpu(), want, atol=3e-2, rtol=2e-2)
    assert (cls.cpu()[mask[:, None, :].expand(B, H, N) == 0] == 0).all()           # exactly zero, like the underflowing softmax
    got32, cls32 = ops.attention_f32(qkv.cuda(), B, N, H, want_cls=True, size=mask.cuda())
    torch.testing.assert_close(got32.cpu(), want, atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(cls32.cpu(), attn[:, :, 0, :].float(), atol=1e-7, rtol=2e-5)


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("B,N,H,K,masked", [(3, 197, 6, 138, False), (2, 138, 6, 97, True), (2, 97, 2, 68, True), (2, 40, 3, 11, False)])
def test_ats_sample(ops, f32, B, N, H, K, masked):
    from tests._params import assert_valid_sampling
    rng = _rng(900 + N + K)
    qkv = _randn(rng, B * N, 3 * H * 64)
    if not f32:
        qkv = _bf(qkv)
    mask = torch.ones(B, N)
    if masked:
        for b in range(B):
            mask[b, N - rng.integers(2, N // 3):] = 0
    cls = torch.softmax(_randn(rng, B, H, N, scale=2.0).masked_fill(mask[:, None, :] == 0, -1e30), dim=-1)
    v = qkv.reshape(B, N, 3, H, 64)[:, :, 2].permute(0, 2, 1, 3)
    steps = oracle.ats_sample_steps(K)
    cdf = oracle.ats_cdf(oracle.ats_scores(cls, v), mask.bool())
    ids, new_mask, cdf_dev = ops.ats_sample(cls.cuda(), (qkv if f32 else qkv.bfloat16()).cuda(), mask.cuda() if masked else None,
                                            steps.cuda(), K, want_cdf=True)
    ids, cdf_dev = ids.cpu().long(), cdf_dev.cpu()
    torch.testing.assert_close(cdf_dev, cdf, atol=2e-6, rtol=0)
    # the sampling decision on the device's own cdf: torch's cdist/argmin/unique give the same ids, bit exact
    want, want_mask = oracle.ats_ids_from_cdf(cdf_dev, steps, pad_to=K)
    np.testing.assert_array_equal(ids.numpy(), want.numpy())
    np.testing.assert_array_equal(new_mask.cpu().numpy(), want_mask.float().numpy())
    assert_valid_sampling((ids[:, 1:] - 1).numpy(), cdf.numpy(), steps.numpy(), tol=5e-4)


def test_ats_gather(ops):
    rng = _rng(77)
    B, N, K, D = 3, 50, 20, 128
    x, ao = _randn(rng, B, N, D), _bf(_randn(rng, B * N, D))
    ids = torch.from_numpy(np.stack([np.concatenate([[0], np.sort(rng.permutation(N - 1)[:K - 4] + 1), [0, 0, 0]]) for _ in range(B)]))
    xo, ao2 = ops.ats_gather(x.cuda(), ao.bfloat16().cuda(), ids.int().cuda())
    assert torch.equal(xo.cpu(), torch.gather(x, 1, ids[:, :, None].expand(B, K, D)))
    assert torch.equal(ao2.float().cpu().reshape(B, K, D), torch.gather(ao.reshape(B, N, D), 1, ids[:, :, None].expand(B, K, D)))


# ---------------------------------------------------------------------------------------- Sinkhorn
def test_rownorm(ops):
    rng = _rng(31)
    x = _randn(rng, 300, 384) * 3
    x[5] = 0                                                     # F.normalize clamps the norm at 1e-12: zero row stays zero
    want = torch.nn.functional.normalize(x, p=2, dim=-1)
    xh, lp = ops.rownorm(x.cuda())
    torch.testing.assert_close(xh.cpu(), want, atol=1e-7, rtol=2e-6)
    torch.testing.assert_close(lp.float().cpu(), want, atol=1e-7, rtol=BF16_ULP)


@pytest.mark.parametrize("B,N,K,ldl,iters", [(3, 197, 137, 144, 3), (2, 138, 96, 96, 3), (2, 97, 67, 72, 5), (1, 9, 3, 8, 1), (1, 30, 7, 8, 0),
                                             (2, 577, 144, 144, 3), (1, 401, 250, 256, 2), (1, 577, 100, 104, 0),      # K*P beyond the LDS: Z stays in global memory ...
                                             (3, 577, 192, 192, 2), (2, 577, 71, 72, 3), (2, 577, 193, 200, 1)])      # ... or, 576 tokens x <= 192 centres, in registers
def test_sinkhorn(ops, B, N, K, ldl, iters):
    rng = _rng(40 + N)
    scores = _randn(rng, B, N, ldl, scale=0.5).clamp(-1, 1)
    want = oracle.sinkhorn_transport(scores[:, 1:, :K].transpose(1, 2).contiguous(), 0.7, iters)      # [B,K,P]
    wt, soft = ops.sinkhorn(scores.cuda(), K, 0.7, iters, want_soft=True)
    torch.testing.assert_close(soft.cpu(), want, atol=1e-6, rtol=2e-5)
    torch.testing.assert_close(wt.cpu()[:, 1:, :K], want.transpose(1, 2), atol=1e-6, rtol=2e-5)


@pytest.mark.parametrize("B,N,K,D,ldl", [(3, 197, 137, 384, 144), (2, 97, 67, 768, 72), (1, 9, 3, 64, 8)])
def test_weighted_merge(ops, B, N, K, D, ldl):
    rng = _rng(50 + N)
    wt = torch.from_numpy(rng.random((B, N, ldl)).astype(np.float32))
    x, src = _randn(rng, B, N, D), _randn(rng, B, N, D)
    want = torch.cat([x[:, :1].double(), torch.bmm(wt[:, 1:, :K].double().transpose(1, 2), src[:, 1:].double())], dim=1).float()
    got = ops.weighted_merge(wt.cuda(), x.cuda(), src.cuda(), K)
    torch.testing.assert_close(got.cpu(), want, atol=5e-5, rtol=2e-5)


# ---------------------------------------------------------------------------------------- K-Medoids
@pytest.mark.parametrize("B,N,H", [(2, 197, 6), (2, 138, 2), (1, 69, 3), (1, 40, 1), (1, 224, 2)])
def test_attention_column_sums(ops, B, N, H):
    """kmedoids.py:240: sum_h sum_q attn[b,h,q,:] from the per-wave partials of both attention kernels."""
    rng = _rng(60 + N)
    qkv = _bf(_randn(rng, B * N, 3 * H * 64, scale=1.5))
    q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    attn = ((q @ k.transpose(-2, -1)) * 0.125).softmax(-1)
    want = attn.sum(dim=1).sum(dim=1).float()                                   # [B,N]
    part = torch.full((B, H, 4, N), float("nan"), device="cuda")
    out, _ = ops.attention(qkv.bfloat16().cuda(), B, N, H, colsum_part=part)
    ref_out, _ = ops.attention(qkv.bfloat16().cuda(), B, N, H)
    assert torch.equal(out, ref_out)                                            # the side output does not change the main one
    torch.testing.assert_close(part.sum(dim=(1, 2)).cpu(), want, atol=2e-4, rtol=2e-5)
    part32 = torch.full((B, H, 4, N), float("nan"), device="cuda")
    ops.attention_f32(qkv.cuda(), B, N, H, colsum_part=part32)
    torch.testing.assert_close(part32.sum(dim=(1, 2)).cpu(), want, atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(want.sum(dim=1).numpy(), H * N, rtol=1e-5)       # every softmax row sums to 1


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("B,N,D,H,K,iters", [(3, 197, 384, 6, 137, 3), (2, 138, 128, 2, 96, 3), (2, 60, 64, 3, 9, 5), (1, 197, 768, 12, 98, 1)])
def test_kmedoids(ops, fast, B, N, D, H, K, iters):
    rng = _rng(3000 + N + K)
    x = _randn(rng, B, N, D)
    part = torch.from_numpy(rng.random((B, H, 4, N)).astype(np.float32))
    w = part.sum(dim=(1, 2))[:, 1:].unsqueeze(2)                                # same fixed summation order as the kernel
    w = torch.zeros(B, N)
    for h in range(H):
        for wv in range(4):
            w = w + part[:, h, wv]
    w = w[:, 1:].unsqueeze(2)
    _, centers_w, assign_w = oracle.kmedoids_fit(x[:, 1:], K, iters, w)
    centers, assign = ops.kmedoids(x.cuda(), part.cuda(), K, iters, fast_dist=fast)
    c_eq = (centers.cpu().long() == centers_w).float().mean().item()
    a_eq = (assign.cpu().long() == assign_w).float().mean().item()
    print(f"\nkmedoids fast={fast} B={B} N={N} K={K}: medoid agreement {c_eq:.4f}, assignment agreement {a_eq:.4f}")
    # identical unless a distance / cost near-tie (fp32 rounding of the matmul-form cdist) moves a decision
    assert c_eq > 0.98 and a_eq > 0.98, (c_eq, a_eq)
    assert int(centers.min()) >= 0 and int(centers.max()) < N - 1 and int(assign.min()) >= 0 and int(assign.max()) < K


# ---------------------------------------------------------------------------------------- long sequences (384^2 inputs)
@pytest.mark.parametrize("B,N,H", [(2, 577, 12), (1, 577, 3), (2, 225, 2), (1, 300, 1), (1, 608, 2)])
def test_attention_long(ops, B, N, H):
    """N > 224: chunked two-pass kernel (with column sums) and online-softmax kernel (without).  Same contract as
    test_attention + CLS row, key bias/mask and column sums."""
    rng = _rng(5000 + N + H)
    qkv = _bf(_randn(rng, B * N, 3 * H * 64, scale=1.5))
    size = torch.from_numpy(rng.integers(1, 5, size=(B, N)).astype(np.float32))
    size[:, N - 7:] = 0                                                          # masked keys (ATS) through the same input
    size[:, 0] = 1
    q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    size_tail = size.clone()
    size_tail[:, N // 2:] = 0                                                    # whole trailing key chunks masked (ATS padding)
    for sz in (None, size, size_tail):
        logits = (q @ k.transpose(-2, -1)) * 0.125
        if sz is not None:
            logits = logits + sz.double().log()[:, None, None, :]
        attn = logits.softmax(-1)
        want = (attn @ v).transpose(1, 2).reshape(B * N, H * 64).float()
        part = torch.full((B, H, 4, N), float("nan"), device="cuda")
        got, cls = ops.attention(qkv.bfloat16().cuda(), B, N, H, want_cls=True, size=None if sz is None else sz.cuda(), colsum_part=part)
        torch.testing.assert_close(got.float().cpu(), want, atol=3e-2, rtol=2e-2)
        torch.testing.assert_close(cls.cpu(), attn[:, :, 0, :].float(), atol=2e-6, rtol=2e-3)
        torch.testing.assert_close(part.sum(dim=(1, 2)).cpu(), attn.sum(dim=1).sum(dim=1).float(), atol=5e-4, rtol=2e-3)
        # without column sums the online-softmax kernel runs (key chunks of 128): same contract, its own rounding points
        got2, cls2 = ops.attention(qkv.bfloat16().cuda(), B, N, H, want_cls=True, size=None if sz is None else sz.cuda())
        torch.testing.assert_close(got2.float().cpu(), want, atol=3e-2, rtol=2e-2)
        torch.testing.assert_close(cls2.cpu(), attn[:, :, 0, :].float(), atol=2e-6, rtol=2e-3)
        got3, _ = ops.attention(qkv.bfloat16().cuda(), B, N, H, size=None if sz is None else sz.cuda())
        assert torch.equal(got3, got2)                                           # the CLS side output does not change the main one
        rel = ((got2.float() - got.float()).norm() / got.float().norm()).item()
        assert rel < 4e-3, rel                                                   # with a key bias AND column sums: the two-pass kernel
        if sz is None:
            assert torch.equal(got2, got)                                        # without a bias both calls take the online-softmax kernel


@pytest.mark.parametrize("softmax", [True, False])
@pytest.mark.parametrize("B,N,K,D,ldl", [(3, 197, 137, 384, 144), (2, 138, 96, 128, 96), (2, 97, 67, 768, 72), (1, 9, 3, 64, 8),
                                         (2, 197, 176, 192, 176), (1, 577, 144, 768, 144)])
def test_softassign_merge_fast(ops, softmax, B, N, K, D, ldl):
    """MFMA soft merge (SiT / PatchMerger / Sinkhorn in the bf16 executor): hi/lo-split bf16 operands, fp32 accumulation --
    within 1e-4 of fp64, against 4e-3 for plain bf16 operands."""
    rng = _rng(8000 + N + K)
    x, src = _randn(rng, B, N, D), _randn(rng, B, N, D)
    scale = 1.3
    if softmax:
        logits = _randn(rng, B, N, ldl, scale=2.0)
        w = torch.softmax(logits[:, 1:, :K].double() * scale, dim=1)                # [B,P,K]
    else:
        logits = torch.from_numpy(rng.random((B, N, ldl)).astype(np.float32))
        w = logits[:, 1:, :K].double()
    want = torch.cat([x[:, :1].double(), torch.bmm(w.transpose(1, 2), src[:, 1:].double())], dim=1).float()
    got, soft = ops.softassign_merge_fast(logits.clone().cuda(), scale, x.cuda(), K, apply_softmax=softmax, want_soft=True, src=src.cuda())
    ref_scale = float(want[:, 1:].abs().max())
    torch.testing.assert_close(got.cpu(), want, atol=1e-4 * ref_scale, rtol=1e-4)
    assert torch.equal(got[:, 0].cpu(), x[:, 0])
    if softmax:
        torch.testing.assert_close(soft.cpu(), w.transpose(1, 2).float(), atol=1e-7, rtol=2e-5)
    else:
        assert soft is None


@pytest.mark.parametrize("B,N,H", [(2, 197, 6), (2, 138, 2), (1, 40, 3), (1, 224, 1), (2, 577, 2), (1, 257, 1)])
def test_attention_with_policy(ops, B, N, H):
    """a11: DyViT's training-time attention (dyvit.py:39-67) against the oracle's restatement of softmax_with_policy.  Beyond 224
    tokens (384 x 384 inputs) the bf16 path is the online-softmax kernel; the fp32 twin holds 256."""
    rng = _rng(9000 + N)
    qkv = _bf(_randn(rng, B * N, 3 * H * 64, scale=1.5))
    policy = torch.from_numpy((rng.random((B, N)) > 0.4).astype(np.float32))
    policy[:, 0] = 1.0                                               # CLS is always kept (dyvit.py:226)
    q, k, v = qkv.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    attn = oracle.dyvit_softmax_with_policy((q @ k.transpose(-2, -1)) * 0.125, policy.unsqueeze(-1))
    want = (attn @ v).transpose(1, 2).reshape(B * N, H * 64)
    got = ops.attention_policy(qkv.bfloat16().cuda(), policy.cuda(), B, N, H)
    torch.testing.assert_close(got.float().cpu(), want, atol=3e-2, rtol=2e-2)
    if N > 256:
        return
    got32 = ops.attention_policy(qkv.cuda(), policy.cuda(), B, N, H)
    torch.testing.assert_close(got32.cpu(), want, atol=2e-5, rtol=2e-5)
    # policy of all ones = plain softmax up to the eps smoothing
    ones = torch.ones(B, N)
    plain, _ = ops.attention_f32(qkv.cuda(), B, N, H)
    torch.testing.assert_close(ops.attention_policy(qkv.cuda(), ones.cuda(), B, N, H).cpu(), plain.cpu(), atol=1e-5, rtol=1e-5)


# ---------------------------------------------------------------------------------------- shape sweeps (edge sizes)
def test_attention_every_sequence_length(ops):
    """All token counts 2..224 in steps of 5 plus the block edges (31, 32, 33, 63, 64, 65, ...): masking of the padded keys,
    partially filled query blocks, CLS row."""
    rng = _rng(123)
    H, B = 2, 2
    ns = sorted(set(list(range(2, 225, 5)) + [31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 159, 160, 161, 191, 192, 193, 223, 224]))
    worst = 0.0
    for N in ns:
        qkv = _bf(_randn(rng, B * N, 3 * H * 64, scale=1.2))
        q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
        attn = ((q @ k.transpose(-2, -1)) * 0.125).softmax(-1)
        want = (attn @ v).transpose(1, 2).reshape(B * N, H * 64).float()
        got, cls = ops.attention(qkv.bfloat16().cuda(), B, N, H, want_cls=True)
        err = (got.float().cpu() - want).abs().max().item()
        worst = max(worst, err)
        assert err < 3e-2, (N, err)
        torch.testing.assert_close(cls.cpu(), attn[:, :, 0, :].float(), atol=2e-6, rtol=2e-3)
    print(f"attention N sweep ({len(ns)} lengths): worst abs error {worst:.2e}")


def test_cls_topk_every_k(ops):
    """K from 1 to P at P = 196 (and a small P): descending order, complement, bit-exact against the oracle."""
    rng = _rng(321)
    for N in (197, 9):
        P = N - 1
        rows = torch.from_numpy(rng.random((2, 3, N)).astype(np.float32))
        scores = rows[:, :, 1:].sum(1) / 3
        for K in sorted(set([1, 2, P // 2, P - 1, P] + list(range(1, P + 1, 13)))):
            idx, compl, sc = ops.cls_topk(rows.cuda(), K, want_compl=True)
            want = oracle.cls_topk_select(sc.cpu(), K)
            np.testing.assert_array_equal(idx.cpu().numpy(), want.numpy())
            np.testing.assert_array_equal(compl.cpu().numpy(), oracle.complement_idx(want, P).numpy())
            torch.testing.assert_close(sc.cpu(), scores, atol=1e-7, rtol=1e-6)


def test_tome_match_parity_sweep(ops):
    """Odd/even token counts and r from 1 to the (N-1)//2 cap."""
    rng = _rng(555)
    for N in (3, 4, 5, 16, 17, 100, 101, 196, 197):
        for r in sorted(set([1, (N - 1) // 4 or 1, (N - 1) // 2])):
            qkv = _randn(rng, 2 * N, 3 * 2 * 64)
            k = qkv.reshape(2, N, 3, 2, 64)[:, :, 1].permute(0, 2, 1, 3)
            unm_w, src_w, dst_w = oracle.tome_match(k.mean(1), r)
            unm, src, dst = ops.tome_match(qkv.cuda(), 2, N, 2, r)
            np.testing.assert_array_equal(unm.cpu().numpy(), unm_w.numpy())
            np.testing.assert_array_equal(src.cpu().numpy(), src_w.numpy())
            np.testing.assert_array_equal(dst.cpu().numpy(), dst_w.numpy())


@pytest.mark.parametrize("B,N,H", [(1, 785, 2), (2, 1025, 1)])
def test_attention_beyond_the_lds_limit(ops, B, N, H):
    """Without column sums the online-softmax kernel has no sequence-length limit (448^2 and 512^2 inputs)."""
    rng = _rng(5100 + N)
    qkv = _bf(_randn(rng, B * N, 3 * H * 64, scale=1.5))
    q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    attn = ((q @ k.transpose(-2, -1)) * 0.125).softmax(-1)
    want = (attn @ v).transpose(1, 2).reshape(B * N, H * 64).float()
    got, cls = ops.attention(qkv.bfloat16().cuda(), B, N, H, want_cls=True)
    torch.testing.assert_close(got.float().cpu(), want, atol=3e-2, rtol=2e-2)
    torch.testing.assert_close(cls.cpu(), attn[:, :, 0, :].float(), atol=2e-6, rtol=2e-3)
    part = torch.full((B, H, 4, N), float("nan"), device="cuda")                 # column sums: second pass over the keys
    got2, _ = ops.attention(qkv.bfloat16().cuda(), B, N, H, colsum_part=part)
    assert torch.equal(got2, got)
    torch.testing.assert_close(part.sum(dim=(1, 2)).cpu(), attn.sum(dim=1).sum(dim=1).float(), atol=5e-4, rtol=2e-3)
    with pytest.raises(Exception):                                               # ... but not together with a key bias
        ops.attention(qkv.bfloat16().cuda(), B, N, H, size=torch.ones(B, N, device="cuda"), colsum_part=part)
