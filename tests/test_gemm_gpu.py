"""The update path's float32 MFMA GEMM kernel (lhw_debug_gemm -> gemm_f32_kernel, csrc/lhw_ppo.hip) against torch float64 on the
shapes lhw_ppo_grad issues -- both block-tile sizes, every operand layout, ragged edges (K not a multiple of 16, M / N not a
multiple of the tile), the fused epilogues (bias + ReLU, ReLU-derivative mask) and the split-K path with the bias gradient's
column sums taken from the same operand tiles."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _gemm(a_kc, b_kc, wt, M, N, K, A, B, C, bias=None, relu=0, mask=None, k_chunk=0, part=None, colsum=None, colsum_out=None):
    from learninghumanoidwalking_amd import _lib
    L = _lib.lib()
    _lib.check(L.lhw_debug_gemm(int(a_kc), int(b_kc), wt, M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), _p(C), C.stride(0), _p(bias), relu,
                                _p(mask), mask.stride(0) if mask is not None else 0, k_chunk, _p(part), _p(colsum), _p(colsum_out), None))


@pytest.mark.parametrize("wt", [1, 2])
@pytest.mark.parametrize("M,N,K", [(1000, 256, 40), (4099, 256, 256), (777, 12, 256), (130, 130, 20)])
def test_forward_layout_bias_relu(wt, M, N, K):
    import torch
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    C = torch.full((M, (N + 3) // 4 * 4), 7.0, device="cuda")
    _gemm(True, True, wt, M, N, K, A, W, C, bias=b, relu=1)
    ref = torch.relu(A.double() @ W.double().t() + b.double())
    assert (C[:, :N].double() - ref).abs().max().item() < 2e-5
    assert (C[:, N:] == 7.0).all()


@pytest.mark.parametrize("wt", [1, 2])
@pytest.mark.parametrize("M,N,K", [(3000, 256, 12), (2049, 256, 256)])
def test_backward_activation_layout_with_mask(wt, M, N, K):
    import torch
    g = torch.Generator(device="cuda").manual_seed(N + K)
    dY = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(K, N, device="cuda", generator=g) / K ** 0.5      # stored [K][N]
    h = torch.randn(M, N, device="cuda", generator=g)
    C = torch.zeros(M, N, device="cuda")
    _gemm(True, False, wt, M, N, K, dY, W, C, mask=h)
    ref = (dY.double() @ W.double()) * (h > 0)
    assert (C.double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("wt", [1, 2])
@pytest.mark.parametrize("M,N,K,kc", [(256, 256, 5000, 512), (12, 256, 4097, 512), (256, 40, 3000, 512), (1, 256, 2000, 256)])
def test_weight_gradient_split_k_with_fused_bias_gradient(wt, M, N, K, kc):
    import torch
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    Mp = (M + 3) // 4 * 4
    dY = torch.randn(K, Mp, device="cuda", generator=g)          # A stored [K][M]
    X = torch.randn(K, N, device="cuda", generator=g)
    nz = (K + kc - 1) // kc
    part = torch.full((nz, M * N), float("nan"), device="cuda")
    cpart = torch.full((nz, M), float("nan"), device="cuda")
    C0 = torch.randn(M, N, device="cuda", generator=g)
    b0 = torch.randn(M, device="cuda", generator=g)
    C, bsum = C0.clone(), b0.clone()
    _gemm(False, False, wt, M, N, K, dY, X, C, k_chunk=kc, part=part, colsum=cpart, colsum_out=bsum)
    ref = C0.double() + dY[:, :M].double().t() @ X.double()
    refb = b0.double() + dY[:, :M].double().sum(0)
    scale = K ** 0.5
    assert (C.double() - ref).abs().max().item() < 2e-5 * scale
    assert (bsum.double() - refb).abs().max().item() < 2e-5 * scale
    # deterministic: a second run gives the same bits
    C2, b2 = C0.clone(), b0.clone()
    _gemm(False, False, wt, M, N, K, dY, X, C2, k_chunk=kc, part=part, colsum=cpart, colsum_out=b2)
    assert torch.equal(C, C2) and torch.equal(bsum, b2)


@pytest.mark.parametrize("K,kc", [(32768, 512), (1000, 512), (4099, 128), (37, 512), (65536, 512)])
def test_wide_weight_gradient_matches_float64(K, kc):
    """wgrad_wide_kernel (csrc/lhw_ppo.hip): dW2 = dh2^T h1 and db2 = colsum(dh2) per k slice, operands straight from global memory
    (reference rl/algos/ppo.py:387-396: the hidden layer's share of loss.backward()), against float64 slice by slice; ragged row
    counts (a last chunk shorter than the 16-row register buffer), run-to-run bitwise determinism."""
    import torch
    from learninghumanoidwalking_amd import _lib
    L = _lib.lib()
    H = 256
    g = torch.Generator(device="cuda").manual_seed(K)
    dh2 = torch.randn(K, H, device="cuda", generator=g) * 0.1
    h1 = torch.relu(torch.randn(K, H, device="cuda", generator=g))
    ns = (K + kc - 1) // kc
    p = lambda t: ctypes.c_void_p(t.data_ptr())

    def run():
        part = torch.full((ns, H, H), 7.0, device="cuda")
        cs = torch.full((ns, H), 7.0, device="cuda")
        _lib.check(L.lhw_debug_wgrad_wide(p(dh2), p(h1), K, kc, p(part), p(cs), None))
        torch.cuda.synchronize()
        return part, cs

    (pa, ca), (pb, cb) = run(), run()
    assert torch.equal(pa, pb) and torch.equal(ca, cb)
    for z in range(ns):
        a64, b64 = dh2[z * kc:(z + 1) * kc].double(), h1[z * kc:(z + 1) * kc].double()
        ref = a64.t() @ b64
        assert float((pa[z].double() - ref).abs().max()) < 2e-5 * (float(ref.abs().max()) + 1.0)
        assert float((ca[z].double() - a64.sum(0)).abs().max()) < 2e-5 * (float(a64.sum(0).abs().max()) + 1.0)


@pytest.mark.parametrize("layout,M,N,K", [("fwd", 1000, 256, 40), ("fwd", 777, 12, 256), ("bwd", 2049, 256, 256), ("bwd", 3000, 256, 12),
                                          ("dw", 256, 256, 5000), ("dw", 12, 256, 4097), ("dw", 256, 40, 3000)])
def test_fp16_operand_mode_equals_half_rounded_operands(layout, M, N, K):
    """wt = 16: both operands rounded to fp16 while staged, fp16 MFMA, float32 accumulate -- every layout of the update."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + K)
    h = lambda x: x.half().double()
    if layout == "fwd":
        A = torch.randn(M, (K + 3) // 4 * 4, device="cuda", generator=g)[:, :K]
        A = torch.nn.functional.pad(A, (0, (-K) % 4)).contiguous()
        W = torch.randn(N, A.shape[1], device="cuda", generator=g) / K ** 0.5
        if K % 4:
            A[:, K:] = 0
        b = torch.randn(N, device="cuda", generator=g)
        C = torch.zeros(M, (N + 3) // 4 * 4, device="cuda")
        _gemm(True, True, 16, M, N, K, A, W, C, bias=b, relu=1)
        ref = torch.relu(h(A[:, :K]) @ h(W[:, :K]).t() + b.double())
        assert (C[:, :N].double() - ref).abs().max().item() < 1e-4
    elif layout == "bwd":
        dY = torch.randn(M, K, device="cuda", generator=g)
        W = torch.randn(K, N, device="cuda", generator=g) / K ** 0.5
        msk = torch.randn(M, N, device="cuda", generator=g)
        C = torch.zeros(M, N, device="cuda")
        _gemm(True, False, 16, M, N, K, dY, W, C, mask=msk)
        ref = (h(dY) @ h(W)) * (msk > 0)
        assert (C.double() - ref).abs().max().item() < 1e-4
    else:
        Mp = (M + 3) // 4 * 4
        dY = torch.randn(K, Mp, device="cuda", generator=g)
        X = torch.randn(K, N, device="cuda", generator=g)
        kc = 512
        nz = (K + kc - 1) // kc
        part = torch.full((nz, M * N), float("nan"), device="cuda")
        cpart = torch.full((nz, M), float("nan"), device="cuda")
        C = torch.zeros(M, N, device="cuda"); bsum = torch.zeros(M, device="cuda")
        _gemm(False, False, 16, M, N, K, dY, X, C, k_chunk=kc, part=part, colsum=cpart, colsum_out=bsum)
        ref = h(dY[:, :M]).t() @ h(X)
        assert (C.double() - ref).abs().max().item() < 2e-5 * K ** 0.5 * 4
        assert (bsum.double() - h(dY[:, :M]).sum(0)).abs().max().item() < 2e-5 * K ** 0.5 * 4


@pytest.mark.parametrize("R,Dp,O,Op", [(1000, 40, 12, 12), (32768, 40, 12, 12), (4099, 36, 10, 12), (300, 44, 1, 4), (65536, 40, 12, 12)])
def test_fused_skinny_weight_gradients_match_float64(R, Dp, O, Op):
    """wgrad_skinny_kernel (csrc/lhw_ppo.hip): dW1 = dh1^T x, db1, dW3 = dy^T h2, db3 of one network in one K-streaming launch
    (reference rl/algos/ppo.py:387-396: the first / last layer's share of loss.backward()), against float64; ragged row counts,
    the critic's single output, accumulation into the outputs, run-to-run bitwise determinism."""
    import torch
    from learninghumanoidwalking_amd import _lib
    L = _lib.lib()
    H = 256
    g = torch.Generator(device="cuda").manual_seed(R)
    dh1 = torch.randn(R, H, device="cuda", generator=g) * 0.1
    h2 = torch.relu(torch.randn(R, H, device="cuda", generator=g))
    x = torch.randn(R, Dp, device="cuda", generator=g)
    dy = torch.zeros(R, Op, device="cuda")
    dy[:, :O] = torch.randn(R, O, device="cuda", generator=g) * 0.01
    kc = max(128, (((R + 255) // 256) + 15) // 16 * 16)
    ns = (R + kc - 1) // kc
    scratch = torch.empty(ns * (H * Dp + H + O * H + O), device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())

    def run():
        out = [torch.full((H, Dp), 0.5, device="cuda"), torch.zeros(H, device="cuda"), torch.zeros(O, H, device="cuda"), torch.full((O,), -1.0, device="cuda")]
        _lib.check(L.lhw_debug_wgrad_skinny(H, Dp, O, Op, p(dh1), p(x), Dp, p(dy), p(h2), R, p(out[0]), p(out[1]), p(out[2]), p(out[3]), p(scratch), None))
        torch.cuda.synchronize()
        return out

    a, b = run(), run()
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    d64, x64, y64, h64 = dh1.double(), x.double(), dy[:, :O].double(), h2.double()
    ref = [d64.t() @ x64 + 0.5, d64.sum(0), y64.t() @ h64, y64.sum(0) - 1.0]
    for u, r in zip(a, ref):
        scale = float(r.abs().max()) + 1.0
        assert float((u.double() - r).abs().max()) < 2e-5 * scale * max(1.0, (R / 32768) ** 0.5)


@pytest.mark.parametrize("layout,M,N,K", [("fwd", 1000, 256, 40), ("fwd", 777, 12, 256), ("fwd", 515, 256, 36), ("bwd", 2049, 256, 256), ("bwd", 3000, 256, 12),
                                          ("dw", 256, 256, 5000), ("dw", 12, 256, 4097), ("dw", 256, 40, 3000)])
def test_fp16_storage_mode_loads_and_stores_fp16(layout, M, N, K):
    """Round 6 (BASELINE config 5 as an fp16 pipeline): gemm_h_kernel with operands that LIVE in fp16 in HBM (wt = 16 + bits: A / B / C /
    mask stored as fp16) -- the update's activations and back-propagated gradients -- against the same half-rounded reference as the
    float32-storage mode: loading an fp16 buffer is loading the rounded value."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(7 * M + 3 * N + K)
    h = lambda x: x.half().double()
    if layout == "fwd":          # x / h (fp16, row stride padded to 8) times float32 weights -> fp16 activations
        Kp = (K + 7) // 8 * 8
        A = torch.zeros(M, Kp, device="cuda", dtype=torch.float16)
        A[:, :K] = torch.randn(M, K, device="cuda", generator=g).half()
        W = torch.zeros(N, (K + 3) // 4 * 4, device="cuda")
        W[:, :K] = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
        b = torch.randn(N, device="cuda", generator=g)
        Np = (N + 3) // 4 * 4
        C = torch.full((M, Np), 7.0, device="cuda", dtype=torch.float16)
        _gemm(True, True, 16 + 1 + 4, M, N, K, A, W, C, bias=b, relu=1)
        ref = torch.relu(A[:, :K].double() @ h(W[:, :K]).t() + b.double())
        assert (C[:, :N].double() - ref).abs().max().item() < 1e-4 + ref.abs().max().item() * 2 ** -10      # the output is rounded to fp16
        assert (C[:, N:] == 7.0).all()
        C32 = torch.zeros(M, Np, device="cuda")
        _gemm(True, True, 16 + 1, M, N, K, A, W, C32, bias=b, relu=1)                                           # float32 output (the read-out layer)
        assert (C32[:, :N].double() - ref).abs().max().item() < 1e-4
    elif layout == "bwd":        # dh = (dy W) * (h > 0): fp16 dy / mask / output, float32 weights
        Kp = (K + 7) // 8 * 8
        dY = torch.zeros(M, Kp, device="cuda", dtype=torch.float16)
        dY[:, :K] = torch.randn(M, K, device="cuda", generator=g).half()
        W = torch.randn(K, N, device="cuda", generator=g) / K ** 0.5
        msk = torch.randn(M, N, device="cuda", generator=g).half()
        C = torch.zeros(M, N, device="cuda", dtype=torch.float16)
        _gemm(True, False, 16 + 1 + 4 + 8, M, N, K, dY, W, C, mask=msk)
        ref = (dY[:, :K].double() @ h(W)) * (msk > 0)
        assert (C.double() - ref).abs().max().item() < 1e-4 + ref.abs().max().item() * 2 ** -10
    else:                        # dW = dh^T h, db = colsum(dh): both operands fp16, float32 split-K partials
        Mp = (M + 7) // 8 * 8
        dY = torch.zeros(K, Mp, device="cuda", dtype=torch.float16)
        dY[:, :M] = torch.randn(K, M, device="cuda", generator=g).half()
        X = torch.randn(K, N, device="cuda", generator=g).half()
        kc = 512
        nz = (K + kc - 1) // kc
        part = torch.full((nz, M * N), float("nan"), device="cuda")
        cpart = torch.full((nz, M), float("nan"), device="cuda")
        C = torch.zeros(M, N, device="cuda"); bsum = torch.zeros(M, device="cuda")
        _gemm(False, False, 16 + 1 + 2, M, N, K, dY, X, C, k_chunk=kc, part=part, colsum=cpart, colsum_out=bsum)
        ref = dY[:, :M].double().t() @ X.double()
        assert (C.double() - ref).abs().max().item() < 2e-5 * K ** 0.5 * 4
        assert (bsum.double() - dY[:, :M].double().sum(0)).abs().max().item() < 2e-5 * K ** 0.5 * 4
