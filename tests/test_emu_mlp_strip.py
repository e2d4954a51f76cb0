"""The LDS-resident MLP strip kernels (csrc/lhw_mlp_strip.hip) on the SIMT emulator: the HIP source, compiled for the CPU with
the f32 MFMA emulated as its fmaf chain, against a float64 numpy evaluation of the same three layers -- slab staging, the
k-major LDS hand-off between the layers, the MFMA operand / accumulator lane maps, the K-split read-out, partial last slab.
tests/test_mlp_strip_gpu.py is the GPU twin (and compares with the per-layer GEMM path)."""
import ctypes

import numpy as np
import pytest


def _ptr(a):
    return a.ctypes.data


def make_case(R, Dp=40, O=12, Op=12, seed=0, H=256):
    rs = np.random.default_rng(seed)
    f = np.float32
    w1 = (rs.normal(size=(H, Dp)) / np.sqrt(Dp)).astype(f); b1 = (rs.normal(size=H) * 0.1).astype(f)
    w2 = (rs.normal(size=(H, H)) / np.sqrt(H)).astype(f); b2 = (rs.normal(size=H) * 0.1).astype(f)
    w3 = np.zeros((Op, H), f); w3[:O] = (rs.normal(size=(O, H)) / np.sqrt(H)).astype(f)
    b3 = np.zeros(Op, f); b3[:O] = (rs.normal(size=O) * 0.1).astype(f)
    x = rs.normal(size=(R, Dp)).astype(f)
    dy = np.zeros((R, Op), f); dy[:, :O] = rs.normal(size=(R, O)).astype(f)
    return dict(w1=w1, b1=b1, w2=w2, b2=b2, w3=w3, b3=b3, x=x, dy=dy, H=H, Dp=Dp, O=O, Op=Op, R=R)


def reference(c):
    d = np.float64
    h1 = np.maximum(c["x"].astype(d) @ c["w1"].astype(d).T + c["b1"], 0)
    h2 = np.maximum(h1 @ c["w2"].astype(d).T + c["b2"], 0)
    y = h2 @ c["w3"].astype(d).T + c["b3"]
    return h1, h2, y


def reference_backward(c, h1, h2):
    d = np.float64
    dh2 = (c["dy"][:, :c["O"]].astype(d) @ c["w3"][:c["O"]].astype(d)) * (h2 > 0)
    dh1 = (dh2 @ c["w2"].astype(d)) * (h1 > 0)
    return dh2, dh1


def run_forward(L, c, sentinel=7.0):
    R, H, Op = c["R"], c["H"], c["Op"]
    h1 = np.full((R + 3, H), sentinel, np.float32); h2 = np.full((R + 3, H), sentinel, np.float32); y = np.full((R + 3, Op), sentinel, np.float32)
    wt = np.zeros((c["Dp"] + 256 + Op) * 256, np.float32)
    rc = L.lhw_debug_mlp_strip_forward(H, c["Dp"], c["O"], Op, _ptr(c["w1"]), _ptr(c["b1"]), _ptr(c["w2"]), _ptr(c["b2"]), _ptr(c["w3"]),
                                       _ptr(c["b3"]), _ptr(c["x"]), c["Dp"], R, _ptr(h1), _ptr(h2), _ptr(y), _ptr(wt), None)
    assert rc == 0
    return h1, h2, y


def test_strip_forward_and_backward_on_the_emulator():
    from tests import emu
    L = emu.lib()
    c = make_case(R=100, Dp=40, O=12, Op=12, seed=1)
    h1, h2, y = run_forward(L, c)
    r1, r2, ry = reference(c)
    R = c["R"]
    assert (h1[R:] == 7.0).all() and (h2[R:] == 7.0).all() and (y[R:] == 7.0).all(), "rows beyond R must not be written"
    np.testing.assert_allclose(h1[:R], r1, rtol=0, atol=2e-5)
    np.testing.assert_allclose(h2[:R], r2, rtol=0, atol=5e-5)
    np.testing.assert_allclose(y[:R, :c["O"]], ry[:, :c["O"]], rtol=0, atol=5e-5)
    # backward from the kernel's own activations (the masks are exact then)
    dh2 = np.full((R + 3, 256), 7.0, np.float32); dh1 = np.full((R + 3, 256), 7.0, np.float32)
    rc = L.lhw_debug_mlp_strip_backward(256, c["O"], c["Op"], _ptr(c["w2"]), _ptr(c["w3"]), _ptr(c["dy"]), R, _ptr(h1), _ptr(h2), _ptr(dh2), _ptr(dh1), None)
    assert rc == 0
    g2, g1 = reference_backward(c, h1[:R].astype(np.float64), h2[:R].astype(np.float64))
    assert (dh2[R:] == 7.0).all() and (dh1[R:] == 7.0).all()
    np.testing.assert_allclose(dh2[:R], g2, rtol=0, atol=5e-5)
    np.testing.assert_allclose(dh1[:R], g1, rtol=0, atol=1e-4)


def test_both_workgroup_shapes_return_the_same_bits(monkeypatch):
    """32-row / 8-wave slabs (rollout inference) and 64-row / 4-wave slabs (the update): same fmaf chains, same partial sums."""
    from tests import emu
    L = emu.lib()
    c = make_case(R=70, Dp=40, O=12, Op=12, seed=4)
    outs = {}
    for shape in ("small", "big"):
        monkeypatch.setenv("LHW_DEBUG_STRIP_SHAPE", shape)
        outs[shape] = run_forward(L, c)
    for a, b in zip(outs["small"], outs["big"]):
        np.testing.assert_array_equal(a, b)
    r1, r2, ry = reference(c)
    np.testing.assert_allclose(outs["big"][2][:70, :12], ry[:, :12], rtol=0, atol=5e-5)


def test_strip_critic_shape_single_output_on_the_emulator():
    from tests import emu
    L = emu.lib()
    c = make_case(R=64, Dp=36, O=1, Op=4, seed=2)
    h1, h2, y = run_forward(L, c)
    r1, r2, ry = reference(c)
    np.testing.assert_allclose(y[:64, 0], ry[:, 0], rtol=0, atol=5e-5)
    assert (y[:64, 1:] == 7.0).all(), "pad columns of the read-out are left alone"


def test_strip_refuses_other_widths():
    from tests import emu
    L = emu.lib()
    z = np.zeros(4, np.float32)
    rc = L.lhw_debug_mlp_strip_forward(128, 40, 12, 12, _ptr(z), _ptr(z), _ptr(z), _ptr(z), _ptr(z), _ptr(z), _ptr(z), 40, 1, _ptr(z), _ptr(z), _ptr(z), _ptr(z), None)
    assert rc != 0


def run_bits_round_trip(L, c, ptr=_ptr, alloc=None):
    """forward with mask bits, then backward (a) from the bits alone and (b) from the activations: (a) == (b) bit for bit, and the forward
    outputs equal the launch without bits.  `alloc(shape, dtype, fill)` / `ptr` let the GPU twin run the same steps on device buffers."""
    R, H, Op = c["R"], c["H"], c["Op"]
    alloc = alloc or (lambda shape, dt, fill: np.full(shape, fill, dt))
    nw = (R + 63) // 64 * 512
    wt = alloc(((c["Dp"] + 256 + Op) * 256,), np.float32, 0)
    outs = []
    for with_bits in (False, True):
        h1, h2, y = alloc((R, H), np.float32, 7.0), alloc((R, H), np.float32, 7.0), alloc((R, Op), np.float32, 7.0)
        b1, b2 = alloc((nw,), np.uint32, 0xFFFFFFFF), alloc((nw,), np.uint32, 0xFFFFFFFF)
        rc = L.lhw_debug_mlp_strip_forward_bits(H, c["Dp"], c["O"], Op, ptr(c["w1"]), ptr(c["b1"]), ptr(c["w2"]), ptr(c["b2"]), ptr(c["w3"]), ptr(c["b3"]),
                                                ptr(c["x"]), c["Dp"], R, ptr(h1), ptr(h2), ptr(y), ptr(wt), ptr(b1) if with_bits else None,
                                                ptr(b2) if with_bits else None, None)
        assert rc == 0
        outs.append((h1, h2, y, b1, b2))
    grads = []
    for mode in ("acts", "bits"):
        h1, h2, y, b1, b2 = outs[1]
        dh2, dh1 = alloc((R, H), np.float32, 7.0), alloc((R, H), np.float32, 7.0)
        rc = L.lhw_debug_mlp_strip_backward_bits(H, c["O"], Op, ptr(c["w2"]), ptr(c["w3"]), ptr(c["dy"]), R, ptr(h1) if mode == "acts" else None,
                                                 ptr(h2) if mode == "acts" else None, ptr(dh2), ptr(dh1), ptr(b1) if mode == "bits" else None,
                                                 ptr(b2) if mode == "bits" else None, None)
        assert rc == 0
        grads.append((dh2, dh1))
    return outs, grads


@pytest.mark.parametrize("R,Dp,O,Op", [(100, 40, 12, 12), (64, 36, 1, 4)])
def test_relu_mask_bits_round_trip_on_the_emulator(R, Dp, O, Op):
    from tests import emu
    L = emu.lib()
    c = make_case(R=R, Dp=Dp, O=O, Op=Op, seed=5)
    outs, grads = run_bits_round_trip(L, c)
    for a, b in zip(outs[0][:3], outs[1][:3]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(grads[0][0], grads[1][0])
    np.testing.assert_array_equal(grads[0][1], grads[1][1])
    g2, g1 = reference_backward(c, outs[1][0].astype(np.float64), outs[1][1].astype(np.float64))
    np.testing.assert_allclose(grads[1][0], g2, rtol=0, atol=5e-5)
    np.testing.assert_allclose(grads[1][1], g1, rtol=0, atol=1e-4)
    # one bit per positive activation (rows beyond R of a ragged last slab are computed from zero inputs and carry bits too)
    used = (R + 63) // 64 * 512
    for layer in (0, 1):
        pop = int(np.unpackbits(np.asarray(outs[1][3 + layer]).view(np.uint8)).sum())
        pos = int((np.asarray(outs[1][layer]) > 0).sum())
        assert pop == pos + (outs[1][3 + layer].size - used) * 32 if R % 64 == 0 else pop >= pos
