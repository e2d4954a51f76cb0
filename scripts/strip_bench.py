"""Isolated timing of the MLP strip kernels against the per-layer GEMM sequence they replace (one stream, nothing else running).
usage: python scripts/strip_bench.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from learninghumanoidwalking_amd import _lib
from tests.test_emu_mlp_strip import make_case

R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = _lib.lib()
Dp, O, Op = 40, 12, 12
c = make_case(R=R, Dp=Dp, O=O, Op=Op, seed=0)
d = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in c.items()}
p = lambda t: t.data_ptr()
h1 = torch.zeros(R, 256, device="cuda"); h2 = torch.zeros(R, 256, device="cuda"); y = torch.zeros(R, Op, device="cuda")
dh2 = torch.zeros(R, 256, device="cuda"); dh1 = torch.zeros(R, 256, device="cuda")
wt = torch.zeros((Dp + 256 + Op) * 256, device="cuda")
z = None


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


def fwd_strip():
    _lib.check(L.lhw_debug_mlp_strip_forward(256, Dp, O, Op, p(d["w1"]), p(d["b1"]), p(d["w2"]), p(d["b2"]), p(d["w3"]), p(d["b3"]), p(d["x"]), Dp, R, p(h1), p(h2), p(y), p(wt), None))


def fwd_gemm():
    _lib.check(L.lhw_debug_gemm(1, 1, 1, R, 256, Dp, p(d["x"]), Dp, p(d["w1"]), Dp, p(h1), 256, p(d["b1"]), 1, z, 0, 0, z, z, z, None))
    _lib.check(L.lhw_debug_gemm(1, 1, 1, R, 256, 256, p(h1), 256, p(d["w2"]), 256, p(h2), 256, p(d["b2"]), 1, z, 0, 0, z, z, z, None))
    _lib.check(L.lhw_debug_gemm(1, 1, 1, R, O, 256, p(h2), 256, p(d["w3"]), 256, p(y), Op, p(d["b3"]), 0, z, 0, 0, z, z, z, None))


def bwd_strip():
    _lib.check(L.lhw_debug_mlp_strip_backward(256, O, Op, p(d["w2"]), p(d["w3"]), p(d["dy"]), R, p(h1), p(h2), p(dh2), p(dh1), None))


def bwd_gemm():
    _lib.check(L.lhw_debug_gemm(1, 0, 1, R, 256, O, p(d["dy"]), Op, p(d["w3"]), 256, p(dh2), 256, z, 0, p(h2), 256, 0, z, z, z, None))
    _lib.check(L.lhw_debug_gemm(1, 0, 1, R, 256, 256, p(dh2), 256, p(d["w2"]), 256, p(dh1), 256, z, 0, p(h1), 256, 0, z, z, z, None))


flops_f = 2.0 * R * (Dp * 256 + 256 * 256 + 256 * O)
flops_b = 2.0 * R * (O * 256 + 256 * 256)
for name, f, fl in (("forward  strip", fwd_strip, flops_f), ("forward  gemm x3", fwd_gemm, flops_f), ("backward strip", bwd_strip, flops_b), ("backward gemm x2", bwd_gemm, flops_b)):
    t = timeit(f)
    print(f"rows {R}  {name:18s} {t:8.1f} us   {fl / t / 1e6:6.1f} TF/s")
