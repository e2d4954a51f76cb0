"""Time the update's GEMM shapes (lhw_debug_gemm) for both block-tile sizes: python scripts/gemm_bench.py [rows]"""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from learninghumanoidwalking_amd import _lib

L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
H, Dp, Op = 256, 40, 12
KS = int(sys.argv[2]) if len(sys.argv) > 2 else 128   # slice length of the skinny weight-gradient GEMMs
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
dev = "cuda"
x = torch.randn(B, Dp, device=dev); h = torch.randn(B, H, device=dev); h2 = torch.randn(B, H, device=dev); y = torch.randn(B, Op, device=dev)
W1 = torch.randn(H, Dp, device=dev); W2 = torch.randn(H, H, device=dev); W3 = torch.randn(Op, H, device=dev); b = torch.randn(H, device=dev)
out = torch.zeros(B, H, device=dev); outy = torch.zeros(B, Op, device=dev)
nz = (B + 511) // 512
part = torch.zeros(nz * H * H, device=dev); part2 = torch.zeros(2 * nz * H * H, device=dev); cpart = torch.zeros(4 * nz * H, device=dev); dW = torch.zeros(H, H, device=dev); db = torch.zeros(H, device=dev)
dW1 = torch.zeros(H, Dp, device=dev); dW3 = torch.zeros(Op, H, device=dev)
cases = [
    ("L1  fwd  [B,40]x[256,40]^T", 2.0 * B * H * Dp, lambda wt: L.lhw_debug_gemm(1, 1, wt, B, H, Dp, p(x), Dp, p(W1), Dp, p(out), H, p(b), 1, None, 0, 0, None, None, None, None)),
    ("L2  fwd  [B,256]x[256,256]^T", 2.0 * B * H * H, lambda wt: L.lhw_debug_gemm(1, 1, wt, B, H, H, p(h), H, p(W2), H, p(out), H, p(b), 1, None, 0, 0, None, None, None, None)),
    ("L3  fwd  [B,256]x[12,256]^T", 2.0 * B * H * Op, lambda wt: L.lhw_debug_gemm(1, 1, wt, B, Op, H, p(h), H, p(W3), H, p(outy), Op, None, 0, None, 0, 0, None, None, None, None)),
    ("dh2 bwd  [B,12]x[12,256] mask", 2.0 * B * H * Op, lambda wt: L.lhw_debug_gemm(1, 0, wt, B, H, Op, p(y), Op, p(W3), H, p(out), H, None, 0, p(h2), H, 0, None, None, None, None)),
    ("dh1 bwd  [B,256]x[256,256] mask", 2.0 * B * H * H, lambda wt: L.lhw_debug_gemm(1, 0, wt, B, H, H, p(h), H, p(W2), H, p(out), H, None, 0, p(h2), H, 0, None, None, None, None)),
    ("dW2      [B,256]^T x [B,256] +colsum", 2.0 * B * H * H, lambda wt: L.lhw_debug_gemm(0, 0, wt, H, H, B, p(h), H, p(h2), H, p(dW), H, None, 0, None, 0, 512, p(part), p(cpart), p(db), None)),
    ("dW2 kc256  +colsum", 2.0 * B * H * H, lambda wt: L.lhw_debug_gemm(0, 0, wt, H, H, B, p(h), H, p(h2), H, p(dW), H, None, 0, None, 0, 256, p(part2), p(cpart), p(db), None)),
    ("dW2 kc1024 +colsum", 2.0 * B * H * H, lambda wt: L.lhw_debug_gemm(0, 0, wt, H, H, B, p(h), H, p(h2), H, p(dW), H, None, 0, None, 0, 1024, p(part), p(cpart), p(db), None)),
    ("dW1      [B,256]^T x [B,40] +colsum", 2.0 * B * H * Dp, lambda wt: L.lhw_debug_gemm(0, 0, wt, H, Dp, B, p(h), H, p(x), Dp, p(dW1), Dp, None, 0, None, 0, KS, p(part), p(cpart), p(db), None)),
    ("dW3      [B,12]^T x [B,256] +colsum", 2.0 * B * H * Op, lambda wt: L.lhw_debug_gemm(0, 0, wt, Op, H, B, p(y), Op, p(h), H, p(dW3), H, None, 0, None, 0, KS, p(part), p(cpart), p(db), None)),
]
print(f"rows {B}")
for name, flops, fn in cases:
    line = f"{name:40s}"
    for wt in (1, 2):
        for _ in range(3):
            _lib.check(fn(wt))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(wt)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f"  wt{wt}: {us:7.1f} us {flops / us / 1e6:6.1f} TF/s"
    print(line)

# the wide weight gradient without LDS (wgrad_wide_kernel): dW2 + db2 partials of 512-row slices, against the LDS-staged GEMM (partials
# + reduction); warm = the operands stay in the caches between launches, cold = 1 GB written between launches (as in the update, where the
# forward pass wrote h1 several hundred MB of traffic before the weight gradients read it)
nsw = (B + 511) // 512
pw = torch.zeros(nsw * H * H, device=dev); cw = torch.zeros(nsw * H, device=dev)
fnw = lambda: L.lhw_debug_wgrad_wide(p(h), p(h2), B, 512, p(pw), p(cw), None)
fng = lambda: L.lhw_debug_gemm(0, 0, 1, H, H, B, p(h), H, p(h2), H, p(dW), H, None, 0, None, 0, 512, p(part), p(cpart), p(db), None)
flush = torch.empty(256 * 1024 * 1024, device=dev)


def timed(fn, cold, n=20):
    ts = []
    for _ in range(n + 3):
        if cold:
            flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(fn()); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts[3:])[n // 2]


import os
for nm, fn in (("dW2 LDS-staged GEMM 64x64 (+ reduction)", fng), (f"dW2 wide, KS={os.environ.get('LHW_WGRAD_WIDE_KS', '32')} (partials only)", fnw)):
    tw, tc = timed(fn, False), timed(fn, True)
    print(f"{nm:44s} warm {tw:7.1f} us {2.0 * B * H * H / tw / 1e6:6.1f} TF/s   cold {tc:7.1f} us {2.0 * B * H * H / tc / 1e6:6.1f} TF/s")

# the fused skinny weight gradients (dW1 + db1 + dW3 + db3 in one K-streaming launch) against the two split-K GEMMs above
kc = max(128, (((B + 255) // 256) + 15) // 16 * 16)
ns = (B + kc - 1) // kc
scr = torch.zeros(ns * (H * Dp + H + Op * H + Op), device=dev); db3 = torch.zeros(Op, device=dev)
fn = lambda: L.lhw_debug_wgrad_skinny(H, Dp, Op, Op, p(h), p(x), Dp, p(y), p(h2), B, p(dW1), p(db), p(dW3), p(db3), p(scr), None)
for _ in range(3):
    _lib.check(fn())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"{'dW1 + dW3 fused (wgrad_skinny) +colsums':40s}  {us:7.1f} us {2.0 * B * H * (Dp + Op) / us / 1e6:6.1f} TF/s   ({2 * B * H * 4 / us / 1e6:.2f} TB/s of activations)")
