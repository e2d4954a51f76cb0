"""Phase timeline of the MLP strip kernels from a -DLHW_STRIP_CLOCK build (scripts/build_variant.sh clock -DLHW_STRIP_CLOCK; run with
LHW_LIB=<that library>): per-wave wall-clock stamps at the phase boundaries -> when the workgroups start, how long each phase takes, how much of
it is waiting at a barrier.  usage: python scripts/strip_clock.py [rows] [fwd|bwd]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from learninghumanoidwalking_amd import _lib
from tests.test_emu_mlp_strip import make_case

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
L = _lib.lib()
Dp, O, Op = 40, 12, 12
c = make_case(R=R, Dp=Dp, O=O, Op=Op, seed=0)
d = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in c.items()}
p = lambda t: t.data_ptr()
h1 = torch.zeros(R, 256, device="cuda"); h2 = torch.zeros(R, 256, device="cuda"); y = torch.zeros(R, Op, device="cuda")
dh2 = torch.zeros(R, 256, device="cuda"); dh1 = torch.zeros(R, 256, device="cuda")
wt = torch.zeros((Dp + 256 + Op) * 256, device="cuda")


def fwd():
    _lib.check(L.lhw_debug_mlp_strip_forward(256, Dp, O, Op, p(d["w1"]), p(d["b1"]), p(d["w2"]), p(d["b2"]), p(d["w3"]), p(d["b3"]), p(d["x"]), Dp, R, p(h1), p(h2), p(y), p(wt), None))


def bwd():
    _lib.check(L.lhw_debug_mlp_strip_backward(256, O, Op, p(d["w2"]), p(d["w3"]), p(d["dy"]), R, p(h1), p(h2), p(dh2), p(dh1), None))


f = fwd if which == "fwd" else bwd
names = {"fwd": ["stage x", "L1 products", "barrier", "L1 epilogue", "barrier", "L2 products", "barrier", "L2 epilogue", "barrier", "read-out products", "barrier",
                 "read-out sum + store"],
         "bwd": ["stage dy", "dh2 products", "barrier", "dh2 epilogue", "barrier", "dh1 products", "dh1 epilogue"]}[which]
for _ in range(3):
    f()
torch.cuda.synchronize()
f()
torch.cuda.synchronize()
buf = np.zeros(2048 * 4 * 16, np.uint64)
L.lhw_debug_strip_clock_read.argtypes = [ctypes.c_void_p]
assert L.lhw_debug_strip_clock_read(buf.ctypes.data) == 0
nb = min(2048, (R + 63) // 64)
t = buf.reshape(2048, 4, 16)[:nb].astype(np.int64)
t0 = t[:, :, 0].min()
ns = 10.0   # 100 MHz
n = len(names)
start = (t[:, :, 0] - t0) * ns / 1e3
end = (t[:, :, n] - t0) * ns / 1e3
print(f"{which} rows {R}: {nb} workgroups; first stamp -> last stamp {end.max():.1f} us")
print(f"  workgroup start (us after the first): median {np.median(start):.1f}  p90 {np.percentile(start, 90):.1f}  max {start.max():.1f};  started after 5 us: {(start.min(1) > 5).sum()} groups")
print(f"  workgroup duration (us): median {np.median(end - start):.1f}  min {(end - start).min():.1f}  max {(end - start).max():.1f}")
first = start.min(1) < 5
for k, nm in enumerate(names):
    dur = (t[:, :, k + 1] - t[:, :, k]) * ns / 1e3
    print(f"  {nm:22s} mean {dur.mean():6.2f} us   first-wave groups {dur[first].mean():6.2f}   later groups {dur[~first].mean() if (~first).any() else float('nan'):6.2f}   max {dur.max():6.2f}")
