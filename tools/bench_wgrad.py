"""Micro-benchmark of one conv layer's kernels through the C ABI (used for rocprofv3 --pmc passes)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riders_amd import engine
from riders_amd.engine import _desc, _p, L, _stream
dev = torch.device("cuda:0")
N, H, W = [int(v) for v in os.environ.get("RD_NHW", "240,240,100").split(",")]
Cin, Cout = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 16
dt_name = sys.argv[3] if len(sys.argv) > 3 else "bf16"
which = sys.argv[4] if len(sys.argv) > 4 else "wgrad"
tdt = torch.bfloat16 if dt_name == "bf16" else torch.float32
dt = engine.rd_of(torch.empty(1, dtype=tdt))
x = torch.randn((N, H, W, Cin), device=dev).to(tdt)
dy = torch.randn((N, H, W, Cout), device=dev).to(tdt)
w = torch.nn.Parameter(torch.randn((Cout, Cin, 3, 3), device=dev))
d = _desc(dt, N, H, W, Cin, 0, False, H, W, Cout, 3, 3, 1, 1, 1, H, W, 0, 0.0, Cout)
lib = L(); st = _stream(x)
ws = torch.empty(lib.rd_conv_wgrad_workspace_bytes(ctypes.byref(d)) // 4, dtype=torch.float32, device=dev)
dw = torch.empty_like(w)
wp = engine.packed_weight(w, 0, dt)
y = torch.empty((N, H, W, Cout), dtype=tdt, device=dev)
if which == "dgrad":   # data gradient of a (Cin -> Cout) layer whose input is a 2-way concat: dy (Cout ch) -> dx1 | dx2 (Cin/2 each)
    dd = _desc(dt, N, H, W, Cout, 0, False, H, W, Cin, 3, 3, 1, 1, 1, H, W, 0, 0.0, Cin // 2)
    wpd = engine.packed_weight(w, 1, dt)
    dx1 = torch.empty((N, H, W, Cin // 2), dtype=tdt, device=dev); dx2 = torch.empty_like(dx1)
def run():
    if which == "dgrad":
        return lib.rd_conv_fwd(ctypes.byref(dd), _p(dy), None, _p(wpd), None, _p(dx1), _p(dx2), None, st)
    if which == "wgrad":
        return lib.rd_conv_wgrad(ctypes.byref(d), _p(x), None, _p(dy), _p(ws), _p(dw), 0, st)
    return lib.rd_conv_fwd(ctypes.byref(d), _p(x), None, _p(wp), None, _p(y), None, None, st)
for _ in range(3): assert run() == 0
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): run()
torch.cuda.synchronize(); dtm = (time.time() - t0) / 10
fl = 2.0 * N * H * W * Cin * Cout * 9
print("%s NHW=%s Cin=%d Cout=%d %s: %.3f ms  %.1f TFLOP/s  %.1f GB/s(alg)" % (which, os.environ.get("RD_NHW", "-"), Cin, Cout, dt_name, dtm * 1e3, fl / dtm / 1e12, (x.numel() + dy.numel()) * x.element_size() / dtm / 1e9))
