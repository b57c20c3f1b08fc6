"""RoI-pool backward micro-benchmark at RC-Net's shapes: python3 tools/bench_roi.py [scale_index]   (0: 1/2 res 32 ch, 1: 1/4 res 64 ch, ...)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from riders_amd import engine

dev = torch.device("cuda:0")
engine.set_compute_dtype("bf16")
B, IH, IW, PH0, PW0, K = 8, 256, 512, 240, 100, 30
cfgs = [(2, 32), (4, 64), (8, 128), (16, 256), (32, 256)]
sel = [int(a) for a in sys.argv[1:]] or range(len(cfgs))
rs = np.random.RandomState(3)
rois = []
for b in range(B):
    for k in range(K):
        x0 = rs.randint(0, IW - PW0); y0 = rs.randint(0, IH - PH0 + 1)
        rois.append([b, x0, y0, x0 + PW0, y0 + PH0])
rois = torch.tensor(rois, dtype=torch.float32, device=dev)
for i in sel:
    s, C = cfgs[i]
    H, W, PH, PW = IH // s, IW // s, PH0 // s, PW0 // s
    x = torch.randn(B, H, W, C, device=dev).to(torch.bfloat16)
    tape = engine.Tape(); tape.mark(x)
    with engine._active(tape):
        out = engine.roi_pool(x, rois, (PH, PW), 1.0 / s)
    g = torch.randn_like(out)
    for rep in range(3):
        t2 = engine.Tape(); t2.mark(x)
        with engine._active(t2):
            o = engine.roi_pool(x, rois, (PH, PW), 1.0 / s)
            t2.grads[id(o)] = g
            torch.cuda.synchronize(); t0 = time.perf_counter()
            t2.backward()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
    elems = out.numel()
    print("roi bwd 1/%d: map %dx%dx%d, %d RoIs -> %dx%d: %.3f ms (%d M elements, %.0f GB/s of dout+argmax)" % (s, H, W, C, rois.shape[0], PH, PW, dt * 1e3, elems // 1000000, elems * 6 / dt / 1e9))
