"""Probe (round 6): is a small device-to-device hipMemcpyAsync (torch's contiguous same-dtype clone / copy_) ordered against a hipGraph launch that
follows it on the same stream?  300 small tensors are cloned, then a captured graph overwrites the originals; every clone must hold the OLD values.
Variants: tensors allocated inside the capture's private pool or before it; clones made with torch (memcpy) or with a kernel (mul by 1)."""
import sys
import torch

dev = torch.device("cuda:0")


def run(pool_alloc, kernel_clone, thread):
    n = 300
    sizes = [128 + 64 * (i % 7) for i in range(n)]
    g = torch.cuda.CUDAGraph()
    src = [torch.full((s,), 1.0, device=dev) for s in sizes]
    torch.cuda.synchronize()
    if pool_alloc:
        with torch.cuda.graph(g):
            bufs = [torch.empty(s, device=dev) for s in sizes]
            for b, s_ in zip(bufs, src):
                torch.add(s_, 0.0, out=b)
            for s_ in src:
                s_.add_(1.0)
    else:
        bufs = [torch.zeros(s, device=dev) for s in sizes]
        with torch.cuda.graph(g):
            for b, s_ in zip(bufs, src):
                torch.add(s_, 0.0, out=b)
            for s_ in src:
                s_.add_(1.0)
    bad = [0]

    def body():
        g.replay()      # bufs = k
        for it in range(20):
            expect = float(it + 1)
            clones = [(b * 1.0) if kernel_clone else b.clone() for b in bufs]
            g.replay()
            torch.cuda.synchronize()
            for c in clones:
                if not bool((c == expect).all()):
                    bad[0] += 1
    if thread:
        import threading
        th = threading.Thread(target=body); th.start(); th.join()
    else:
        body()
    return bad[0]


for pool_alloc in (False, True):
    for kernel_clone in (False, True):
        for thread in (False, True):
            print("buffers %-22s clone by %-7s %-12s -> %d corrupted clones of 6000" % (
                "inside the capture" if pool_alloc else "before the capture", "kernel" if kernel_clone else "memcpy", "other thread" if thread else "main thread",
                run(pool_alloc, kernel_clone, thread)), flush=True)
