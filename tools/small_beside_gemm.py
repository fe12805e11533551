"""How long do the SVGP encoder's small launches take beside the first GAT layer's GEMM?  Stream A replays that GEMM (library
bf16, 9980 x 3072 -> 2048) back to back; stream B (high priority) runs one small launch at a time between two events.
Prints the median / p90 duration of each small launch alone and beside the GEMM."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd import ops

dev = "cuda:0"
torch.manual_seed(0)
xg = torch.randn(9984, 3072, device=dev).bfloat16()
wg = torch.randn(2048, 3072, device=dev).bfloat16()
h1 = torch.randn(512, 256, device=dev)
h2 = torch.randn(512, 64, device=dev)
bn1 = torch.nn.BatchNorm1d(256).to(dev)
bn2 = torch.nn.BatchNorm1d(64).to(dev)
lb1, lb2 = torch.zeros(256, device=dev), torch.zeros(64, device=dev)
W2 = torch.randn(64, 256, device=dev)
Wfc, bfc = torch.randn(20, 64, device=dev), torch.zeros(20, device=dev)
a1 = torch.randn(512, 256, device=dev)
a2 = torch.randn(512, 64, device=dev)
small = {
    "bn_act 512x256": lambda: ops.bn_act(h1, lb1, bn1, 0.01),
    "bn_act 512x64": lambda: ops.bn_act(h2, lb2, bn2, 0.01),
    "hidden map 512x256->64 (library fp32)": lambda: torch.nn.functional.linear(a1, W2),
    "SVGP_fc 512x64->20 (library fp32)": lambda: torch.addmm(bfc, a2, Wfc.t()),
}
A = torch.cuda.Stream()
B = torch.cuda.Stream(priority=-1)


def measure(fn, busy, reps=60):
    out = []
    stop = False
    for r in range(reps):
        if busy:
            with torch.cuda.stream(A):
                for _ in range(3):
                    torch.nn.functional.linear(xg, wg)
        with torch.cuda.stream(B):
            time.sleep(0.00015 if busy else 0)          # land inside the GEMMs
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(B); fn(); e1.record(B)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    out = np.sort(np.array(out[5:]))
    return out[len(out) // 2], out[int(len(out) * 0.9)]


with torch.no_grad():
    for _ in range(5):
        torch.nn.functional.linear(xg, wg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        torch.nn.functional.linear(xg, wg)
    torch.cuda.synchronize()
    print(f"GEMM alone: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us")
    for name, fn in small.items():
        for _ in range(3):
            fn()
        a = measure(fn, False)
        b = measure(fn, True)
        print(f"{name:42s} alone {a[0]:6.1f} us (p90 {a[1]:6.1f})   beside the GEMM {b[0]:6.1f} us (p90 {b[1]:6.1f})")
