"""The library's bf16 GEMM on the step's dominant shapes, one at a time on an idle GPU (events, median of 30):
TFLOP/s of each and the fraction of the box's own bare-MFMA ceiling (tools/mfma_peak.hip, pass it as argv[1] in
TFLOP/s; default 1250).  Shapes at cfg3: n_sub = 9980 rows, G = 3000 genes padded to 3072, H C = 2048."""
import sys, numpy as np, torch
peak = float(sys.argv[1]) if len(sys.argv) > 1 else 1250.0
dev = "cuda"
n, G, Gp, HC = 9980, 3000, 3072, 2048
def bench(fn, flop, name):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = float(np.median(ts)) * 1e-3
    print(f"{name:58s} {t*1e6:7.1f} us  {flop/t/1e12:7.0f} TFLOP/s  {100*flop/t/1e12/peak:5.1f} % of the bare-MFMA rate  {100*flop/t/2.5e15:5.1f} % of 2.5 PF")
x = torch.randn(n, Gp, device=dev).bfloat16(); W1 = torch.randn(HC, Gp, device=dev).bfloat16()
h = torch.randn(n, HC, device=dev).bfloat16(); W2 = torch.randn(HC, HC, device=dev).bfloat16()
g = torch.randn(n, HC, device=dev).bfloat16()
out32 = torch.empty(HC, G, device=dev)
bench(lambda: torch.nn.functional.linear(x, W1), 2.0 * n * Gp * HC, "layer 1 forward   [9980 x 3072] . [3072 x 2048]")
bench(lambda: torch.mm(g.t(), x[:, :G], out_dtype=torch.float32, out=out32), 2.0 * n * G * HC, "layer 1 weight gradient [2048 x 9980] . [9980 x 3000] -> fp32")
bench(lambda: torch.nn.functional.linear(h, W2), 2.0 * n * HC * HC, "layer 2 forward   [9980 x 2048] . [2048 x 2048]")
bench(lambda: g @ W2, 2.0 * n * HC * HC, "layer 2 input gradient [9980 x 2048] . [2048 x 2048]")
bench(lambda: torch.mm(g.t(), h, out_dtype=torch.float32), 2.0 * n * HC * HC, "layer 2 weight gradient [2048 x 9980] . [9980 x 2048] -> fp32")
