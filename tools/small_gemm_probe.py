"""The decoder's small weight-gradient GEMMs ([N x 512] x [512 x K], fp32): the library's pick against a split-K
batched form (8 slices of the 512 batch rows, then a sum)."""
import torch, time
dev = "cuda:0"
def t(fn, reps=200):
    for _ in range(5): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps // 10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K) in [(512, 256, 64), (512, 64, 20), (512, 20, 512), (512, 256, 3000)]:
    g_ = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
    a = t(lambda: g_.t() @ x)
    for S in (4, 8, 16):
        b = t(lambda: torch.bmm(g_.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K)).sum(0))
        print(f"M={M} N={N} K={K}: library {a:6.1f} us   split-{S} {b:6.1f} us", flush=True)
    c = t(lambda: (x.t() @ g_))
    print(f"   transposed problem (x^T g): {c:6.1f} us")
