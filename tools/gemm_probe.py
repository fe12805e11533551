"""Layout probe for the GAT dense maps at the cfg3 shape (bf16): forward NT, dX as NN vs NT (pre-transposed W),
dW as TN, and whether mm can write fp32 directly."""
import torch, time
dev = "cuda"
n, K, N = 10000, 2048, 2048
x = torch.randn(n, K, device=dev, dtype=torch.bfloat16)
W = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
G = torch.randn(n, N, device=dev, dtype=torch.bfloat16)
Wt = W.t().contiguous()
def t(f, name):
    for _ in range(5): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): f()
    e.record(); torch.cuda.synchronize()
    print(f"{name:40s} {s.elapsed_time(e)/50*1e3:8.1f} us")
t(lambda: torch.nn.functional.linear(x, W), "fwd  x @ W^T (NT)")
t(lambda: G @ W, "dX   G @ W (NN)")
t(lambda: torch.nn.functional.linear(G, Wt), "dX   linear(G, W^T copy) (NT)")
t(lambda: W.t().contiguous(), "transpose copy of W")
t(lambda: G.t() @ x, "dW   G^T @ x (TN)")
Gt = G.t().contiguous(); xt = x.t().contiguous()
t(lambda: torch.nn.functional.linear(Gt, xt), "dW   linear(G^T copy, x^T copy) (NT)")
try:
    r = torch.mm(G.t(), x, out_dtype=torch.float32)
    print("mm out_dtype fp32 ok", r.dtype)
    t(lambda: torch.mm(G.t(), x, out_dtype=torch.float32), "dW   mm(G^T, x, out_dtype=fp32)")
except Exception as ex:
    print("mm out_dtype not available:", type(ex).__name__, str(ex)[:100])
x3 = torch.randn(n, 3000, device=dev, dtype=torch.bfloat16); W3 = torch.randn(2048, 3000, device=dev, dtype=torch.bfloat16)
t(lambda: torch.nn.functional.linear(x3, W3), "fwd1 x[10k,3000] @ W1^T")
t(lambda: G.t() @ x3, "dW1  G^T @ x[10k,3000]")
x3p = torch.randn(n, 3072, device=dev, dtype=torch.bfloat16); W3p = torch.randn(2048, 3072, device=dev, dtype=torch.bfloat16)
t(lambda: torch.nn.functional.linear(x3p, W3p), "fwd1 padded K=3072")
t(lambda: G.t() @ x3p, "dW1  padded 3072")
