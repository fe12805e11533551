"""GPU: csrc/gemm_bf16.hip against the library GEMM (torch.nn.functional.linear, bf16) at the training shapes.
Checks the result (fp32-accumulated bf16 products: equal up to bf16 output rounding and summation order) and times both."""
import json
import sys

import torch

from spadot_amd import _lib


def _load():
    """The product library, or (GEMM_LIB=path) a stand-alone build of csrc/gemm_bf16.hip (ablation variants)."""
    import ctypes
    import os
    path = os.environ.get("GEMM_LIB")
    if not path:
        return _lib.model_lib()
    lib = ctypes.CDLL(path)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.spadot_gemm_tn_bf16.argtypes = [vp, ci, vp, ci, vp, ci, ci, ci, ci, vp]
    lib.spadot_gemm_tn_bf16.restype = ci
    return lib


def run(M, N, K, reps=30):
    lib = _load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.randn((M, K), device="cuda", generator=g) * 0.5).bfloat16()
    B = (torch.randn((N, K), device="cuda", generator=g) * 0.05).bfloat16()
    C = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.spadot_gemm_tn_bf16(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    ref = torch.nn.functional.linear(A, B)
    ref32 = A.float() @ B.float().t()
    err = (C.float() - ref32).abs().max().item()
    err_lib = (ref.float() - ref32).abs().max().item()
    scale = ref32.abs().max().item()

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    mine = lambda: lib.spadot_gemm_tn_bf16(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, st)
    libf = lambda: torch.nn.functional.linear(A, B)
    tm, tl = [], []
    timed(mine, 20), timed(libf, 20)
    for _ in range(7):                      # interleaved rounds in one process: median of each
        tm.append(timed(mine, reps))
        tl.append(timed(libf, reps))
    t_mine, t_lib = sorted(tm)[3], sorted(tl)[3]
    fl = 2.0 * M * N * K
    return dict(M=M, N=N, K=K, us=round(t_mine, 1), us_library=round(t_lib, 1), tflops=round(fl / t_mine / 1e6, 1),
                tflops_library=round(fl / t_lib / 1e6, 1), max_err=err, max_err_library=err_lib, scale=scale)


if __name__ == "__main__" and "wgrad" not in sys.argv:
    import os
    shapes = [(10240, 2048, 3072), (9980, 2048, 3072), (9980, 2048, 2048), (8500, 2048, 2048), (321, 256, 128), (9980, 2048, 3072)]
    for s in shapes:
        r = run(*s)
        print(json.dumps(r), flush=True)
        if not os.environ.get("GEMM_LIB"):
            assert r["max_err"] <= 2 ** -7 * r["scale"] + 1e-3, r


def run_wgrad(M, N, K, Kp, slices, tile_k=256, reps=30):
    """csrc/gemm_wgrad_bf16.hip against the library's g^T x (fp32 out), interleaved rounds."""
    lib = _lib.model_lib()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    G = (torch.randn((M, N), device="cuda", generator=g) * 0.3).bfloat16()
    X = torch.zeros((M, Kp), device="cuda", dtype=torch.bfloat16)
    X[:, :K] = (torch.randn((M, K), device="cuda", generator=g) * 0.5).bfloat16()
    dW = torch.empty((N, K), device="cuda")
    ref = torch.empty((N, K), device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    ws = torch.empty(max(4, int(lib.spadot_gemm_wgrad_bf16_workspace_tiled(M, N, K, slices, tile_k))), device="cuda:0")
    zrow = torch.zeros(256, device="cuda:0", dtype=torch.bfloat16)
    mine = lambda: lib.spadot_gemm_wgrad_bf16_tiled(G.data_ptr(), N, X.data_ptr(), Kp, dW.data_ptr(), K, M, N, K, slices, tile_k,
                                                    ws.data_ptr(), zrow.data_ptr(), st)
    libf = lambda: torch.mm(G.t(), X[:, :K], out_dtype=torch.float32, out=ref)
    assert mine() == 0
    timed(mine, 10), timed(libf, 10)
    tm, tl = [], []
    for _ in range(7):
        tm.append(timed(mine, reps))
        tl.append(timed(libf, reps))
    t_mine, t_lib = sorted(tm)[3], sorted(tl)[3]
    fl = 2.0 * M * N * K
    err = (dW - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
    return dict(kind="wgrad", M=M, N=N, K=K, slices=slices, tile_k=tile_k, us=round(t_mine, 1), us_library=round(t_lib, 1),
                tflops=round(fl / t_mine / 1e6, 1), tflops_library=round(fl / t_lib / 1e6, 1), max_rel_diff_vs_library=err)


if __name__ == "__main__" and "wgrad" in sys.argv:
    for shp in [(9980, 2048, 2048, 2048, 4), (8031, 2048, 2048, 2048, 4), (9980, 2048, 3000, 3072, 2), (9980, 2048, 3000, 3072, 2, 192),
                (10112, 2048, 3000, 3072, 2, 192), (10112, 2048, 3000, 3072, 2), (512, 256, 3000, 3072, 8), (512, 256, 3000, 3072, 8, 192)]:
        print(json.dumps(run_wgrad(*shp)), flush=True)
