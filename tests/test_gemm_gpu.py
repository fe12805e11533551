"""GPU: csrc/gemm_bf16.hip (the GAT layers' dense map, h = x W^T, on the matrix cores) against a torch fp32 reference of the
same op on the same bf16-rounded operands.  Tolerance: the kernel accumulates in fp32 and rounds the result to bf16 once, so
every element is within one bf16 ulp (2^-8 relative) of the fp32 reference plus the fp32 summation-order noise (K products:
<= 1e-5 * sum |a b|)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _call(A, B, M=None):
    from spadot_amd import _lib
    lib = _lib.model_lib()
    M = A.shape[0] if M is None else M
    N, K = B.shape
    C = torch.full((A.shape[0], N), float("nan"), device=DEV, dtype=torch.bfloat16)
    rc = lib.spadot_gemm_tn_bf16(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0), M, N, K,
                                 torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return rc, C


@pytest.mark.parametrize("M,N,K", [
    (1, 256, 64),            # one row, one K-step (no steady state in the pipeline)
    (319, 256, 128),         # one short tile, two K-steps
    (321, 512, 192),         # one full tile + one row; odd number of K-steps
    (2000, 2048, 2048),      # layer-2 shape, several row tiles
    (9980, 2048, 3072),      # the cfg3 layer-1 shape (3000 genes padded to 3072): 32 x 8 tiles, the last 60 rows high
])
def test_gemm_matches_fp32_reference(M, N, K):
    g = torch.Generator(device=DEV).manual_seed(M * 7 + K)
    A = (torch.randn((M, K), device=DEV, generator=g) * 0.7).bfloat16()
    B = (torch.randn((N, K), device=DEV, generator=g) * 0.1).bfloat16()
    rc, C = _call(A, B)
    assert rc == 0
    ref = A.float() @ B.float().t()
    mag = A.float().abs() @ B.float().abs().t()
    err = (C.float() - ref).abs()
    bound = 2.0 ** -8 * ref.abs() + 1e-5 * mag + 1e-30
    assert torch.isfinite(C.float()).all()
    assert bool((err <= bound).all()), float((err / bound).max())
    # asymmetric operands: a transposed or row-permuted result cannot pass; and every launch gives the same bits
    rc2, C2 = _call(A, B)
    assert rc2 == 0 and torch.equal(C, C2)


def test_gemm_row_strides_and_untouched_rows():
    """Operands as views of wider buffers (row strides > K, N), and M smaller than the buffers: rows >= M stay as they were."""
    g = torch.Generator(device=DEV).manual_seed(5)
    M, N, K = 700, 256, 128
    Abuf = (torch.randn((M + 50, K + 64), device=DEV, generator=g)).bfloat16()
    Bbuf = (torch.randn((N, K + 8), device=DEV, generator=g) * 0.1).bfloat16()
    A, B = Abuf[:, :K], Bbuf[:, :K]
    rc, C = _call(A, B, M=M)
    assert rc == 0
    ref = A[:M].float() @ B.float().t()
    np.testing.assert_allclose(C[:M].float().cpu().numpy(), ref.cpu().numpy(), rtol=2 ** -7, atol=1e-3)
    assert torch.isnan(C[M:].float()).all()


@pytest.mark.parametrize("M,N,K", [(64, 128, 64), (64, 256, 32), (64, 256, 96)])
def test_gemm_refuses_shapes_it_does_not_cover(M, N, K):
    A = torch.zeros((M, K), device=DEV, dtype=torch.bfloat16)
    B = torch.zeros((N, K), device=DEV, dtype=torch.bfloat16)
    rc, _ = _call(A, B)
    assert rc == -22


def test_dense_map_uses_own_gemm_and_matches_library():
    """ops.gemm_tn at a layer shape: the same result (to bf16 rounding) with the kernel and with the library."""
    from spadot_amd import ops
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn((3000, 2048), device=DEV, generator=g).bfloat16()
    w = (torch.randn((2048, 2048), device=DEV, generator=g) * 0.02).bfloat16()
    ops.GEMM_OWN[0] = True
    a = ops.gemm_tn(x, w)
    ops.GEMM_OWN[0] = False
    try:
        b = ops.gemm_tn(x, w)
    finally:
        ops.GEMM_OWN[0] = True
    np.testing.assert_allclose(a.float().cpu().numpy(), b.float().cpu().numpy(), rtol=2 ** -7, atol=2e-3)
