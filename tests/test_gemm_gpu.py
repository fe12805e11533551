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


@pytest.mark.parametrize("M,N,K,tail,slices", [
    (640, 256, 256, 1, 2),             # two row panels, the second in two slices
    (1000, 512, 192, 2, 3),            # rows past M in the sliced panels; 3 K-steps in 3 slices
    (10112, 2048, 2048, 5, 4),         # the second GAT layer at the bench shape: 216 whole tiles + 40 x 4 slices
    (9980, 2048, 2048, 5, 4),
    (700, 256, 512, 3, 8),             # every panel sliced
])
def test_gemm_tn_with_sliced_tail_panels_matches_the_whole_tile_form(M, N, K, tail, slices):
    """spadot_gemm_tn_bf16_split: the last row panels as slices of the contraction + one fp32 sum per element -- every row of
    the whole panels is bit-identical to spadot_gemm_tn_bf16, the sliced rows agree with an fp64 product like the whole-tile
    form does (one rounding to bf16 of an fp32 sum), repeats are bit-identical, bad arguments are refused."""
    from spadot_amd import _lib
    lib = _lib.model_lib()
    g = torch.Generator(device=DEV).manual_seed(M + K)
    A = (torch.randn((M, K), device=DEV, generator=g) * 0.5).bfloat16()
    B = (torch.randn((N, K), device=DEV, generator=g) * 0.3).bfloat16()
    st = torch.cuda.current_stream().cuda_stream
    ref_whole = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    assert lib.spadot_gemm_tn_bf16(A.data_ptr(), K, B.data_ptr(), K, ref_whole.data_ptr(), N, M, N, K, st) == 0
    need = int(lib.spadot_gemm_bf16_split_workspace(M, N, tail, slices))
    mt = (M + 319) // 320
    assert need == tail * (N // 256) * slices * 320 * 256
    outs = []
    for _ in range(2):
        C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        ws = torch.full((need,), float("nan"), device=DEV)
        assert lib.spadot_gemm_tn_bf16_split(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, tail, slices, ws.data_ptr(), st) == 0
        torch.cuda.synchronize()
        outs.append(C)
    C = outs[0]
    assert torch.equal(C, outs[1]) and torch.isfinite(C.float()).all()
    whole_rows = (mt - tail) * 320
    assert torch.equal(C[:whole_rows], ref_whole[:whole_rows])
    ref = (A.double() @ B.double().t())
    scale = float(ref.abs().max())
    assert float((C.double() - ref).abs().max()) <= 2 ** -7 * scale + 1e-3
    assert float((C[whole_rows:].double() - ref_whole[whole_rows:].double()).abs().max()) <= 2 ** -7 * scale + 1e-3
    assert lib.spadot_gemm_tn_bf16_split(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, mt + 1, slices, ws.data_ptr(), st) == -22
    assert lib.spadot_gemm_tn_bf16_split(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, tail, 9, ws.data_ptr(), st) == -22
    assert lib.spadot_gemm_tn_bf16_split(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, tail, slices, None, st) == -22


@pytest.mark.parametrize("M,N,K,Kp,slices,tile_k", [
    (64, 256, 256, 256, 1, 256),            # one chunk, one tile, no slicing
    (100, 256, 256, 256, 2, 256),           # rows past M in the second chunk come from the zero row
    (1000, 512, 300, 512, 3, 256),          # partial last column tile, three slices (one chunk short of even)
    (9980, 2048, 2048, 2048, 4, 256),       # layer-2 shape: 64 tiles x 4 slices
    (9980, 2048, 3000, 3072, 2, 256),       # layer-1 shape: 96 tiles x 2 slices, output row stride 3000
    (64, 256, 192, 192, 1, 192),            # 256 x 192 tiles: one tile whose X rows END at the tile (columns behind: zero row)
    (100, 256, 400, 576, 2, 192),           # partial third column tile, rows past M
    (1000, 512, 300, 384, 3, 192),
    (9980, 2048, 3000, 3072, 2, 192),       # layer-1 shape: 128 tiles x 2 slices = one round of the chip
    (10112, 2048, 3000, 3072, 2, 192),      # the same with rows padded to a multiple of 128
])
def test_wgrad_matches_fp32_reference(M, N, K, Kp, slices, tile_k):
    """csrc/gemm_wgrad_bf16.hip: dW = G^T X in fp32 from bf16 operands.  fp32 accumulation over M products in the kernel's own
    order: within 2e-6 * sum |g x| (+ 1e-6 relative) of the fp64 value; the same bits on every launch (slice order is fixed)."""
    from spadot_amd import _lib
    lib = _lib.model_lib()
    g = torch.Generator(device=DEV).manual_seed(M + K)
    G = (torch.randn((M, N), device=DEV, generator=g) * 0.3).bfloat16()
    X = torch.zeros((M, Kp), device=DEV, dtype=torch.bfloat16)
    X[:, :K] = (torch.randn((M, K), device=DEV, generator=g) * 0.5).bfloat16()
    outs = []
    need = int(lib.spadot_gemm_wgrad_bf16_workspace_tiled(M, N, K, slices, tile_k))
    assert need >= 0 and (need == 0) == (slices <= 1 or M <= 64)
    if tile_k == 256:
        assert need == int(lib.spadot_gemm_wgrad_bf16_workspace(M, N, K, slices))
    zrow = torch.zeros(256, device=DEV, dtype=torch.bfloat16)
    for _ in range(2):
        dW = torch.full((N, K), float("nan"), device=DEV)
        ws = torch.full((max(need, 4),), float("nan"), device=DEV)          # caller-owned partials: the library keeps no state
        rc = lib.spadot_gemm_wgrad_bf16_tiled(G.data_ptr(), N, X.data_ptr(), Kp, dW.data_ptr(), K, M, N, K, slices, tile_k,
                                              ws.data_ptr(), zrow.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0
        outs.append(dW)
    ref = G.double().t() @ X[:, :K].double()
    mag = G.double().abs().t() @ X[:, :K].double().abs()
    err = (outs[0].double() - ref).abs()
    assert torch.isfinite(outs[0]).all()
    assert bool((err <= 2e-6 * mag + 1e-6 * ref.abs() + 1e-12).all()), float((err / (2e-6 * mag + 1e-12)).max())
    assert torch.equal(outs[0], outs[1])


def test_wgrad_refuses_what_it_does_not_cover():
    from spadot_amd import _lib
    lib = _lib.model_lib()
    G = torch.zeros((64, 128), device=DEV, dtype=torch.bfloat16)
    X = torch.zeros((64, 256), device=DEV, dtype=torch.bfloat16)
    dW = torch.zeros((128, 256), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    z = torch.zeros(256, device=DEV, dtype=torch.bfloat16)
    assert lib.spadot_gemm_wgrad_bf16(G.data_ptr(), 128, X.data_ptr(), 256, dW.data_ptr(), 256, 64, 128, 256, 1, None, z.data_ptr(), st) == -22   # N % 256
    G = torch.zeros((64, 256), device=DEV, dtype=torch.bfloat16)
    X = torch.zeros((64, 200), device=DEV, dtype=torch.bfloat16)
    dW = torch.zeros((256, 200), device=DEV)
    assert lib.spadot_gemm_wgrad_bf16(G.data_ptr(), 256, X.data_ptr(), 200, dW.data_ptr(), 200, 64, 256, 200, 1, None, z.data_ptr(), st) == -22   # X rows too short
    # several slices without a workspace, or without the zero row: refused before anything is launched
    G = torch.zeros((512, 256), device=DEV, dtype=torch.bfloat16)
    X = torch.zeros((512, 256), device=DEV, dtype=torch.bfloat16)
    dW = torch.zeros((256, 256), device=DEV)
    assert lib.spadot_gemm_wgrad_bf16_workspace(512, 256, 256, 4) == 4 * 65536
    assert lib.spadot_gemm_wgrad_bf16(G.data_ptr(), 256, X.data_ptr(), 256, dW.data_ptr(), 256, 512, 256, 256, 4, None, z.data_ptr(), st) == -22
    assert lib.spadot_gemm_wgrad_bf16(G.data_ptr(), 256, X.data_ptr(), 256, dW.data_ptr(), 256, 512, 256, 256, 1, None, None, st) == -22
    # tile widths other than 256 / 192; X rows that end inside the last 192-wide tile
    assert lib.spadot_gemm_wgrad_bf16_tiled(G.data_ptr(), 256, X.data_ptr(), 256, dW.data_ptr(), 256, 512, 256, 256, 1, 128, None, z.data_ptr(), st) == -22
    assert lib.spadot_gemm_wgrad_bf16_tiled(G.data_ptr(), 256, X.data_ptr(), 256, dW.data_ptr(), 256, 512, 256, 256, 1, 192, None, z.data_ptr(), st) == -22
    assert lib.spadot_gemm_wgrad_bf16_workspace_tiled(512, 256, 256, 4, 192) == 4 * 2 * 256 * 192


@pytest.mark.parametrize("M,N,K", [(1, 256, 64), (321, 512, 192), (2000, 2048, 2048), (9980, 2048, 2048)])
def test_gemm_nn_matches_fp32_reference(M, N, K):
    """The B-stored-[K x N] form of csrc/gemm_bf16.hip (input gradient of a dense map): transposed LDS reads for B."""
    from spadot_amd import _lib
    lib = _lib.model_lib()
    g = torch.Generator(device=DEV).manual_seed(M * 3 + K)
    A = (torch.randn((M, K), device=DEV, generator=g) * 0.7).bfloat16()
    B = (torch.randn((K, N), device=DEV, generator=g) * 0.1).bfloat16()
    outs = []
    for _ in range(2):
        C = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
        rc = lib.spadot_gemm_nn_bf16(A.data_ptr(), K, B.data_ptr(), N, C.data_ptr(), N, M, N, K, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0
        outs.append(C)
    ref = A.float() @ B.float()
    mag = A.float().abs() @ B.float().abs()
    err = (outs[0].float() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-5 * mag + 1e-30).all()), float(err.max())
    assert torch.equal(outs[0], outs[1])
