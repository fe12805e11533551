"""Why the SPD inverse of the SVGP branch stays a SCALAR sweep (round 3).  A blocked sweep -- 16 x 16 pivot blocks P, panel
C P^-1 and the rank-16 update M[i][j] -= M[i][k] P^-1 M[k][j] on v_mfma_f64_16x16x4 -- was written for gfx950 and run on the
MI355X (one workgroup per matrix, tiles in accumulator layout, every wave inverting P itself): it reproduced this file's numbers
bit for bit in spirit and was (a) no faster than k_spd_sweep (212 vs 202 us for 40 matrices of m = 236: the fp64 MFMA rate of
the chip equals its fp64 vector rate, 32 FLOP/clk/SIMD, and one compute unit per matrix bounds both forms), and (b) UNSTABLE on
matrices conditioned like Sigma_l: with an explicit inverse of the pivot block, the error of P^-1 (eps cond(P)) is multiplied
by |C|^2 in the Schur update and the jitter-sized complements that follow are then wrong in the fourth digit.  The scalar
sweep divides by one pivot at a time and has no such term.  This script shows (b) in numpy; run it anywhere."""
import numpy as np


def sweep(A, B):
    """Symmetric sweep with B x B pivot blocks (B = 1: the scalar sweep of k_spd_sweep); returns A^-1."""
    m = A.shape[0]
    NB = (m + B - 1) // B
    n = NB * B
    M = np.eye(n)
    M[:m, :m] = A
    for k in range(NB):
        ks = slice(k * B, (k + 1) * B)
        Pinv = np.linalg.inv(M[ks, ks])
        C = M[:, ks].copy()
        CP = C @ Pinv
        M -= CP @ C.T
        M[:, ks] = CP
        M[ks, :] = CP.T
        M[ks, ks] = -Pinv
    return -M[:m, :m]


if __name__ == "__main__":
    for m in (32, 95, 128, 236):
        rng = np.random.default_rng(m)
        Bm = rng.normal(size=(m, max(2, m // 3)))                      # tests/test_model_gpu.py: rank m / 3 + jitter 1e-2
        A = Bm @ Bm.T * 50.0 + 1e-2 * np.eye(m)
        eye = np.eye(m)
        print(f"m = {m:3d}  cond {np.linalg.cond(A):.1e}   |A X - I|:  16 x 16 pivot blocks {np.abs(A @ sweep(A, 16) - eye).max():.1e}   "
              f"scalar sweep {np.abs(A @ sweep(A, 1) - eye).max():.1e}   numpy.linalg.inv {np.abs(A @ np.linalg.inv(A) - eye).max():.1e}")
