// ot_small.hip -- whole unbalanced-OT solves of SMALL problems (I, J <= 64), one wavefront per problem, for gfx950.
//
// What it replaces: the only OT problem the training loop solves is the 10 x 10 coupling of the K-means centres of
// consecutive time points (/root/reference/SpaDOT/utils/_train_utils.py:309-321 -> ot_solvers.py:95-121 -> 12 ctypes
// crossings into ot_func.cpp per solve).  The streaming solver of ot_sinkhorn.hip pays ~40 launches and ~25 host
// synchronisations per stage for such a problem (3.6 ms per solve; the reference's single CPU thread needs 0.07 ms).
// Here the WHOLE solve -- cost matrix from the latents, its median, all six epsilon stages with the reference's batch /
// tau-absorb / stopping rules, the plan and its row-normalised form -- is ONE launch with no host synchronisation, and
// the T - 1 pair problems of an epoch are the workgroups of that one launch.
//
// Layout (per workgroup = one 64-lane wave, everything in LDS, fp64 throughout = the reference's arithmetic):
//   C, K        I x ldk, ldk = J | 1 (odd row stride: a lane per row walks its row without bank conflicts,
//               a lane per column reads consecutive addresses)
//   lane t      owns row t (a_t, old_a_t, u_t, p_t) and column t (b_t, old_b_t, v_t, q_t) in registers;
//               u, v, a.dx and b.dy are mirrored in LDS for the sums and the rebuild of K
// One wave needs no s_barrier: DS operations of a wave execute in issue order; __syncthreads() below compiles to a
// wave barrier under __launch_bounds__(64).
//
// Arithmetic follows ot_func.cpp line by line (cited at each step); sums over a row / a column run in index order
// like gemv / gemtv (:131-208); the scaling update is exp(alpha log(p / s) - u / (lambda + eps)) as in ot_sinkhorn.hip.
// The loop is bounded: a stage that has not converged after HARD_CAP iterations ends the solve with status bit 2
// (the reference would spin; the caller then falls back to the streaming solver, which keeps that behaviour).
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdio>

#include "../../include/spadot_ot.h"
#include "per_device.h"

namespace {

constexpr int SMALL_MAX = 64;            // max I, J
constexpr int SMALL_MAX_D = 32;          // max latent dimension
constexpr int SMALL_BATCH = 24;          // problems per launch (descriptors travel as kernel arguments)
constexpr int HARD_CAP = 1 << 20;        // scaling iterations per stage before the kernel gives up

struct SmallBatch { spadot_ot_small_problem p[SMALL_BATCH]; };

__device__ __forceinline__ double wsum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;   // every lane
}

__device__ __forceinline__ double clamp_inf_s(double x) {     // ot_func.cpp:29-40
    if (isinf(x)) return x < 0 ? -(double)FLT_MAX : (double)FLT_MAX;
    return x;
}

__device__ __forceinline__ double scale_update_s(double num, double sum, double alpha, double shift) {
    return exp(alpha * log(num / sum) - shift);               // (num / sum)^alpha * exp(-shift), ot_func.cpp:633-636
}

// K = exp((u_i + v_j - C_ij) / eps), ot_func.cpp:563-567 / :802-806
__device__ __forceinline__ void rebuild_K(double *K, const double *C, const double *uL, const double *vL, double eps,
                                          int I, int J, int ldk, int lane) {
    for (int t = lane; t < I * J; t += 64) {
        const int i = t / J, j = t - i * J;
        K[i * ldk + j] = exp((uL[i] + vL[j] - C[i * ldk + j]) / eps);
    }
}

__global__ __launch_bounds__(64) void k_ot_small(SmallBatch B, int d, int divide_by_median, spadot_ot_config cfg,
                                                 spadot_ot_small_info *__restrict__ info_dev, int first) {
    extern __shared__ double sm[];
    const spadot_ot_small_problem P = B.p[blockIdx.x];
    const int I = P.I, J = P.J, lane = threadIdx.x;
    const int ldk = J | 1;
    const int n = I * J;
    double *C = sm;                              // I x ldk
    double *K = C + SMALL_MAX * (SMALL_MAX + 1); // I x ldk (before the solve: sort space for the median, n <= 4096)
    double *uL = K + SMALL_MAX * (SMALL_MAX + 1);
    double *vL = uL + SMALL_MAX;
    double *wa = vL + SMALL_MAX;                 // a.dx (length I)
    double *wb = wa + SMALL_MAX;                 // b.dy (length J)
    double *xy = wb + SMALL_MAX;                 // latents: x (I x d) then y (J x d)

    // ------------------------------------------------------------------ cost (ot_solvers.py:101-103)
    if (P.x_dev != nullptr) {
        double *xs = xy, *ys = xy + SMALL_MAX * SMALL_MAX_D;
        for (int t = lane; t < I * d; t += 64) xs[t] = P.x_dev[t];
        for (int t = lane; t < J * d; t += 64) ys[t] = P.y_dev[t];
        __syncthreads();
        // sklearn euclidean_distances(squared=True): -2 x.y + |x|^2 + |y|^2, clipped at 0 (same chain as ot_cost.hip)
        double xx = 0.0, yy = 0.0;
        if (lane < I) for (int k = 0; k < d; k++) xx += xs[lane * d + k] * xs[lane * d + k];
        if (lane < J) for (int k = 0; k < d; k++) yy += ys[lane * d + k] * ys[lane * d + k];
        wa[lane] = xx; wb[lane] = yy;
        __syncthreads();
        for (int t = lane; t < n; t += 64) {
            const int i = t / J, j = t - i * J;
            double dot = 0.0;
            for (int k = 0; k < d; k++) dot += xs[i * d + k] * ys[j * d + k];
            double v = -2.0 * dot;
            v += wa[i];
            v += wb[j];
            C[i * ldk + j] = v > 0.0 ? v : 0.0;
        }
    } else {
        for (int t = lane; t < n; t += 64) {
            const int i = t / J, j = t - i * J;
            C[i * ldk + j] = P.C_dev[t];
        }
    }
    __syncthreads();
    if (divide_by_median) {
        // np.median: bitonic sort of the n entries (padded with +inf to a power of two) in the K region
        int m = 64;
        while (m < n) m <<= 1;
        for (int t = lane; t < m; t += 64) {
            const int i = t / J, j = t - i * J;
            K[t] = t < n ? C[i * ldk + j] : INFINITY;
        }
        __syncthreads();
        for (int k = 2; k <= m; k <<= 1) {
            for (int s = k >> 1; s > 0; s >>= 1) {
                for (int t = lane; t < m; t += 64) {
                    const int o = t ^ s;
                    if (o > t) {
                        const double lo = K[t], hi = K[o];
                        const bool up = (t & k) == 0;
                        if ((lo > hi) == up) { K[t] = hi; K[o] = lo; }
                    }
                }
                __syncthreads();
            }
        }
        const double med = (n & 1) ? K[n / 2] : (K[n / 2 - 1] + K[n / 2]) / 2.0;
        __syncthreads();
        for (int t = lane; t < n; t += 64) {
            const int i = t / J, j = t - i * J;
            C[i * ldk + j] = C[i * ldk + j] / med;
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ p = G, q = mean(G), dx = 1/I, dy = 1/J
    const double dx = 1.0 / I, dy = 1.0 / J;                      // ot_solvers.py:220-227
    double p = 1.0;
    if (lane < I && P.G_dev != nullptr) p = P.G_dev[lane];
    wa[lane] = lane < I ? p : 0.0;
    __syncthreads();
    double gs = 0.0;
    for (int i = 0; i < I; i++) gs += wa[i];                      // sequential, like the host sum
    const double q = gs / I;
    __syncthreads();

    double a = 1.0, b = 1.0, old_a = 1.0, old_b = 1.0, u = 0.0, v = 0.0;
    const int S = 5;
    const double f = exp(-log(cfg.epsilon) / S);                  // ot_solvers.py:217-218
    double eps_i = cfg.epsilon0 * f;
    double gap = INFINITY;
    int absorbs = 0, checks = 0, status = 0;
    int stage_iters[6] = {0, 0, 0, 0, 0, 0};
    const double IJ = (double)(I * J);

    for (int e = 0; e <= S; e++) {
        // ot_solvers.py:249-254: absorb the scalings, a = b = 1
        if (lane < I) u = u + eps_i * log(a);
        if (lane < J) v = v + eps_i * log(b);
        a = b = old_a = old_b = 1.0;
        eps_i = eps_i / f;
        const double eps = eps_i;
        const double al1 = cfg.lambda1 / (cfg.lambda1 + eps), al2 = cfg.lambda2 / (cfg.lambda2 + eps);
        const double inv_l1e = 1.0 / (cfg.lambda1 + eps), inv_l2e = 1.0 / (cfg.lambda2 + eps);
        const bool last = e == S;
        const double thr = last ? cfg.tolerance : 1e-6;           // ot_solvers.py:262
        uL[lane] = u; vL[lane] = v;
        __syncthreads();
        rebuild_K(K, C, uL, vL, eps, I, J, ldk, lane);            // update_k, ot_func.cpp:546-568
        double sum_kbar = 0.0;
        if (last) {
            double acc = 0.0;
            for (int t = lane; t < n; t += 64) {
                const int i = t / J, j = t - i * J;
                acc += exp(-C[i * ldk + j] / eps);                // Kbar, :558-560 (only its sum enters the gap)
            }
            sum_kbar = wsum(acc);
        }
        __syncthreads();

        gap = 1e100;                                              // update_process, ot_func.cpp:830-930
        int cur_iter = 0, done = 0;                               // cur_iter restarts per stage (ot_solvers.py:282-289)
        const int iters = last ? cfg.batch_size : 5;
        while (gap > thr) {
            for (int it = 0; it < iters; it++) {                  // step1_process, :689-828
                cur_iter += 1;
                old_a = a; old_b = b;
                wb[lane] = lane < J ? b * dy : 0.0;
                __syncthreads();
                if (lane < I) {
                    double s = 0.0;
                    const double *kr = K + lane * ldk;
                    for (int j = 0; j < J; j++) s += kr[j] * wb[j];                  // gemv, :131-159
                    a = scale_update_s(p, s, al1, u * inv_l1e);                      // :633-636
                }
                wa[lane] = lane < I ? a * dx : 0.0;
                __syncthreads();
                if (lane < J) {
                    double s = 0.0;
                    for (int i = 0; i < I; i++) s += K[i * ldk + lane] * wa[i];      // gemtv, :177-208
                    b = scale_update_s(q, s, al2, v * inv_l2e);                      // :665-668
                }
                done++;
                const bool over = (lane < I && a > cfg.tau) || (lane < J && b > cfg.tau);   // :778-790 (no abs)
                if (__any(over)) {                                                   // :792-814
                    if (lane < I) u = u + eps * log(a);
                    if (lane < J) v = v + eps * log(b);
                    a = b = 1.0;
                    __syncthreads();
                    uL[lane] = u; vL[lane] = v;
                    __syncthreads();
                    rebuild_K(K, C, uL, vL, eps, I, J, ldk, lane);
                    __syncthreads();
                    absorbs++;
                }
                if (cur_iter >= cfg.max_iter) { cur_iter = -1; status |= 1; break; } // :821-824 (+ :869: the loop goes on)
            }
            // convergence measure with the un-stabilised scalings, :880-923
            const double eu = exp(u / eps), ev = exp(v / eps);
            const double ta = a * eu, tb = b * ev;
            if (!last) {
                const double t1 = ta - old_a * eu, t2 = tb - old_b * ev;
                const double d1 = wsum(lane < I ? t1 * t1 : 0.0), n1 = wsum(lane < I ? ta * ta : 0.0);
                const double d2 = wsum(lane < J ? t2 * t2 : 0.0), n2 = wsum(lane < J ? tb * tb : 0.0);
                const double g1 = sqrt(d1) / (1.0 + sqrt(n1)), g2 = sqrt(d2) / (1.0 + sqrt(n2));
                gap = (g1 < g2) ? g2 : g1;                                           // std::max: a NaN g1 wins
            } else {
                // R = a K b (:570-584); primal (:357-462), dual (:464-490), gap (:492-544)
                wa[lane] = lane < I ? a : 0.0;
                wb[lane] = lane < J ? b : 0.0;
                __syncthreads();
                double ent = 0.0, cost = 0.0, sumR = 0.0;
                for (int t = lane; t < n; t += 64) {
                    const int i = t / J, j = t - i * J;
                    const double r = K[i * ldk + j] * wa[i] * wb[j];
                    ent += r * clamp_inf_s(log(r)) - r;
                    cost += r * C[i * ldk + j];
                    sumR += r;
                }
                ent = wsum(ent) + sum_kbar; cost = wsum(cost); sumR = wsum(sumR);
                double rs = 0.0, cs = 0.0;
                if (lane < I) {
                    const double *kr = K + lane * ldk;
                    for (int j = 0; j < J; j++) rs += (kr[j] * a * wb[j]) * dy;
                }
                if (lane < J)
                    for (int i = 0; i < I; i++) cs += (K[i * ldk + lane] * wa[i] * b) * dx;
                // fdiv (:308-322): lambda sum dx (x log(x / p) - x + p);  fdivstarexp (:340-355)
                const double k1 = wsum(lane < I ? dx * (rs * log(rs / p) - rs + p) : 0.0);
                const double k2 = wsum(lane < J ? dy * (cs * log(cs / q) - cs + q) : 0.0);
                const double pri = cfg.lambda1 * k1 + cfg.lambda2 * k2 + (eps * ent + cost) / IJ;
                const double c1 = wsum(lane < I ? (p * dx) * (exp((-eps * log(ta)) / cfg.lambda1) - 1.0) : 0.0);
                const double c2 = wsum(lane < J ? (q * dy) * (exp((-eps * log(tb)) / cfg.lambda2) - 1.0) : 0.0);
                const double dua = -cfg.lambda1 * c1 - cfg.lambda2 * c2 - eps * (sumR - sum_kbar) / IJ;
                gap = (pri - dua) / fabs(pri);
                __syncthreads();
            }
            gap = __shfl(gap, 0, 64);            // (already identical on every lane: butterfly sums)
            checks++;
            if (done > HARD_CAP) { status |= 2; break; }
        }
        stage_iters[e] = done;
        if (status & 2) break;
    }

    // ------------------------------------------------------------------ plan R / J (ot_solvers.py:449) and its row-normalised form
    wa[lane] = lane < I ? a : 0.0;
    wb[lane] = lane < J ? b : 0.0;
    __syncthreads();
    if (P.plan_dev != nullptr)
        for (int t = lane; t < n; t += 64) {
            const int i = t / J, j = t - i * J;
            P.plan_dev[t] = (K[i * ldk + j] * wa[i] * wb[j]) / (double)J;
        }
    if (P.gamma_rownorm_dev != nullptr && !(status & 2)) {
        // _train_utils.py:299-300: gamma / gamma.sum(axis=1), NaN / inf -> 0.  (A solve that ended on the iteration cap did
        // not converge: the tensor the OT-loss kernel reads keeps the previous plan until the caller's fallback rewrites it.)
        double rsum = 0.0;
        if (lane < I) {
            const double *kr = K + lane * ldk;
            for (int j = 0; j < J; j++) rsum += (kr[j] * a * wb[j]) / (double)J;
        }
        uL[lane] = rsum;
        __syncthreads();
        for (int t = lane; t < n; t += 64) {
            const int i = t / J, j = t - i * J;
            const double g = ((K[i * ldk + j] * wa[i] * wb[j]) / (double)J) / uL[i];
            P.gamma_rownorm_dev[t] = (isnan(g) || isinf(g)) ? 0.f : (float)g;
        }
    }
    if (info_dev != nullptr && lane == 0) {
        spadot_ot_small_info r;
        r.gap = gap;
        for (int e = 0; e < 6; e++) r.stage_iters[e] = stage_iters[e];
        r.absorbs = absorbs; r.gap_checks = checks; r.status = status; r.reserved = 0;
        info_dev[first + blockIdx.x] = r;
    }
}

}  // namespace

extern "C" {

int spadot_ot_small_max(void) { return SMALL_MAX; }

// include/spadot_ot.h.  Asynchronous on `stream`: no allocation, no synchronisation.
int spadot_ot_small_solve(int nprob, const spadot_ot_small_problem *probs, int d, int divide_by_median,
                          const spadot_ot_config *cfg, spadot_ot_small_info *info_dev, void *stream) {
    if (nprob < 0 || (nprob > 0 && !probs) || !cfg) return -22;
    if (cfg->batch_size < 1) return -22;
    for (int k = 0; k < nprob; k++) {
        const spadot_ot_small_problem &p = probs[k];
        if (p.I < 1 || p.J < 1 || p.I > SMALL_MAX || p.J > SMALL_MAX) return -22;
        if (p.x_dev != nullptr) {
            if (p.y_dev == nullptr || d < 1 || d > SMALL_MAX_D) return -22;
        } else if (p.C_dev == nullptr) {
            return -22;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "libspadot_ot: no HIP device -- there is no CPU path\n");
        (void)hipGetLastError();
        return -5;
    }
    const size_t lds = sizeof(double) * (2 * (size_t)SMALL_MAX * (SMALL_MAX + 1) + 4 * SMALL_MAX + 2 * (size_t)SMALL_MAX * SMALL_MAX_D);
    static PerDeviceFlag attr_set;     
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_ot_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return -5;
        }
        attr_set = true;
    }
    for (int first = 0; first < nprob; first += SMALL_BATCH) {
        const int nb = nprob - first < SMALL_BATCH ? nprob - first : SMALL_BATCH;
        SmallBatch B;
        for (int k = 0; k < nb; k++) B.p[k] = probs[first + k];
        for (int k = nb; k < SMALL_BATCH; k++) B.p[k] = probs[first];
        hipLaunchKernelGGL(k_ot_small, dim3(nb), dim3(64), lds, (hipStream_t)stream, B, d, divide_by_median, *cfg, info_dev, first);
        if (hipGetLastError() != hipSuccess) return -5;
    }
    return 0;
}

}  // extern "C"
