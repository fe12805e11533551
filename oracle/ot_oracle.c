/*
 * oracle/ot_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (fp64, single thread) restatement of the unbalanced entropic-OT
 * scaling iterations of the reference's native library
 * (/root/reference/SpaDOT/utils/OT_loss/ot_func.cpp).  It exists so that the
 * HIP path can be checked on the GPU box, where the reference cannot travel.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's library; the product (spadot_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_ot.py checks every function here
 * against golden vectors produced by the reference itself
 * (oracle/gen_golden_ot.py -> tests/golden/ot_*.npz) and, in the build
 * container, against oracle/_ref/libot_ref.so compiled from the reference's
 * own ot_func.cpp.
 *
 * Each function cites the reference lines it follows.  Summation orders are
 * kept sequential (as in the reference) so the oracle agrees with it to a few
 * ulp; nothing here is vectorised or threaded on purpose.
 */
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* ot_func.cpp:29-40 -- +-inf clamps to +-FLT_MAX (the float constant, also for doubles). */
static double clamp_inf(double x)
{
    if (isinf(x)) return signbit(x) ? -(double)FLT_MAX : (double)FLT_MAX;
    return x;
}

/* out[i] = sum_j M[i,j]*x[j], j ascending.  ot_func.cpp:131-159 (its 4-row unroll keeps
 * each row's own sum sequential, so a plain loop is the same arithmetic). */
static void mat_vec(double *out, const double *M, const double *x, int rows, int cols)
{
    for (int i = 0; i < rows; i++) {
        const double *r = M + (size_t)i * cols;
        double s = 0.0;
        for (int j = 0; j < cols; j++) s += r[j] * x[j];
        out[i] = s;
    }
}

/* out[j] = sum_i M[i,j]*x[i], i ascending.  ot_func.cpp:177-208: the reference walks rows
 * in groups of four and adds the four products to out[j] one after the other, which is
 * the same order as adding row by row. */
static void matT_vec(double *out, const double *M, const double *x, int rows, int cols)
{
    for (int j = 0; j < cols; j++) out[j] = 0.0;
    for (int i = 0; i < rows; i++) {
        const double *r = M + (size_t)i * cols;
        const double xi = x[i];
        for (int j = 0; j < cols; j++) out[j] += r[j] * xi;
    }
}

/* ot_func.cpp:546-568.  Kbar = exp(-C/eps);  K = exp((u_i + v_j - C_ij)/eps). */
void orc_update_k(double *K, double *Kbar, const double *C, const double *u, const double *v,
                  double eps, int m, int n)
{
    size_t tot = (size_t)m * n;
    for (size_t t = 0; t < tot; t++) Kbar[t] = exp(-C[t] / eps);
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            size_t t = (size_t)i * n + j;
            K[t] = exp((u[i] + v[j] - C[t]) / eps);
        }
}

/* ot_func.cpp:570-584.  R = diag(a) K diag(b); the product is (K*a)*b in that order. */
void orc_update_R(double *R, const double *K, const double *a, const double *b, int m, int n)
{
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            size_t t = (size_t)i * n + j;
            R[t] = K[t] * a[i] * b[j];
        }
}

/* ot_func.cpp:586-687.  One scaling iteration:
 *   a = (p / (K (b.dy)))^alpha1 * exp(-u/(lambda1+eps))
 *   b = (q / (K^T (a.dx)))^alpha2 * exp(-v/(lambda2+eps))      (uses the NEW a) */
void orc_update_a_b(double *a, double *b, const double *K, const double *dx, const double *dy,
                    const double *p, const double *q, const double *u, const double *v,
                    double lambda1, double lambda2, double alpha1, double alpha2, double eps,
                    int m, int n)
{
    int l = m > n ? m : n;
    double *w = (double *)malloc(sizeof(double) * (size_t)l);
    double *s = (double *)malloc(sizeof(double) * (size_t)l);
    for (int j = 0; j < n; j++) w[j] = b[j] * dy[j];
    mat_vec(s, K, w, m, n);
    for (int i = 0; i < m; i++)
        a[i] = pow(p[i] / s[i], alpha1) * exp(-u[i] / (lambda1 + eps));
    for (int i = 0; i < m; i++) w[i] = a[i] * dx[i];
    matT_vec(s, K, w, m, n);
    for (int j = 0; j < n; j++)
        b[j] = pow(q[j] / s[j], alpha2) * exp(-v[j] / (lambda2 + eps));
    free(w);
    free(s);
}

/* ot_func.cpp:689-828.  `iters` iterations; after each one, if any a_i > tau or b_j > tau
 * (no abs), the scalings are absorbed into u, v, K is rebuilt from C and a = b = 1.
 * old_a/old_b hold the values from before the LAST iteration (pre-absorb).  Returns the
 * iteration counter, or -1 once it reaches max_iter. */
int orc_step1(double *a, double *b, double *old_a, double *old_b, double *K, const double *C,
              const double *dx, const double *dy, const double *p, const double *q,
              double *u, double *v, int cur_iter, int max_iter, int iters, double tau,
              double lambda1, double lambda2, double alpha1, double alpha2, double eps,
              int m, int n)
{
    for (int it = 0; it < iters; it++) {
        cur_iter += 1;
        memcpy(old_a, a, sizeof(double) * (size_t)m);
        memcpy(old_b, b, sizeof(double) * (size_t)n);
        orc_update_a_b(a, b, K, dx, dy, p, q, u, v, lambda1, lambda2, alpha1, alpha2, eps, m, n);

        int over = 0;
        for (int i = 0; i < m && !over; i++) over = a[i] > tau;
        for (int j = 0; j < n && !over; j++) over = b[j] > tau;
        if (over) {
            for (int i = 0; i < m; i++) u[i] = u[i] + eps * log(a[i]);
            for (int j = 0; j < n; j++) v[j] = v[j] + eps * log(b[j]);
            for (int i = 0; i < m; i++)
                for (int j = 0; j < n; j++) {
                    size_t t = (size_t)i * n + j;
                    K[t] = exp((u[i] + v[j] - C[t]) / eps);
                }
            for (int i = 0; i < m; i++) a[i] = 1.0;
            for (int j = 0; j < n; j++) b[j] = 1.0;
        }
        if (cur_iter >= max_iter) return -1;
    }
    return cur_iter;
}

/* ot_func.cpp:308-322.  lambda * sum dx (x log(x/p) - x + p) */
static double kl_div(double lam, const double *x, const double *p, const double *dx, int m)
{
    double r = 0.0;
    for (int i = 0; i < m; i++) r += dx[i] * (x[i] * log(x[i] / p[i]) - x[i] + p[i]);
    return lam * r;
}

/* ot_func.cpp:340-355.  lambda * sum p dx (exp(-eps log(x)/lambda) - 1) */
static double kl_conj_exp(double lam, double eps, const double *x, const double *p,
                          const double *dx, int m)
{
    double r = 0.0;
    for (int i = 0; i < m; i++) r += (p[i] * dx[i]) * (exp((-eps * log(x[i])) / lam) - 1.0);
    return lam * r;
}

/* ot_func.cpp:357-462.  Kbar is the un-stabilised Gibbs kernel exp(-C/eps). */
double orc_primal(const double *C, const double *Kbar, const double *R, const double *dx,
                  const double *dy, const double *p, const double *q, const double *a,
                  const double *b, double eps, double lambda1, double lambda2, int m, int n)
{
    (void)a; (void)b;
    double *rs = (double *)malloc(sizeof(double) * (size_t)m);
    double *cs = (double *)malloc(sizeof(double) * (size_t)n);
    mat_vec(rs, R, dy, m, n);
    matT_vec(cs, R, dx, m, n);
    size_t tot = (size_t)m * n;
    double ent = 0.0, cost = 0.0;
    for (size_t t = 0; t < tot; t++) ent += R[t] * clamp_inf(log(R[t])) - R[t] + Kbar[t];
    for (size_t t = 0; t < tot; t++) cost += R[t] * C[t];
    double ret = kl_div(lambda1, rs, p, dx, m) + kl_div(lambda2, cs, q, dy, n)
               + (eps * ent + cost) / (double)(m * n);
    free(rs);
    free(cs);
    return ret;
}

/* ot_func.cpp:464-490 */
double orc_dual(const double *C, const double *Kbar, const double *R, const double *dx,
                const double *dy, const double *p, const double *q, const double *a,
                const double *b, double eps, double lambda1, double lambda2, int m, int n)
{
    (void)C;
    size_t tot = (size_t)m * n;
    double am = 0.0;
    for (size_t t = 0; t < tot; t++) am += R[t] - Kbar[t];
    double t1 = -kl_conj_exp(lambda1, eps, a, p, dx, m);
    double t2 = -kl_conj_exp(lambda2, eps, b, q, dy, n);
    double t3 = -eps * am / (double)(m * n);
    return t1 + t2 + t3;
}

/* ot_func.cpp:492-544 */
double orc_duality_gap(const double *C, const double *Kbar, const double *R, const double *dx,
                       const double *dy, const double *p, const double *q, const double *a,
                       const double *b, double eps, double lambda1, double lambda2, int m, int n)
{
    double pri = orc_primal(C, Kbar, R, dx, dy, p, q, a, b, eps, lambda1, lambda2, m, n);
    double dua = orc_dual(C, Kbar, R, dx, dy, p, q, a, b, eps, lambda1, lambda2, m, n);
    return (pri - dua) / fabs(pri);
}

/* ot_func.cpp:830-930.  Loop `iters` scaling iterations (5, or batch_size in the last
 * epsilon stage) until the convergence measure drops to `threshold`:
 *   stages < last : max over {a,b} of ||x e^{w/eps} - x_old e^{w/eps}|| / (1 + ||x e^{w/eps}||)
 *   last stage    : R = a K b, true primal-dual gap with the un-stabilised scalings.
 * A NaN measure also ends the loop (NaN > thr is false).  `iters_done`, if not NULL,
 * receives the number of scaling iterations run (not part of the reference signature:
 * the oracle reports it so tests can compare iteration counts). */
double orc_update_process(double *R, double *a, double *b, double *old_a, double *old_b,
                          double *K, const double *Kbar, const double *C, const double *dx,
                          const double *dy, const double *p, const double *q, double *u,
                          double *v, int eps_scalings, int cur_scaling, int batch_size,
                          double eps, double threshold, double tau, double lambda1,
                          double lambda2, double alpha1, double alpha2, int cur_iter,
                          int max_iter, int m, int n, int *iters_done)
{
    double gap = 1e100;
    double *ta = (double *)malloc(sizeof(double) * (size_t)m);
    double *tb = (double *)malloc(sizeof(double) * (size_t)n);
    int done = 0;
    while (gap > threshold) {
        int iters = (cur_scaling == eps_scalings) ? batch_size : 5;
        cur_iter = orc_step1(a, b, old_a, old_b, K, C, dx, dy, p, q, u, v, cur_iter, max_iter,
                             iters, tau, lambda1, lambda2, alpha1, alpha2, eps, m, n);
        done += iters;
        for (int i = 0; i < m; i++) ta[i] = a[i] * exp(u[i] / eps);
        for (int j = 0; j < n; j++) tb[j] = b[j] * exp(v[j] / eps);
        if (cur_scaling == eps_scalings) {
            orc_update_R(R, K, a, b, m, n);
            gap = orc_duality_gap(C, Kbar, R, dx, dy, p, q, ta, tb, eps, lambda1, lambda2, m, n);
        } else {
            double d1 = 0.0, n1 = 0.0, d2 = 0.0, n2 = 0.0;
            for (int i = 0; i < m; i++) {
                double t = ta[i] - old_a[i] * exp(u[i] / eps);
                d1 += t * t;
            }
            for (int i = 0; i < m; i++) n1 += ta[i] * ta[i];
            for (int j = 0; j < n; j++) {
                double t = tb[j] - old_b[j] * exp(v[j] / eps);
                d2 += t * t;
            }
            for (int j = 0; j < n; j++) n2 += tb[j] * tb[j];
            double g1 = sqrt(d1) / (1.0 + sqrt(n1));
            double g2 = sqrt(d2) / (1.0 + sqrt(n2));
            /* std::max(v1, v2) == (v1 < v2) ? v2 : v1 -- a NaN v1 wins, a NaN v2 loses
             * (ot_func.cpp:922); keeps the reference's early exit on overflow. */
            gap = (g1 < g2) ? g2 : g1;
        }
    }
    free(ta);
    free(tb);
    if (iters_done) *iters_done = done;
    return gap;
}

/* ot_solvers.py:164-449 (C path: use_C, c_for_v2).  Whole 6-stage epsilon-scaling solve.
 *   p = G, q = mean(G), dx = 1/I, dy = 1/J, u = v = 0, a = b = 1
 *   eps_i runs epsilon0 * f^{-e}, f = exp(-ln(epsilon)/5), e = 0..5
 *   per stage: absorb a,b into u,v; a = b = 1; rebuild K, Kbar; run update_process
 * Output: plan[I*J] = R / J; stage_iters[6] = scaling iterations per stage; returns the
 * final gap (NaN means the reference would raise, ot_solvers.py:446-447). */
double orc_transport_duality_gap(double *plan, const double *C, const double *G, int I, int J,
                                 double lambda1, double lambda2, double epsilon, int batch_size,
                                 double tolerance, double tau, double epsilon0, int max_iter,
                                 int *stage_iters, double *u_out, double *v_out)
{
    const int S = 5;
    size_t tot = (size_t)I * J;
    double f = exp(-log(epsilon) / S);
    double *K = (double *)malloc(sizeof(double) * tot);
    double *Kbar = (double *)malloc(sizeof(double) * tot);
    double *R = (double *)calloc(tot, sizeof(double));
    double *vec = (double *)malloc(sizeof(double) * (size_t)(5 * I + 5 * J));
    double *dx = vec, *p = dx + I, *u = p + I, *a = u + I, *old_a = a + I;
    double *dy = old_a + I, *q = dy + J, *v = q + J, *b = v + J, *old_b = b + J;
    double gsum = 0.0;
    for (int i = 0; i < I; i++) gsum += G[i];
    for (int i = 0; i < I; i++) { dx[i] = 1.0 / I; p[i] = G[i]; u[i] = 0.0; a[i] = 1.0; }
    for (int j = 0; j < J; j++) { dy[j] = 1.0 / J; q[j] = gsum / I; v[j] = 0.0; b[j] = 1.0; }
    double eps_i = epsilon0 * f;
    double gap = INFINITY;
    for (int e = 0; e <= S; e++) {
        for (int i = 0; i < I; i++) { u[i] = u[i] + eps_i * log(a[i]); a[i] = 1.0; old_a[i] = 1.0; }
        for (int j = 0; j < J; j++) { v[j] = v[j] + eps_i * log(b[j]); b[j] = 1.0; old_b[j] = 1.0; }
        eps_i = eps_i / f;
        double alpha1 = lambda1 / (lambda1 + eps_i);
        double alpha2 = lambda2 / (lambda2 + eps_i);
        double thr = (e == S) ? tolerance : 1e-6;
        orc_update_k(K, Kbar, C, u, v, eps_i, I, J);
        memset(R, 0, sizeof(double) * tot);
        int it = 0;
        gap = orc_update_process(R, a, b, old_a, old_b, K, Kbar, C, dx, dy, p, q, u, v, S, e,
                                 batch_size, eps_i, thr, tau, lambda1, lambda2, alpha1, alpha2,
                                 0, max_iter, I, J, &it);
        if (stage_iters) stage_iters[e] = it;
    }
    for (size_t t = 0; t < tot; t++) plan[t] = R[t] / (double)J;
    if (u_out) memcpy(u_out, u, sizeof(double) * (size_t)I);
    if (v_out) memcpy(v_out, v, sizeof(double) * (size_t)J);
    free(K); free(Kbar); free(R); free(vec);
    return gap;
}
