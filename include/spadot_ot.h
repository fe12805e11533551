/*
 * spadot_ot.h -- C-ABI of libspadot_ot.so, the MI355X (gfx950) implementation of SpaDOT's
 * unbalanced entropic optimal-transport solver.
 *
 * PART A is a drop-in for the reference's libot.so: the same 15 extern "C" names, argument
 * order and in-place semantics as /root/reference/SpaDOT/utils/OT_loss/ot_func.cpp:938-1373,
 * which the reference binds with ctypes at ot_func.py:10-315.  All pointers in part A are
 * HOST pointers to caller-owned, C-contiguous, row-major (m, n) arrays; the library uploads
 * them, runs the HIP kernels on the current device's null stream, and writes every array the
 * reference mutates back in place.  Nothing is kept between calls.  There is no CPU path and no abort():
 * if no HIP device is usable, or a HIP call fails, the call prints a message on stderr and returns an error value --
 * NaN from the double / float valued entries (the reference's driver turns a NaN gap into a RuntimeError,
 * ot_solvers.py:446-447), -5 (SPADOT_EHIP) from step1_process_double and from every int-valued entry of part B,
 * NULL from the pointer-valued ones; bad arguments are -22.
 *
 * PART B is the device-resident solver the hot path actually uses: cost, kernel and scalings
 * stay in HBM across all six epsilon stages; only 8-byte convergence scalars cross PCIe.
 * Pointers named *_dev are DEVICE pointers; `stream` is a hipStream_t passed as void*.
 *
 * No torch types appear anywhere in this file.
 */
#ifndef SPADOT_OT_H
#define SPADOT_OT_H

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * PART A -- libot.so replacement (host pointers, fp64 live path + the float/dummy exports)
 * ---------------------------------------------------------------------------------------- */

/* ot_func.cpp:1203-1223 (update_k<double> :546-568).  K_ = exp(-C/eps), K = exp((u_i+v_j-C_ij)/eps). */
void update_k_double(double *K, double *K_, double *C, double *u, double *v, double epsilon, int m, int n);
/* ot_func.cpp:1181-1201 */
void update_k_float(float *K, float *K_, float *C, float *u, float *v, float epsilon, int m, int n);

/* ot_func.cpp:1243-1259 (update_R<double> :570-584).  R = diag(a) K diag(b). */
void update_R_double(double *R, double *K, double *a, double *b, int m, int n);
/* ot_func.cpp:1225-1241 */
void update_R_float(float *R, float *K, float *a, float *b, int m, int n);

/* ot_func.cpp:1261-1311 (step1_process<double> :689-828).  `iters` scaling iterations with the
 * tau-absorb; mutates a, b, old_a, old_b, K, u, v; returns cur_iter+iters, or -1 (after printing the
 * reference's message on stdout) once the counter reaches max_iter. */
int step1_process_double(double *a, double *b, double *old_a, double *old_b, double *K, double *C,
                         double *dx, double *dy, double *p, double *q, double *u, double *v,
                         int cur_iter, int max_iter, int iters, double tau, double lambda1,
                         double lambda2, double alpha1, double alpha2, double epsilon, int m, int n);

/* ot_func.cpp:1313-1373 (update_process<double> :830-930).  Runs scaling iterations until the
 * convergence measure is <= threshold (or NaN); mutates R (last stage only), a, b, old_a, old_b, K,
 * u, v; returns the measure.  The reference's Python shim passes m, n as surplus positional ints
 * after a 26-entry argtypes list (ot_func.py:286-313 vs :561-567); this signature is the C one. */
double update_process_double(double *R, double *a, double *b, double *old_a, double *old_b, double *K,
                             double *_K, double *C, double *dx, double *dy, double *p, double *q,
                             double *u, double *v, int epsilon_scalings, int cur_epsilon_scaling,
                             int batch_size, double epsilon, double threshold, double tau,
                             double lambda1, double lambda2, double alpha1, double alpha2,
                             int cur_iter, int max_iter, int m, int n);

/* ot_func.cpp:1011-1043, :1079-1111, :1147-1179 (primal :357-462, dual :464-490, gap :492-544).
 * `K` is the un-stabilised Gibbs kernel exp(-C/eps) the reference calls _K. */
double primal_double(double *C, double *K, double *R, double *dx, double *dy, double *p, double *q,
                     double *a, double *b, double epsilon, double lambda1, double lambda2, int m, int n);
double dual_double(double *C, double *K, double *R, double *dx, double *dy, double *p, double *q,
                   double *a, double *b, double epsilon, double lambda1, double lambda2, int m, int n);
double compute_duality_gap_double(double *C, double *K, double *R, double *dx, double *dy, double *p,
                                  double *q, double *a, double *b, double epsilon, double lambda1,
                                  double lambda2, int m, int n);
/* ot_func.cpp:977-1009, :1045-1077, :1113-1145 */
float primal_float(float *C, float *K, float *R, float *dx, float *dy, float *p, float *q, float *a,
                   float *b, float epsilon, float lambda1, float lambda2, int m, int n);
float dual_float(float *C, float *K, float *R, float *dx, float *dy, float *p, float *q, float *a,
                 float *b, float epsilon, float lambda1, float lambda2, int m, int n);
float compute_duality_gap_float(float *C, float *K, float *R, float *dx, float *dy, float *p, float *q,
                                float *a, float *b, float epsilon, float lambda1, float lambda2,
                                int m, int n);
/* ot_func.cpp:938-974: no-ops returning 0 (ctypes call-overhead probes in the reference). */
float dummy_float(float *C, float *K, float *R, float *dx, float *dy, float *p, float *q, float *a,
                  float *b, float epsilon, float lambda1, float lambda2, int m, int n);
double dummy_double(double *C, double *K, double *R, double *dx, double *dy, double *p, double *q,
                    double *a, double *b, double epsilon, double lambda1, double lambda2, int m, int n);

/* ------------------------------------------------------------------------------------------
 * PART B -- device-resident solver (replaces ot_solvers.py:164-449 + its 12 ctypes crossings)
 * ---------------------------------------------------------------------------------------- */

typedef struct spadot_ot_solver spadot_ot_solver;

enum { SPADOT_F64 = 0, SPADOT_F32 = 1 };

/* Scalar configuration = the ot_config keys the solver reads (config.yaml:39-57,
 * ot_solvers.py:164-179).  epsilon_scalings is fixed to 5 by the reference (:217). */
typedef struct spadot_ot_config {
    double lambda1, lambda2, epsilon, epsilon0, tolerance, tau;
    int batch_size;
    int max_iter;
} spadot_ot_config;

/* Per-solve report. */
typedef struct spadot_ot_info {
    double gap;            /* final convergence measure (NaN => the reference raises, ot_solvers.py:446) */
    int stage_iters[6];    /* scaling iterations run in each epsilon stage */
    int absorbs;           /* number of tau-stabilisations */
    int gap_checks;        /* convergence checks (host syncs) */
} spadot_ot_info;

/* Allocate a solver for an I x J problem whose kernel matrix is stored as `storage`
 * (SPADOT_F64: reference arithmetic; SPADOT_F32: fp32 K/C in HBM, fp64 scalings and sums).
 * Returns 0, or a negative errno-style code; *out is NULL on failure. */
int spadot_ot_create(spadot_ot_solver **out, int I, int J, int storage, void *stream);
void spadot_ot_destroy(spadot_ot_solver *s);

/* Leading dimension (elements) of the solver's internal I x ld matrices, and raw device views of
 * its state for zero-copy consumers/tests.  which: 0=C 1=K ; vectors: 0=a 1=b 2=u 3=v 4=old_a 5=old_b */
int spadot_ot_ld(const spadot_ot_solver *s);
/* Fused single-sweep geometry: out[4] = {vectors per thread, rows per group, workgroups, rows per
 * workgroup}; all 0 when the two-sweep kernels are in use. */
void spadot_ot_fused_geometry(const spadot_ot_solver *s, int *out);
void *spadot_ot_matrix_dev(spadot_ot_solver *s, int which);
double *spadot_ot_vector_dev(spadot_ot_solver *s, int which);
/* Host copy of one state vector (I or J doubles), synchronises the solver's stream. */
int spadot_ot_vector_host(spadot_ot_solver *s, int which, double *out);
/* Host fp64 copy (I x J, contiguous) of the cost (which=0) or kernel (which=1) matrix. */
int spadot_ot_matrix_host(spadot_ot_solver *s, int which, double *out);

/* Cost matrix from a device array (row-major, leading dim ldc elements, dtype SPADOT_F64/F32). */
int spadot_ot_set_cost_dev(spadot_ot_solver *s, const void *C_dev, int dtype, int ldc);
/* Cost matrix from a host fp64 array (row-major I x J, contiguous). */
int spadot_ot_set_cost_host(spadot_ot_solver *s, const double *C_host);
/* Cost from latents: C_ij = |x_i - y_j|^2, optionally divided by median(C) -- what
 * ot_solvers.py:101-103 does with sklearn + numpy.  x_dev (I x d), y_dev (J x d) fp64 row-major. */
int spadot_ot_set_cost_from_latents_dev(spadot_ot_solver *s, const double *x_dev, const double *y_dev,
                                        int d, int divide_by_median);

/* One whole solve (6 epsilon stages).  G_host: growth vector (I doubles) or NULL for ones.
 * Returns 0 on success, 1 if the final gap is NaN (the reference raises RuntimeError there). */
int spadot_ot_solve(spadot_ot_solver *s, const double *G_host, const spadot_ot_config *cfg,
                    spadot_ot_info *info);

/* Transport plan R/J (ot_solvers.py:449) of the last solve. */
int spadot_ot_plan_dev(spadot_ot_solver *s, void *plan_dev, int dtype, int ldp);
int spadot_ot_plan_host(spadot_ot_solver *s, double *plan_host);
/* Column-group sums of the plan R/J without materialising it: Q_dev[i*ngroups + g] = sum over columns j with
 * col_labels_dev[j] == g of plan[i][j] (labels int32 in [0, ngroups), ngroups <= 64, Q fp64 I x ngroups).
 * one-hot(row labels)^T Q is the cluster transition table of the analyze stage (_analyze_utils.py:131-137). */
int spadot_ot_plan_group_sums_dev(spadot_ot_solver *s, const int *col_labels_dev, int ngroups, double *Q_dev);

/* Row sums of the plan (what compute_transport_map feeds back as growth, ot_solvers.py:117). */
int spadot_ot_plan_rowsums_host(spadot_ot_solver *s, double *rowsums_host);

/* Benchmark hook: run `iters` scaling iterations (one iteration = one update_a_b, ot_func.cpp:586-687)
 * at the solver's current state with the given stage epsilon, no convergence checks, no host syncs.
 * ms_out (may be NULL) receives the HIP-event time of the timed region in milliseconds.  Returns 2 if a
 * scaling exceeded tau during the run (the fast schedule is then not what a solve would execute).
 * With ms_out == NULL nothing is read back and the tau flag is NOT cleared either: it accumulates over consecutive
 * untimed calls and spadot_ot_run_tau_flag() reads it once behind them. */
int spadot_ot_run_iterations(spadot_ot_solver *s, const spadot_ot_config *cfg, double eps_stage,
                             int iters, float *ms_out);

/* The tau flag spadot_ot_run_iterations raises (synchronises the solver's stream): 1 if a scaling exceeded tau since the
 * flag was last cleared, 0 if not, negative on error.  reset != 0 clears it afterwards. */
int spadot_ot_run_tau_flag(spadot_ot_solver *s, int reset);

/* The solver's real inner loop `nbatches` times (snapshot, batch_size or 5 iterations, convergence measure,
 * read-back + stream sync), ignoring the threshold: what "iterations per second" costs inside a solve.
 * last_stage != 0 uses the duality-gap measure, else the dual-drift measure. */
int spadot_ot_run_checked(spadot_ot_solver *s, const spadot_ot_config *cfg, double eps_stage, int last_stage,
                          int nbatches, int *iters_out, float *ms_out);

/* Per-kernel live timing for the roofline: each kernel of one scaling iteration launched `reps` times
 * between two HIP events on the solver's stream.  ms_out[6] = average ms per launch of
 * {row pass, column pass, column finalise, idle tau-absorb pair, fused single-sweep pass, fused
 * column finalise}; the last two are 0 when the shape does not admit the fused path. */
int spadot_ot_time_kernels(spadot_ot_solver *s, const spadot_ot_config *cfg, double eps_stage, int reps,
                           float *ms_out);

/* ------------------------------------------------------------------------------------------
 * PART C -- whole solves of SMALL problems (I, J <= spadot_ot_small_max() = 64), batched: one launch, one
 * wavefront per problem, everything in LDS, NO host synchronisation.  This is the solve the training loop runs:
 * _update_OT_matrix (_train_utils.py:309-321) couples the 10 x 10 K-means centres of consecutive time points through
 * compute_transport_map (ot_solvers.py:95-121: cost = sqeuclidean / median, first growth solve returned) and
 * optimal_transport_duality_gap (:164-449); all T - 1 pairs go into one call.  fp64 throughout.
 * ---------------------------------------------------------------------------------------- */

typedef struct spadot_ot_small_problem {
    const double *x_dev;       /* I x d latents, row-major fp64 (device); NULL => the cost comes from C_dev */
    const double *y_dev;       /* J x d */
    const double *C_dev;       /* I x J cost, row-major fp64 (device); read only when x_dev == NULL */
    const double *G_dev;       /* growth p (I doubles, device) or NULL for ones; q = mean(G) (ot_solvers.py:223-224) */
    double *plan_dev;          /* out (may be NULL): I x J plan R / J (ot_solvers.py:449) */
    float *gamma_rownorm_dev;  /* out (may be NULL): I x J fp32, rows of the plan normalised to sum 1, NaN / inf -> 0:
                                  what _compute_OT_loss makes of it every step (_train_utils.py:299-300) */
    int I, J;
} spadot_ot_small_problem;

typedef struct spadot_ot_small_info {
    double gap;                /* final convergence measure; NaN => the reference raises (ot_solvers.py:446-447) */
    int stage_iters[6];        /* scaling iterations run in each epsilon stage */
    int absorbs;               /* tau-stabilisations */
    int gap_checks;            /* convergence checks */
    int status;                /* bit 0: a stage reached max_iter (the reference prints and goes on, ot_func.cpp:821-824);
                                  bit 1: a stage passed 2^20 iterations and the kernel gave up (plan not converged) */
    int reserved;
} spadot_ot_small_info;

int spadot_ot_small_max(void);

/* Enqueues the solves of `nprob` problems on `stream` (a hipStream_t) and returns at once: nothing is allocated, nothing
 * is synchronised.  `probs` is a HOST array (the descriptors travel as kernel arguments; the pointers inside are device
 * pointers), d the latent dimension (<= 32), divide_by_median as in spadot_ot_set_cost_from_latents_dev, info_dev a
 * DEVICE array of nprob records or NULL.  Returns 0, -22 for a bad argument (a size outside 1..64 included), -5 when no HIP
 * device is usable or the launch fails. */
int spadot_ot_small_solve(int nprob, const spadot_ot_small_problem *probs, int d, int divide_by_median,
                          const spadot_ot_config *cfg, spadot_ot_small_info *info_dev, void *stream);

/* Library/version probe. */
const char *spadot_ot_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SPADOT_OT_H */
