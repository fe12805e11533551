/*
 * spadot_model.h -- C-ABI of libspadot_model.so: the hand-written HIP kernels (gfx950) behind the
 * model side of SpaDOT's training step.  All pointers are DEVICE pointers unless named *_host;
 * `stream` is a hipStream_t passed as void*; every function returns 0 or a negative errno-style
 * code and only ENQUEUES work (no host synchronisation).  No torch types appear here; the Python
 * host (spadot_amd/model) passes tensor.data_ptr() values.
 *
 * Reference interfaces replaced (all under /root/reference/SpaDOT):
 *   spadot_gat_*       torch_geometric GATConv's edge phase as called at model/encoder.py:56-58
 *                      (scatter-softmax over incoming edges + weighted scatter-add)
 *   spadot_kernel_matrix   model/svgp.py:110-125 (Kernel.forward: cdist + Gaussian/Cauchy/Quadratic)
 *   spadot_rowdot_*    the diag(A B^T) pattern of svgp.py:80, :98 and the trace identity replacing
 *                      the (b,m,m) tensor of svgp.py:99-101
 *   spadot_elbo_*      the scalar reductions of svgp.py:102-104 and model/SpaDOT.py:67-77,:125-142
 *   spadot_vae_*       model/SpaDOT.py:78-93 (reparameterisation, GAT KL, reconstruction, alignment)
 *   spadot_kmeans_*    utils/_train_utils.py:240-253 (loss) and sklearn KMeans.predict (labels, :266-269)
 *   spadot_adamw_*     utils/_train_utils.py:214-217 (clip_grad_norm_(0.3) + AdamW.step)
 */
#ifndef SPADOT_MODEL_H
#define SPADOT_MODEL_H

#ifdef __cplusplus
extern "C" {
#endif

enum { SPADOT_DT_F32 = 0, SPADOT_DT_BF16 = 1, SPADOT_DT_F64 = 2 };
enum { SPADOT_KERNEL_GAUSSIAN = 0, SPADOT_KERNEL_CAUCHY = 1, SPADOT_KERNEL_QUADRATIC = 2 };

const char *spadot_model_version(void);

/* ---------------------------------------------------------------- GAT edge phase
 * Graph in CSR by TARGET node: rowptr[n+1], col[E] = source node of each incoming edge (self loops
 * already present exactly once per node, as GATConv(add_self_loops=True) leaves them).
 * h: [n, H*C] features after the layer's linear map (dtype = SPADOT_DT_F32 or _BF16);
 * s_src, s_dst: [n, H] fp32 attention logits (h . att_src, h . att_dst per head).
 *
 * forward:  e_ij = leaky_relu(s_src[j] + s_dst[i], 0.2); alpha = softmax over incoming edges of i
 *           (exp(e - max) / (sum + 1e-16)); agg_i = sum_j alpha_ij h_j
 *           concat != 0: out[i, H*C] = act(agg + bias[H*C])      (act = leaky_relu(0.01) if act != 0)
 *           concat == 0: out[i, C]   = act(mean_h agg + bias[C])
 *           alpha_out [E, H] fp32 is kept for the backward pass.
 */
int spadot_gat_forward(const void *h, int dtype, const float *s_src, const float *s_dst,
                       const int *rowptr, const int *col, const float *bias, int n, int H, int C,
                       int concat, int act, void *out, float *alpha_out, void *stream);

/* backward, target-side half: from g_out (gradient w.r.t. `out`, same shape/dtype as out) and the
 * saved forward tensors computes
 *   g_pre [n, H*C] (dtype) = gradient w.r.t. agg (activation and head-mean undone)
 *   dz    [E, H] fp32      = gradient w.r.t. the pre-leaky logits s_src[j] + s_dst[i]
 *   ds_dst[n, H] fp32      = sum over incoming edges of dz
 */
int spadot_gat_backward_target(const void *g_out, const void *out, const void *h, int dtype,
                               const float *s_src, const float *s_dst, const float *alpha,
                               const int *rowptr, const int *col, int n, int H, int C, int concat,
                               int act, void *g_pre, float *dz, float *ds_dst, void *stream);

/* backward, source-side half (no atomics): transposed CSR rowptr_t[n+1], col_t[E] = TARGET of each
 * outgoing edge, eid_t[E] = position of that edge in the target-ordered arrays (alpha, dz).
 *   dh[j]     = sum_{j->i} alpha_ij g_pre[i]      [n, H*C] (dtype)
 *   ds_src[j] = sum_{j->i} dz_ij                  [n, H] fp32
 */
int spadot_gat_backward_source(const void *g_pre, int dtype, const float *alpha, const float *dz,
                               const int *rowptr_t, const int *col_t, const int *eid_t, int n, int H,
                               int C, void *dh, float *ds_src, const float *ds_dst, const float *att_src,
                               const float *att_dst, void *stream);
/*   If att_src / att_dst ([H, C] fp32) are given (not NULL), the logits were produced by spadot_gat_logits
 *   and their gradient is folded into dh in the same pass:
 *       dh[j,h,c] += ds_src[j,h] att_src[h,c] + ds_dst[j,h] att_dst[h,c]        (ds_dst from the target half). */

/* Attention logits from h: s_src[j,h] = sum_c h[j,h,c] att_src[h,c], s_dst[j,h] likewise (h read once). */
int spadot_gat_logits(const void *h, int dtype, const float *att_src, const float *att_dst, int n, int H,
                      int C, float *s_src, float *s_dst, void *stream);
/* Gradient of the attention vectors: datt_src[h,c] = sum_j ds_src[j,h] h[j,h,c] (datt_dst with ds_dst).
 * datt_src and datt_dst must be the two halves of ONE [2, H*C] fp32 buffer (datt_dst == datt_src + H*C);
 * scratch: fp32 work space of scratch_floats >= 2*H*C entries (more = more parallel slabs).
 * With g_pre ([n_pre, H*C], n_pre <= n; else NULL) the buffer is [3, H*C] (scratch >= 3*H*C) and its third
 * block receives sum_i g_pre[i, :], the bias gradient before any head reduction, from the same pass. */
int spadot_gat_att_grad(const void *h, int dtype, const float *ds_src, const float *ds_dst, int n, int H,
                        int C, float *scratch, int scratch_floats, float *datt_src, float *datt_dst,
                        const void *g_pre, int n_pre, void *stream);

/* out[c] = sum_r x[r, c] for a row-major fp32 [rows, width] array, rows added in a fixed order (bias gradients of the
 * dense maps: no semaphore-based library reduction inside the replayed graphs). */
int spadot_colsum(const float *x, int rows, int width, float *out, void *stream);

/* ---------------------------------------------------------------- SVGP pieces */

/* K[i,j] = k(|x_i - z_j|^2 / scale), x [n,d], z [m,d], K [n,m]; dtype F32 or F64 for all three.
 * Gaussian exp(-d2/scale); Cauchy 1/(1+d2/scale); Quadratic 1 - d2/(d2+scale)  (svgp.py:116-124). */
int spadot_kernel_matrix(const void *x, const void *z, int n, int m, int d, double scale, int kind,
                         int dtype, void *K, void *stream);

/* Batched SPD inverse and log-determinant (fp64): A [L, m, m] symmetric positive definite ->
 * Ainv [L, m, m], logdet [L].  One workgroup per matrix, symmetric sweep operator, matrix resident in
 * registers + LDS (m <= 310; returns -34 beyond: the host splits larger matrices into blocks, spadot_amd/ops.py).  Replaces the
 * torch.linalg.inv / cholesky calls of svgp.py:50,75,87-88 on the L latent dimensions at once. */
int spadot_spd_inverse_logdet(const double *A, int L, int m, double *Ainv, double *logdet, void *stream);
/* Same for L matrices read as A[b mod Lsrc] + add0 (+ add1 for b >= Lsrc) (add0 / add1: m x m or NULL; Lsrc = 0: plain).
 * The SVGP step stores G_l = c K_mn diag(1/var_l) K_nm once and inverts Sigma_l = G_l + (K_mm + jI) and
 * Sigma_l + K_mm^2 / j (svgp.py:65-75 and the Sylvester form of :88) for every latent dimension in this one launch. */
int spadot_spd_inverse_logdet2(const double *A, int Lsrc, int L, int m, const double *add0, const double *add1, double *Ainv,
                               double *logdet, void *stream);

/* Batched row-wise dot products: out[l, i] = sum_k A[l, i, k] * B[i, k]   (A [L,n,m], B [n,m]).
 * dtype F32 or F64. */
int spadot_rowdot_forward(const void *A, const void *B, int L, int n, int m, int dtype, void *out,
                          void *stream);
/* gA[l,i,k] = g[l,i] * B[i,k]   (B is a constant of the step: no gradient) */
int spadot_rowdot_backward(const void *g, const void *B, int L, int n, int m, int dtype, void *gA,
                           void *stream);

/* ELBO scalar reductions over a [b, L] batch (row-major, L latent dims):
 *   l3 = -0.5 * sum_{i,l} [ (ktilde_i + tr_{i,l}) / var_{i,l} + log var_{i,l} + log(2 pi)
 *                           + (mu_{i,l} - mv_{i,l})^2 / var_{i,l} ]                (svgp.py:97-104)
 *   ce = sum_{i,l} -0.5 * [ log(2 pi) + log var + (pv + pm^2 - 2 pm mu + mu^2) / var ]  (SpaDOT.py:125-142)
 * out2[0] = l3, out2[1] = ce (fp64 accumulators written as `dtype`).
 * inputs: mu, var (encoder mean / variance), mv (K_nm K_mm^-1 mu_hat), tr (trace term), pm, pv
 * (posterior mean / variance): all [b, L]; ktilde [b]. */
int spadot_elbo_forward(const void *mu, const void *var, const void *mv, const void *tr, const void *pm,
                        const void *pv, const void *ktilde, int b, int L, int dtype, void *out2,
                        void *stream);
/* element-wise gradients given g2 = (dLoss/dl3, dLoss/dce) on the device: g_mu, g_var, g_mv, g_tr, g_pm,
 * g_pv, each [b, L]. */
int spadot_elbo_backward(const void *g2, const void *mu, const void *var, const void *mv, const void *tr,
                         const void *pm, const void *pv, const void *ktilde, int b, int L, int dtype,
                         void *g_mu, void *g_var, void *g_mv, void *g_tr, void *g_pm, void *g_pv,
                         void *stream);

/* ---------------------------------------------------------------- small-MLP stages (encoder.py:7-34, decoder.py:3-20)
 * BatchNorm1d in training mode + LeakyReLU(slope) in one launch: y = leaky((x + lin_bias - mean) invstd gamma + beta);
 * x [b, F] fp32 or bf16 (the preceding Linear's output WITHOUT its bias; lin_bias [F] or NULL is added here), batch
 * statistics over the b rows, running_mean / running_var / num_batches_tracked updated as nn.BatchNorm1d does
 * (momentum; unbiased running variance).  save_mean / save_invstd [F] feed the backward, which returns dx (dtype of x),
 * dgamma, dbeta (the gradient of lin_bias is identically zero: the batch mean removes it).
 * LayerNorm over the F features + LeakyReLU likewise (fp32; save_mean / save_invstd [b]). */
int spadot_bn_act_forward(const void *x, int x_dtype, const float *lin_bias, const float *gamma, const float *beta,
                          float *running_mean, float *running_var, long long *num_batches_tracked, int b, int F,
                          double momentum, double eps, double slope, float *y, float *save_mean, float *save_invstd,
                          void *stream);
int spadot_bn_act_backward(const float *dy, const float *y, const void *x, int x_dtype, const float *lin_bias,
                           const float *gamma, const float *save_mean, const float *save_invstd, int b, int F,
                           double slope, void *dx, float *dgamma, float *dbeta, void *stream);
int spadot_ln_act_forward(const float *x, const float *gamma, const float *beta, int b, int F, double eps, double slope,
                          float *y, float *save_mean, float *save_invstd, void *stream);
int spadot_ln_act_backward(const float *dy, const float *y, const float *x, const float *gamma, const float *save_mean,
                           const float *save_invstd, int b, int F, double slope, float *dx, float *dgamma, float *dbeta,
                           void *stream);

/* SVGP branch after the batched inverse, all latent dimensions, fp64 (svgp.py:62-104, SpaDOT.py:72-77).
 * forward: raw = X2 r^T [2b, L] (X2 = [K_nm; K_nm K^-1 K_mm]), rd = rowdot(X2 S, X2) [L, 2b], r and Mr = r M [L, m],
 * ld [2L] log-determinants, sm [L] = <S_l, M>, mu / var [b, L] encoder output, ktilde [b]  ->  p_m, mv, p_v, tr [b, L]
 * and out4 = (l3, ce, KL sum, SVGP_KL = -|ce - (l3 - (b/N) KL)| / L); kl_const = log|K| - m log j - m.
 * backward: g_skl fp32 device scalar, G_pm / G_pv upstream gradients or NULL -> direct g_mu, g_var [b, L], the operands
 * G1 [2b, L] and G2T [L, 2b] of the algebra's backward, g_kl, gMr = g_kl c^2 Mr [L, m], gM = g_kl/2 M [m, m].
 * grad_tail: the last element-wise step of that backward (see model/svgp.py:_SVGPCore). */
int spadot_svgp_post_forward(const double *raw, const double *rd, const double *r, const double *Mr, const double *ld,
                             const double *sm, const double *mu, const double *var, const double *ktilde, int b, int L,
                             int m, double c, double kl_const, double b_over_N, double *p_m, double *mv, double *p_v,
                             double *tr, double *out4, float *skl32 /* may be NULL: SVGP_KL once more in fp32 */, void *stream);
int spadot_svgp_post_backward(const float *g_skl, const double *out4, const double *G_pm, const double *G_pv,
                              const double *mu, const double *var, const double *mv, const double *tr, const double *p_m,
                              const double *p_v, const double *ktilde, const double *Mr, const double *M, int b, int L,
                              int m, double c, double b_over_N, double *g_mu, double *g_var, double *G1, double *G2T,
                              double *g_kl, double *gMr, double *gM, void *stream);
/* q1[l, i] = sum_n G2T[l, n] T[l, n, i]^2 + g_kl / 2 * m0[l, i]: the diag(K_nm dSigma K_mn) term of _SVGPCore.backward
 * (svgp.py:62-104 differentiated) from T_l = X2 S_l K_mn, given as its two halves Ta = K_nm S_l K_mn and Tb = P S_l K_mn
 * (each [L, nh, b]; G2T [L, 2 nh]), and m0 [L, b], all formed ahead of the backward pass. */
int spadot_svgp_q1t(const double *Ta, const double *Tb, const double *G2T, const double *m0, const double *g_kl, int L, int nh,
                    int b, double *q1, void *stream);
/* p_m = c raw[:b] and p_v = k~ + rd_a^T (rd_a [L, b] = diag(K_nm S_l K_mn)): what the loss tail needs of
 * spadot_svgp_post_forward (svgp.py:62-84), which may then run later. */
int spadot_svgp_post_pm_pv(const double *raw, const double *rd_a, const double *ktilde, int b, int L, double c, double *p_m,
                           double *p_v, void *stream);
int spadot_svgp_grad_tail(const double *q1, const double *q2, const double *Kdt, const double *p_v, const double *ktilde,
                          const double *p_m, const double *mu, const double *w, const double *g_kl, const double *g_mu,
                          const double *g_var, int b, int L, double c, double *dmu, double *dvar, float *dz,
                          void *stream);
/* z [b, 2L] fp32 = SVGP_fc output (mu | logvar) -> mu, var = exp(logvar), w = 1/var, mu w, each [b, L] fp64
 * (encoder.py:31-34 + the first element-wise steps of svgp.py:62-70).  grad_tail's dz [b, 2L] fp32 is the matching
 * gradient (d/dlogvar = d/dvar * var); dmu/dvar may then be NULL. */
int spadot_svgp_pre(const float *z, int b, int L, double *mu, double *var, double *w, double *muw, void *stream);
/* ... and A[l, i, :] = K_nm[i, :] / var[i, l]  ([L, b, m] fp64) in the same launch */
int spadot_svgp_pre2(const float *z, const double *Kn, int b, int L, int m, double *mu, double *var, double *w, double *muw,
                     double *A, void *stream);
/* The small products behind the inverse for all L latent dimensions in two launches: r_l = S_l t_l, Mr_l = M r_l,
 * raw[:, l] = X2 r_l (X2 [rows2, m]) and sm_l = <S_l, M>; smpart: scratch of >= L * 4 * ceil(m / 4) doubles. */
int spadot_svgp_mid(const double *S, const double *t, const double *M, const double *X2, int L, int m, int rows2,
                    double *r, double *Mr, double *raw, double *sm, double *smpart, int smpart_doubles, void *stream);

/* ---------------------------------------------------------------- loss tail of a training step
 * Single-workgroup kernels for the b x 20 / 10 x 10 arithmetic after the encoders (each replaces a few dozen
 * library launches; reductions in a fixed order, fp64 accumulators).
 *
 * latent head (SpaDOT.py:78-93): zg [b, 2 Lg] fp32 = GAT_fc output (mu | logvar), p_m / p_v [b, Ls] fp64 = SVGP
 * posterior, eps [b, Ls+Lg] fp32 standard normal.  latent [b, Ls+Lg] = (p_m + eps sqrt(p_v) | mu + eps
 * sqrt(exp(logvar))); scal2 = (GAT_KL = -1/2 sum(1 + logvar - mu^2 - var) / Lg,
 * alignment = sum_i (|s_i| / Ls - |g_i| / Lg)^2); Ls + Lg <= 32; partials: 2 * ceil(b/8) doubles of work space, counter: one
 * unsigned that is 0 before the first launch (the kernel leaves it 0).  rng_state NULL: eps is an input; else eps is
 * WRITTEN by the kernel (counter-based standard normals from rng_state = (seed, launch count); the count advances by
 * one per launch), so that no library RNG launch sits in the replayed graph.  backward: g_latent [b, Ls+Lg] or NULL, g_kl / g_align device
 * scalars or NULL -> d_zg [b, 2 Lg], d_pm, d_pv [b, Ls] fp64. */
int spadot_latent_head_forward(const float *zg, const double *p_m, const double *p_v, float *eps, int b, int Ls,
                               int Lg, float *latent, float *scal2, double *partials, unsigned *counter,
                               unsigned long long *rng_state, void *stream);
int spadot_latent_head_backward(const float *zg, const double *p_v, const float *eps, const float *latent,
                                const float *g_latent, const float *g_kl, const float *g_align, int b, int Ls, int Lg,
                                float *d_zg, double *d_pm, double *d_pv, void *stream);
/* K-means loss (_train_utils.py:240-253) and OT loss (_train_utils.py:272-307) of one batch.  z [b, D] fp32;
 * labels_all[seed_ids[i]] (int64) = cluster of seed i; centres [K, D]; prev_centres [Kp, D]; gamma [Kp, Kl]
 * row-normalised plan; cluster_list [Kl] int64 = clusters of the time point, ascending.  K <= 64, D <= 64.
 * out2 = (||z - c[label]||^2 / D / #distinct labels, mean(gamma * cdist(prev, batch means or stored centre))).
 * work: K*D + K + 1 + b floats, written by forward, read by backward.  do_km / do_ot switch the terms. */
int spadot_cluster_losses_forward(const float *z, const long long *labels_all, const long long *seed_ids,
                                  const float *centres, const float *prev_centres, const float *gamma,
                                  const long long *cluster_list, int b, int D, int K, int Kp, int Kl, int do_km,
                                  int do_ot, float *out2, float *work, void *stream);
int spadot_cluster_losses_backward(const float *z, const float *centres, const float *prev_centres, const float *gamma,
                                   const long long *cluster_list, const float *work, const float *g_km,
                                   const float *g_ot, int b, int D, int K, int Kp, int Kl, int do_km, int do_ot,
                                   float *dz, void *stream);
/* forward AND dz = d(g_km[0] * km + g_ot[0] * ot) / dz in ONE launch, for gradient seeds known when the forward runs (the loss
 * weights of _train_utils.py:205-212, device scalars).  Same arithmetic as forward + backward.  Returns -95 when the shape
 * does not take the kernel's one-chunk fast path (K <= 16, b * D <= 10240): call forward / backward then. */
int spadot_cluster_losses_fb(const float *z, const long long *labels_all, const long long *seed_ids, const float *centres,
                             const float *prev_centres, const float *gamma, const long long *cluster_list, int b, int D, int K,
                             int Kp, int Kl, int do_km, int do_ot, const float *g_km, const float *g_ot, float *out2,
                             float *work, float *dz, void *stream);
/* elbo = sum_k w6[k] * *terms6[k] (_train_utils.py:205-212); out7 [8 floats] = (elbo, the six terms, elbo again).
 * terms6 is a HOST array of six device pointers.  backward: g6[k] = g_elbo[0] * w6[k]. */
int spadot_mix_losses_forward(const float *const *terms6, const float *w6, float *out7, void *stream);
int spadot_mix_losses_backward(const float *g_elbo, const float *w6, float *g6, void *stream);

/* ---------------------------------------------------------------- VAE head losses (SpaDOT.py:78-93)
 * out1[0] = inv_scale * sum_k (y[k] - yhat[k])^2 over `count` elements (recon: inv_scale = 1/G),
 * deterministic two-stage reduction through `scratch` (>= 2048 doubles).
 * backward writes g_yhat[k] = -2 inv_scale (y[k] - yhat[k]) * g1[0]. */
int spadot_sqerr_forward(const void *y, const void *yhat, long long count, double inv_scale, int dtype,
                         double *scratch, void *out1, void *stream);
int spadot_sqerr_backward(const void *g1, const void *y, const void *yhat, long long count,
                          double inv_scale, int dtype, void *g_yhat, void *stream);

/* ---------------------------------------------------------------- k-means glue */
/* labels[i] = argmin_c |x_i - c_c|^2 (first minimum wins), x [n,d], centers [k,d], int32 labels.
 * Distances are accumulated in fp64 whatever `dtype` (F32/F64) the inputs have. */
int spadot_kmeans_assign(const void *x, const void *centers, int n, int k, int d, int dtype, int *labels,
                         void *stream);

/* Small fp32 products of the MLP stages with a small footprint (32 x 32 tiles, 256 threads, 8 KB of LDS): they find room on a
 * compute unit beside the GAT branch's GEMMs, where the library's 256 x 64 macro tiles waited 150-185 us for a whole unit.
 *   C [M x N] (row stride ldc) = sum_k a(m, k) b(k, n) (+ bias[n] when bias != NULL), k ascending (fixed order):
 *   mode 0: a = A[m * lda + k], b = B[k * ldb + n]   (dx = g W)
 *   mode 1: a = A[m * lda + k], b = B[n * ldb + k]   (y = x W^T + bias: the forward map of nn.Linear)
 *   mode 2: a = A[k * lda + m], b = B[k * ldb + n]   (dW = g^T x)
 * `batch` such products in one launch, entry z using A + z strideA, B + z strideB, C + z strideC (elements): a long
 * contraction (mode 2 over hundreds of rows) is cut into row slices whose partial results the caller adds in slice order.
 * Meant for products of at most a few hundred MFLOP. */
int spadot_sgemm_small(int mode, const float *A, int lda, const float *B, int ldb, float *C, int ldc, int M, int N, int K,
                       const float *bias, int batch, long long strideA, long long strideB, long long strideC, void *stream);

/* ---- The SVGP encoder behind its first map as three launches (csrc/enc_fused.hip, round 5; encoder.py:7-34 in training mode) ----
 * h1 [b x F1] (fp32, the first map's output without its bias) -> BatchNorm1d + LeakyReLU -> y1 -> hidden map -> h2 [b x F2] ->
 * BatchNorm1d + LeakyReLU -> y2 -> SVGP_fc -> z [b x Q].  The two maps cross workgroups as PARTIAL PRODUCTS in a caller-owned
 * workspace (spadot_enc_fused_workspace floats: part [F1 / 16][b][F2], then pz [F2 / 4][b][Q]), summed in group order by the
 * next launch: spadot_enc_bn_map (statistics of 16 columns of h1, y1, running statistics, part), spadot_enc_bn_fc (h2 = sum of
 * part -- stored without the map's bias, which BatchNorm's lin_bias carries --, y2, running statistics, pz), and either
 * spadot_enc_sum_z (z = bias + sum of pz) or spadot_svgp_pre2_partials (spadot_svgp_pre2 reading z from pz; it also stores z).
 * b <= 512, F1 % 16 == 0, F2 % 4 == 0, F2 <= 128, Q <= 32.  y1 is bit for bit what spadot_bn_act_forward writes. */
int spadot_enc_fused_supported(int b, int F1, int F2, int Q);
long long spadot_enc_fused_workspace(int b, int F1, int F2, int Q);
int spadot_enc_bn_map(const float *h1, const float *lin_bias, const float *gamma, const float *beta, float *running_mean,
                      float *running_var, long long *num_batches_tracked, int b, int F1, double momentum, double eps, double slope,
                      float *y1, float *save_mean, float *save_invstd, const float *W2, int F2, float *part, void *stream);
int spadot_enc_bn_fc(const float *part, int nparts, const float *lin_bias, const float *gamma, const float *beta, float *running_mean,
                     float *running_var, long long *num_batches_tracked, int b, int F2, double momentum, double eps, double slope,
                     float *h2, float *y2, float *save_mean, float *save_invstd, const float *Wfc, int Q, float *pz, void *stream);
int spadot_enc_sum_z(const float *pz, int nparts, const float *bias, int b, int Q, float *z, void *stream);
int spadot_svgp_pre2_partials(const float *pz, int nparts, const float *bias, const double *Kn, int b, int L, int m, float *z,
                              double *mu, double *var, double *w, double *muw, double *A, void *stream);

/* ---- The decoder's output map, the reconstruction term and their backward as one launch (csrc/recon_fb.hip, round 5) ----------
 * decoder.py:3-20 (last Linear: hidden K = 256 -> G) + SpaDOT.py:89 (recon = inv_scale * sum (y - (h W^T + bias))^2) forward AND
 * backward for a gradient seed known at forward time: grad_weight[0] = d loss / d recon (the loss weight lambda1,
 * _train_utils.py:205-212, a device scalar).  h_bf16 [b x K], W_bf16 [G x K] (bf16 images of h and of the weight), y [b x G] fp32.
 * Leaves: g_bf16 [b x G] = -2 inv_scale grad_weight (y - o - bias) in bf16 (the weight gradient g^T h is the caller's GEMM),
 * dh [b x K] fp32 = g W (gene blocks summed in block order), in `workspace` (spadot_recon_fb_workspace floats) behind the
 * per-gene-block partials of dh the bias-gradient partials dbp [ceil(b / 64)][G] (sum them: spadot_colsum), and in loss_parts
 * (ceil(b / 64) * ceil(G / 128) doubles) the partial sums of (y - o - bias)^2 (spadot_sum_parts with scale = inv_scale gives the
 * term's value).  The fp32 image of o = h W^T is never formed.  WT_bf16 [K x ldt]: the TRANSPOSED bf16 image of the weight, rows
 * padded to ldt >= G rounded up to 128 (the second product reads it like the first reads W: 16 bytes per lane straight from
 * memory, no weight tile in LDS -- the workgroups stay small enough to be placed beside another stream's GEMMs).
 * K == 256, G % 8 == 0, G >= 128. */
int spadot_recon_fb_supported(int b, int K, int G);
long long spadot_recon_fb_workspace(int b, int K, int G);
int spadot_recon_fb(const void *h_bf16, const void *W_bf16, const void *WT_bf16, int ldt, const float *bias, const float *y, int b, int K,
                    int G, double inv_scale, const float *grad_weight, void *g_bf16, float *workspace, double *loss_parts, float *dh,
                    void *stream);
int spadot_sum_parts(const double *part, int n, double scale, float *out, void *stream);

/* Measurement aid: buf[slot] = the device's constant-rate timestamp counter (100 MHz: 10 ns units) when the launch runs.
 * Launched at the head and the end of a captured stage it dates the stage on the GPU with no profiler attached. */
int spadot_stamp(unsigned long long *buf, int slot, void *stream);

/* One Lloyd iteration of K-means for R restarts at once (fp64, no atomics: two fits of the same data are bitwise
 * identical).  X [n, D] (centred data), C [R, K, D] centres (updated in place unless done[r]), part: work space of
 * R * ceil(n/256) * (K*(D+1) + 1) doubles, done [R] int flags (set when the squared centre shift <= tol), inertia [R]
 * = inertia of the centres the iteration started from, labels [R, n] int32 or NULL.  update = 0: assignment, partial
 * sums and labels only.  K <= 32, D <= 32.  (KMeans of _train_utils.py:255-269.) */
int spadot_lloyd_step(const double *X, double *C, int n, int D, int K, int R, double tol, double *part, int *done,
                      double *inertia, int *labels, int update, void *stream);
/* The same iteration for SEVERAL data sets in one launch pair (the per-epoch K-means of all time points,
 * _train_utils.py:255-269): group g owns rows xoff[g] .. xoff[g] + npts[g] - 1 of X [sum npts, D] and the restarts
 * g * rpg .. (g + 1) * rpg - 1 of C [groups * rpg, K, D]; tol[g] is its stopping threshold (device arrays); n_max = the largest
 * npts (grid size); part needs groups * rpg * ceil(n_max / 256) * (K (D + 1) + 1) doubles.  skip_done != 0: restarts whose
 * done flag is set are left alone entirely (their centres are final; inertia[r] then still belongs to the centres of the
 * iteration that froze them -- measure the final one with a call that has skip_done = 0 and every done flag set). */
int spadot_lloyd_step_groups(const double *X, double *C, const int *xoff, const int *npts, int n_max, int groups, int rpg, int D,
                             int K, const double *tol, double *part, int *done, double *inertia, int update, int skip_done,
                             void *stream);

/* Exact kk nearest neighbours of every point among all n points (self included), brute force in fp64, ordered by
 * (squared distance, index): out [n, kk] int32.  x [n, d] fp64, d <= 4, kk <= min(n, 128).  Replaces the host
 * NearestNeighbors call of _Cal_Spatial_Net (_utils.py:66-75) when the coordinates already live in HBM. */
int spadot_knn(const double *x, int n, int d, int kk, int *out, void *stream);

/* ---------------------------------------------------------------- GAT edge phase on the matrix cores (bf16 rows)
 * Same arithmetic as spadot_gat_forward / _backward_target / _backward_source (GATConv message passing of
 * /root/reference/SpaDOT/model/encoder.py:41-58) for bf16 rows with C = 512 channels per head, H in
 * {1, 2, 4, 8} and head concat, organised by BLOCK PLANS (spadot_amd/graph.py: BlockPlan): blocks of 32 rows, each
 * with the list of its distinct columns (plan_cols, padded to a multiple of 32, offsets plan_sptr).  Every distinct
 * source row of a block is fetched once (LDS-DMA) and the weighted sum is a dense product on v_mfma_f32_32x32x16_bf16.
 * The weights travel as a dense image `acell` [chunk of 16 columns][head][hi | lo][512] of bf16 (weights = hi + lo, 16
 * significant bits; zero where the tile has no edge; the 512 positions of a 32 x 16 tile are in MFMA fragment order),
 * which the per-node kernels fill through cellq[e] = chunk * 512 + position of edge e.  The image must be ZERO before
 * its first use and may then be reused: edges always overwrite the same cells.
 *   spadot_gat_alpha             alpha[e, hd] = softmax over the incoming edges of each of the n_tgt targets, written as
 *                                fp32 [E, H] and into the by-target plan's image (cellq = plan's cellq) and, optionally, into
 *                                the by-SOURCE plan's image (cellq_s indexed by by-target edge position): round 5, the
 *                                backward product's weights are gradient-independent and are written by the forward pass
 *   spadot_gat_aggregate         mode 0: out[row] = act?(sum_cols w x[col] + vec_a (bias));              plan by target
 *                                mode 1: out[row] = sum_cols w x[col] + ds_src[row] vec_a + ds_dst[row] vec_b  (att_src,
 *                                att_dst: the logits' own gradient path);                                plan by source
 *                                with att_part (mode 1; h_rows = the layer's input h): block b also leaves the partial sums
 *                                sum_rows ds_src[row] h[row] at att_part[b * part_width + 0 ..] and the ds_dst ones at
 *                                [.. + H C ..] -- the attention-vector gradients up to a column sum over the blocks;
 *                                rows pad_from .. pad_to - 1 of `out` (allocated past the last node so that the next GEMM
 *                                sees a row count that is a multiple of 128) are written as zeros; with dz (mode 1) the
 *                                workgroup sums dz over the outgoing edges of its own rows itself (transposed CSR rowptr_t /
 *                                eid_t: what spadot_gat_ds_src computes in a launch of its own), ds_src may then be NULL and
 *                                ds_src_out (may be NULL) receives the sums
 *   spadot_gat_edge_dot          g_pre = g_out * (act ? LeakyReLU'(out) : 1) (written), dz[e] = <g_pre[row], h[col]> through
 *                                plan_cell [chunk][32 rows][16] -> edge position or -1; with bias_part block b leaves the
 *                                column sums of its 32 rows of g_pre at bias_part[b * part_width + part_col ..] (the bias
 *                                gradient up to a column sum over the blocks: spadot_colsum, fixed order)
 *   spadot_gat_softmax_backward  dz (raw d alpha) -> d logits in place, ds_dst[i] = sum over incoming edges; with cellq_s /
 *                                acell_s (both or neither) alpha is also written into the by-SOURCE plan's image
 *   spadot_gat_ds_src            ds_src[j] = sum of dz over the outgoing edges of j (transposed CSR)
 * spadot_gat_mfma_supported(dtype, H, C, max_cols) tells whether a plan whose longest column list is max_cols can run. */
int spadot_gat_mfma_supported(int dtype, int H, int C, int max_cols);

/* ---- The last GAT layer for the seeds only, aggregate-first (csrc/gat_tail.hip, round 4) -----------------------------
 * Replaces, for gat3 = GATConv(H*C -> C, heads = H, concat = False) (encoder.py:45,58; PyG GATConv semantics)
 * whose targets are the first n_tgt << n nodes (only the seeds' rows reach the loss, SpaDOT.py:82), the dense map over
 * ALL n source rows by dense maps over the n_tgt aggregated rows:
 *     out_i = 1/H sum_h ( sum_j alpha_ij^h x_j ) W_h^T + bias,   e_ij^h = leaky_relu_0.2( x_j . w_src^h + x_i . w_dst^h ),
 *     w_src^h = W_h^T att_src^h, w_dst^h = W_h^T att_dst^h          (W_h = rows h C .. h C + C - 1 of lin.weight [H C, K]).
 * dtype: 0 = fp32, 1 = bf16 rows (x, A, dA, dx, O); K <= 2048, K % 8 == 0, H in {1, 2, 4, 8}; everything else fp32.
 * s [n][2 H]: s[j][2 h] = x_j . w_src^h, s[j][2 h + 1] = x_j . w_dst^h.  CSR by target (rowptr [n_tgt + 1], col) and by
 * source (rowptr_t [n + 1], col_t = target, eid_t = position of the edge in the by-target order), self loops included.
 *   spadot_gat_tail_wvec             wv [2 H][K] from W, att (part: workspace of slices * 2 H * K floats); whi / wlo (both or neither
 *                                    NULL): bf16 images [2 H][K] with wv = whi + wlo to 16 bits, for the matrix-core logits
 *   spadot_gat_tail_logits           s = x wv^T for all n rows (bf16 rows with whi / wlo and K % 32 == 0: on the matrix cores)
 *   spadot_gat_tail_aggregate        alpha [E][H] (softmax over the incoming edges of each target: exp(e - max) / (sum + 1e-16))
 *                                    and A [H][n_tgt][K] = sum_j alpha x_j
 *   spadot_gat_tail_headmean         out [n_tgt][C] = 1/H sum_h O[h] + bias            (O [H][n_tgt][C] = A_h W_h^T, a library GEMM)
 *   spadot_gat_tail_colsum_rows      column sums of g [rows][C] in fp32 (the bias gradient)
 *   spadot_gat_tail_edge_backward    dz [E][H] = d logit per edge (through aggregation, softmax and leaky_relu), ds_dst [n_tgt][H]
 *   spadot_gat_tail_source_backward  dx [rows_out][lddx] (rows n .. rows_out - 1 zero), ds_src [n][H]; with act_out (the layer's
 *                                    input rows [rows_out][ldm], may be NULL) dx is multiplied by `slope` where act_out <= 0
 *   spadot_gat_tail_dwvec            per-block partials [spadot_gat_tail_dwvec_rows(n)][2 H][K] of d wv (spadot_colsum adds them)
 *   spadot_gat_tail_wvec_backward    dW (+)= att (x) d wv per row of W, datt = W d wv
 * All sums in fixed orders, no atomics: repeated calls are bit-identical. */
int spadot_gat_tail_supported(int dtype, int H, int K);
int spadot_gat_tail_wvec(const float *W, int ldw, const float *att_src, const float *att_dst, int H, int C, int K, float *part,
                         int slices, float *wv, void *whi, void *wlo, void *stream);
int spadot_gat_tail_logits(const void *x, int dtype, int ldx, const float *wv, const void *whi, const void *wlo, int n, int H, int K,
                           float *s, void *stream);
int spadot_gat_tail_aggregate(const void *x, int dtype, int ldx, const float *s, const int *rowptr, const int *col, int n_tgt, int H,
                              int K, void *A, float *alpha, void *stream);
int spadot_gat_tail_headmean(const void *O, int dtype, const float *bias, int n_tgt, int H, int C, void *out, void *stream);
int spadot_gat_tail_colsum_rows(const void *g, int dtype, int rows, int C, float *out, void *stream);
/* ... and gs = scale * g (same dtype) in the same pass: the backward's d O_h = g / H and the bias gradient in one launch */
int spadot_gat_tail_scale_colsum(const void *g, int dtype, int rows, int C, double scale, void *gs, float *out, void *stream);
int spadot_gat_tail_edge_backward(const void *x, int dtype, int ldx, const void *dA, const float *s, const float *alpha, const int *rowptr,
                                  const int *col, int n_tgt, int H, int K, float *dz, float *ds_dst, void *stream);
int spadot_gat_tail_source_backward(const void *dA, int dtype, const float *alpha, const float *dz, const float *ds_dst, const float *wv,
                                    const int *rowptr_t, const int *col_t, const int *eid_t, int n, int n_tgt, int rows_out, int H, int K,
                                    void *dx, int lddx, float *ds_src, const void *act_out, int ldm, double slope, void *stream);
int spadot_gat_tail_dwvec_rows(int n);
int spadot_gat_tail_dwvec(const void *x, int dtype, int ldx, const float *ds_src, const float *ds_dst, int n, int n_tgt, int H, int K,
                          float *part, void *stream);
int spadot_gat_tail_wvec_backward(const float *W, int ldw, const float *att_src, const float *att_dst, const float *dwv, int H, int C, int K,
                                  float *dW, int lddw, int accumulate, float *datt_src, float *datt_dst, void *stream);
/* (cellq_s / acell_s, both or neither: the by-SOURCE plan's cell map and image -- the same weights once more, for the backward
 * product; a layer that will not be differentiated passes NULL) */
int spadot_gat_alpha(const float *s_src, const float *s_dst, const int *rowptr, const int *col, const int *cellq, int n_tgt,
                     int H, float *alpha, void *acell, const int *cellq_s, void *acell_s, void *stream);
int spadot_gat_aggregate(const void *x, int dtype, const void *acell, const int *plan_rows, const int *plan_sptr,
                         const int *plan_cols, int nb, int max_cols, int H, int C, int mode, const float *vec_a,
                         const float *vec_b, int act, const float *ds_src, const float *ds_dst, void *out,
                         const void *h_rows, float *att_part, int part_width, int pad_from, int pad_to, const float *dz,
                         const int *rowptr_t, const int *eid_t, float *ds_src_out, void *stream);
int spadot_gat_edge_dot(const void *g_out, const void *out, const void *h, int dtype, const int *plan_rows,
                        const int *plan_sptr, const int *plan_cols, const int *plan_cell, int nb, int max_cols, int H,
                        int C, int act, void *g_pre, float *dz, float *bias_part, int part_width, int part_col, void *stream);
/* (ds_dst has n_all >= n_tgt rows: the rows of nodes that are sources only are written as zeros) */
int spadot_gat_softmax_backward(const float *alpha, const float *s_src, const float *s_dst, const int *rowptr,
                                const int *col, const int *cellq_s, int n_tgt, int n_all, int H, float *dz, float *ds_dst,
                                void *acell_s, void *stream);
int spadot_gat_ds_src(const float *dz, const int *rowptr_t, const int *eid_t, int n, int H, float *ds_src, void *stream);

/* ---- dense map of a GAT layer on the matrix cores (csrc/gemm_bf16.hip) -------------------------------------------------
 * C [M x N] (bf16, row stride ldc) = A [M x K] (bf16, lda) . B^T with B stored [N x K] (bf16, ldb); fp32 accumulation.
 * Replaces the library GEMM under GATConv's `lin` (/root/reference/SpaDOT/model/encoder.py:41-58 -> torch_geometric
 * GATConv.lin) at the training shapes.  Requires N % 256 == 0, K % 32 == 0, 16-byte aligned pointers and strides that are
 * multiples of 8 elements; returns -22 otherwise (the caller then uses the library).  One 320 x 256 tile per workgroup. */
int spadot_gemm_tn_bf16(const void *A, int lda, const void *B, int ldb, void *C, int ldc, int M, int N, int K, void *stream);
/* The same kernel with B stored [K x N] (contraction index = row; transposed LDS reads for that operand): C = A . B, the
 * input gradient of a dense map (dx = g W with W the [N_out x K_in] weight image).  Same shape conditions. */
int spadot_gemm_nn_bf16(const void *A, int lda, const void *B, int ldb, void *C, int ldc, int M, int N, int K, void *stream);
/* ... with C[m][n] multiplied by `slope` wherever act_out[m][n] <= 0 (act_out [M x N] bf16, row stride ldm): the input gradient
 * of a dense map, dx = g W, times the LeakyReLU' of the activation that produced the map's INPUT (encoder.py:56-57) -- handed
 * to the layer below already masked, so that its edge backward neither reads its own output nor writes a masked copy. */
int spadot_gemm_nn_bf16_masked(const void *A, int lda, const void *B, int ldb, void *C, int ldc, int M, int N, int K,
                               const void *act_out, int ldm, double slope, void *stream);
/* spadot_gemm_tn_bf16 with the LAST `tail_row_tiles` row panels (320 rows each) cut into `slices` (2..8) slices of the
 * contraction; fp32 partial tiles in the caller's `workspace` (spadot_gemm_bf16_split_workspace floats, 16-byte aligned),
 * added in slice order and rounded once to bf16 by a second launch: bit-reproducible.  For a map whose grid is exactly one
 * round of workgroups and that shares the chip with another stream's long kernel (the second GAT layer beside the SVGP
 * branch's inverse: 40 of 256 compute units busy): 216 whole + 160 quarter tiles take 1.25 tile times instead of 2. */
long long spadot_gemm_bf16_split_workspace(int M, int N, int tail_row_tiles, int slices);
int spadot_gemm_tn_bf16_split(const void *A, int lda, const void *B, int ldb, void *C, int ldc, int M, int N, int K,
                              int tail_row_tiles, int slices, float *workspace, void *stream);


/* ---- weight gradient of a GAT layer's dense map on the matrix cores (csrc/gemm_wgrad_bf16.hip) ---------------------------
 * dW [N x K] (fp32, row stride ldw) = G^T X with G [M x N] (bf16, ldg) and X [M x >= K] (bf16, ldx; columns K .. the next
 * multiple of 256 must be readable -- the padded gene axis of the batch cache is).  256 x 256 output tiles x `slices`
 * slices of the M rows (0: one), partial tiles added in slice order by a second launch: bit-reproducible.
 * Requires N % 256 == 0, 16-byte aligned operands, strides % 8 == 0, operand images below 4 GiB; -22 otherwise.
 * The library keeps NO state: `workspace` (fp32, spadot_gemm_wgrad_bf16_workspace(M, N, K, slices) floats, 16-byte aligned;
 * may be NULL when that is 0) and `zero_row` (512 bytes of zeros, 16-byte aligned: the source of rows past M) belong to the
 * caller, so that calls on different streams, or captured into different graphs, never share scratch. */
long long spadot_gemm_wgrad_bf16_workspace(int M, int N, int K, int slices);
int spadot_gemm_wgrad_bf16(const void *G, int ldg, const void *X, int ldx, float *dW, int ldw, int M, int N, int K,
                           int slices, float *workspace, const void *zero_row, void *stream);
/* The same with the tile width along K chosen by the caller: tile_k = 256 (the entry points above) or 192 (256 x 192 tiles:
 * for shapes whose square-tile grid leaves the chip partly empty, e.g. 2048 x 3072 -- 96 square tiles x 2 slices = 192
 * workgroups, but 128 x 2 = 256 of these).  X columns up to the next multiple of tile_k must be readable. */
long long spadot_gemm_wgrad_bf16_workspace_tiled(int M, int N, int K, int slices, int tile_k);
int spadot_gemm_wgrad_bf16_tiled(const void *G, int ldg, const void *X, int ldx, float *dW, int ldw, int M, int N, int K,
                                 int slices, int tile_k, float *workspace, const void *zero_row, void *stream);

/* ---- hidden stages of the decoder as one launch each way (csrc/mlp_chain.hip) -------------------------------------------
 * /root/reference/SpaDOT/model/decoder.py:3-20: [Linear, LayerNorm, LeakyReLU] x n_layers on x [b, dims[0]] (fp32);
 * stage l maps dims[l] -> dims[l + 1].  forward keeps, per stage, a (linear output), y (stage output), mean, invstd.
 * backward: dy = gradient at the last stage's output; writes dx [b, dims[0]] (may be NULL) and `grads`, laid out per stage
 * as [dW (dout x din) | dbias (dout) | dgamma (dout) | dbeta (dout)] (spadot_mlp_chain_workspace gives that width and the
 * number of workspace rows; workspace = rows x width floats of scratch).  Fixed summation order.
 * supported: <= 4 stages, widths <= 256, input widths % 4 == 0, output widths % 8 == 0; all pointers 16-byte aligned. */
int spadot_mlp_chain_supported(int n_layers, const int *dims);
int spadot_mlp_chain_workspace(int b, int n_layers, const int *dims, int *n_rows, int *width);
int spadot_mlp_chain_forward(const float *x, int b, int n_layers, const int *dims, const float *const *W,
                             const float *const *bias, const float *const *gamma, const float *const *beta, const double *eps,
                             const double *slope, float *const *a, float *const *y, float *const *mean, float *const *invstd,
                             void *stream);
/* ... the same launch, also leaving a bf16 copy of the LAST stage's output [b, dims[n_layers]] in y_last_bf16 (may be NULL):
 * the operand of the output map's matrix-core GEMM (decoder.py:20), so that no cast launch sits between them. */
int spadot_mlp_chain_forward_bf16(const float *x, int b, int n_layers, const int *dims, const float *const *W,
                                  const float *const *bias, const float *const *gamma, const float *const *beta,
                                  const double *eps, const double *slope, float *const *a, float *const *y, float *const *mean,
                                  float *const *invstd, void *y_last_bf16, void *stream);
/* grads may be NULL: the per-workgroup partials stay in `workspace` and the caller adds them later (spadot_colsum).
 * spadot_mlp_chain_backward_add: dx = (the chain's input gradient) + dx_add [b, dims[0]] (may be NULL) -- a gradient that
 * reaches the chain's input by another path, added here instead of by a launch of its own. */
int spadot_mlp_chain_backward(const float *dy, const float *x, int b, int n_layers, const int *dims, const float *const *W,
                              const float *const *gamma, const double *slope, float *const *a, float *const *y,
                              float *const *mean, float *const *invstd, float *dx, float *workspace, float *grads, void *stream);
int spadot_mlp_chain_backward_add(const float *dy, const float *x, int b, int n_layers, const int *dims, const float *const *W,
                                  const float *const *gamma, const double *slope, float *const *a, float *const *y,
                                  float *const *mean, float *const *invstd, float *dx, const float *dx_add, float *workspace,
                                  float *grads, void *stream);

/* ---- reconstruction term on the output map's GEMM result (csrc/mlp_chain.hip) -------------------------------------------
 * out[0] = inv_scale * sum_{r,c} (y[r,c] - (o[r,c] + bias[c]))^2   (/root/reference/SpaDOT/model/SpaDOT.py:89 on the
 * decoder's output map, decoder.py:20), o = h W^T [b x G] fp32 without the bias: two launches (b <= 4096; scratch: >= 4096 doubles).  backward: g (bf16 [b x G]) = g1[0] * d out / d o and
 * dbias[c] = sum_r of the same values in fp32, fixed order. */
int spadot_bias_sqerr_forward(const float *o, const float *bias, const float *y, int b, int G, double inv_scale, double *scratch,
                              float *out, void *stream);
int spadot_bias_sqerr_backward(const float *g1, const float *o, const float *bias, const float *y, int b, int G,
                               double inv_scale, void *g_bf16, float *dbias, void *stream);

/* ---- GAT_fc on the bf16 rows of the last GAT layer (csrc/mlp_chain.hip) -------------------------------------------------
 * out [b x N] (fp32) = h [b x K] (bf16) . W^T [N x K] (fp32) + bias, N <= 32, K % 8 == 0: the (mu | logvar) head of
 * /root/reference/SpaDOT/model/encoder.py:59-61.  The forward is a cast + library GEMM; backward (fixed order): dh (bf16) and
 * per-8-row partials of dW, db in one launch, summed by a second. */
int spadot_headfc_backward(const float *g, const void *h_bf16, const float *W, int b, int K, int N, void *dh_bf16,
                           float *workspace /* ceil(b / 8) x (N K + N) floats */, float *grads /* [dW (N x K) | db (N)] */,
                           void *stream);

/* dst[t][r, 0:K[t]] = (bf16) src[t][r, 0:K[t]] for n <= 4 row-major matrices in one launch (fp32 weights -> their
 * compute-dtype images; dst rows have Kp[t] >= K[t] elements, the padding is not touched; K, Kp multiples of 4). */
int spadot_cast_rows_multi(const float *const *src, void *const *dst, const int *rows, const int *K, const int *Kp, int n,
                           void *stream);

/* ---------------------------------------------------------------- optimiser (one flat fp32 buffer)
 * sumsq[0] = sum g^2 over `count` gradients (deterministic two-stage reduction; scratch >= 2048 doubles). */
int spadot_grad_sumsq(const float *grad, long long count, double *scratch, float *sumsq, void *stream);
/* clip_grad_norm_(max_norm) folded into AdamW (torch semantics: decoupled weight decay, bias
 * correction, eps outside the sqrt):  coef = min(1, max_norm / (sqrt(sumsq) + 1e-6)); g = coef * grad;
 * p -= lr*wd*p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v/(1-b2^t)) + eps).
 * sumsq is read on the device (no host round trip). */
/* clip_grad_norm_ + AdamW with the step count on the device, TWO launches: the gradient's sum of squares (per-workgroup
 * partials in `scratch` (>= 2048 doubles), the last workgroup adds them in order, writes sumsq[0] and advances
 * step_dev[0]; `counter` is one unsigned that is 0 before the first call and left 0), then the update. */
int spadot_clip_adamw_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long count, double lr,
                          double beta1, double beta2, double eps, double weight_decay, double max_norm, double *scratch,
                          float *sumsq, int *step_dev, unsigned *counter, const float *grad_scale_dev, void *stream);
/* grad_scale_dev (device scalar, may be NULL = 1): `grad` stands for grad_scale * grad -- the data-parallel step
 * all-reduces a SUM over replicas and passes 1 / (number of replicas that had a batch in this step), so that the clip
 * threshold and the update see the MEAN gradient, as a single replica would (spadot_amd/parallel.py). */
/* The same two-stage clip + AdamW, with the update ALSO keeping bf16 images of registered weight matrices current: image w
 * is the row-major [rows x Kp] bf16 copy (columns K .. Kp-1 are padding the kernel never writes) of the fp32 weight that
 * occupies flat elements [offset, offset + rows * K).  The GAT layers' dense maps read such images; casting them took a
 * launch at the head of every step.  Requires count % 4 == 0, 16-byte aligned buffers, K, Kp, offset % 4 == 0,
 * rows * K < 2^31, at most 8 images; -22 otherwise. */
typedef struct spadot_weight_image {
    long long offset;
    int rows, K, Kp, reserved;
    void *image;
} spadot_weight_image;
typedef struct spadot_weight_images {
    int n, reserved;
    spadot_weight_image w[8];
} spadot_weight_images;
int spadot_clip_adamw_images_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long count, double lr,
                                 double beta1, double beta2, double eps, double weight_decay, double max_norm, double *scratch,
                                 float *sumsq, int *step_dev, const float *grad_scale_dev, const spadot_weight_images *images,
                                 void *stream);
/* The same update in two parts: the gradient norm + step count (once per step), then the element update of a RANGE
 * [offset, offset + count) of the flat buffers (offset, count multiples of 4; `images` may be NULL; image offsets refer to
 * the whole buffer).  A caller can update the parameters something is waiting for first and the rest afterwards: same
 * arithmetic per element, bit-identical to the one-call form (_train_utils.py:214-217). */
int spadot_grad_norm_step_dev(const float *grad, long long count, double *scratch, float *sumsq, int *step_dev, void *stream);
int spadot_adamw_range_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long offset, long long count,
                           double lr, double beta1, double beta2, double eps, double weight_decay, double max_norm,
                           const float *sumsq, const int *step_dev, const float *grad_scale_dev, const spadot_weight_images *images,
                           void *stream);

int spadot_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                      const float *sumsq, long long count, double lr, double beta1, double beta2,
                      double eps, double weight_decay, double max_norm, int step, void *stream);

/* Same update with the step count kept ON THE DEVICE (step_dev[0] is incremented, then used for the bias
 * corrections): nothing in the launch depends on a host-side counter, so a captured hipGraph of a whole
 * training step can be replayed. */
int spadot_adamw_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                          const float *sumsq, long long count, double lr, double beta1, double beta2,
                          double eps, double weight_decay, double max_norm, int *step_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPADOT_MODEL_H */
