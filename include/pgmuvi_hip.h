/*
 * pgmuvi_hip.h -- C ABI of the MI355X (gfx950) exact-GP hot path.
 *
 * The reference (ICSM/pgmuvi) has no C / FFI / plugin interface: it reaches this
 * arithmetic through GPyTorch's Python object protocol.
 * Each entry point below names the reference call site whose arithmetic it
 * replaces; the Python shim `pgmuvi_amd.gpytorch` binds them with ctypes (see
 * INTEGRATION.md).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to row-major, contiguous fp64 data owned
 *    by the caller, unless the parameter name ends in `_host`;
 *  - all work is enqueued on the caller's `stream` (a hipStream_t passed as
 *    void*; NULL = the default stream); nothing synchronises the host except the
 *    functions documented to do so;
 *  - return value: 0 ok, <0 = -(index of the first bad argument) or -99 (a launch failed).  A failed
 *    factorisation is NOT a return value (nothing synchronises): it is reported through the device-side
 *    `info` output of the evaluation calls, LAPACK style (0 ok, k > 0 = 1-based index of the first
 *    non-positive pivot), with NaN in the value and in every gradient output of that problem;
 *  - a workspace belongs to the device it was created on: every call that takes one runs there whatever
 *    the caller's current device is, and puts the caller's current device back before it returns (the
 *    stream and the pointers must belong to the workspace's device);
 *  - no function throws or aborts; a workspace is not re-entrant (one caller
 *    thread per handle at a time).
 */
#ifndef PGMUVI_HIP_H
#define PGMUVI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgm_ws pgm_ws;

/* Library / build identification ("pgmuvi_hip <ver> gfx950"). */
const char* pgm_version(void);

/* Largest Q*d the kernels are built for (LDS staging of per-point factors). */
int pgm_max_qd(void);

/* Most points one light curve may have: 16384 = 128 block rows of 128 (see below). */
int64_t pgm_max_n(void);

/*
 * Workspace: every device buffer the path needs for `max_batch` simultaneous
 * problems of up to `max_n` points (factor matrix, diagonal-block inverses,
 * per-point factors, partial sums).  Allocated once, reused by every call.
 * `max_d` in {1,2}; `max_q * max_d <= pgm_max_qd()`.
 *
 * SIZE CONTRACT.  1 <= max_n <= pgm_max_n() = 16384; anything else returns -3
 * before a device is touched.  Up to 64 block rows (N <= 8192) one light curve
 * runs the fused sweep (its launch plans are 64-bit row masks; 41-64 block rows:
 * in windows); from 65 block rows on (N = 8193 .. 16384) the panel sweep takes
 * over -- the schedule of large batches, applied to the one matrix -- with the
 * same kernels, results and error reporting.  Both sides of that boundary and
 * the upper end are held to committed oracle fixtures at 1e-9 (value) / 1e-7
 * (gradients): N = 8192 (config 4), 8320 and 16384 (tests/golden,
 * tests/test_gpu_parity.py).  Larger matrices are refused rather than run
 * untested: nothing in the index arithmetic stops at 16384 (tile addresses are
 * 64-bit), but no parity value exists beyond it, and the reference itself does
 * not go there -- its exact GPs "scale to datasets of up to ~1000 points"
 * (/root/reference/paper/paper.md:144; Lightcurve subsamples to max_samples,
 * pgmuvi/lightcurve.py:1733).
 * Memory: about 9.2 * max_np^2 bytes per problem, max_np = max_n rounded up to
 * 128 (2.5 GB at 16384).
 */
int pgm_workspace_create(pgm_ws** ws, int device, int64_t max_n, int max_q, int max_d, int max_batch);
int pgm_workspace_destroy(pgm_ws* ws);
size_t pgm_workspace_bytes(const pgm_ws* ws);

/*
 * Dense spectral-mixture kernel matrix  K[i,j] = k(x1_i, x2_j)  (n1 x n2, leading
 * dimension ldk), optionally + diag(noise) (+ noise_scalar) when x1 == x2.
 * Replaces gpytorch SpectralMixtureKernel.forward as constructed at
 * pgmuvi/gps.py:208 (1-D) and :305 (ard_num_dims=2), i.e. `covar_module(x)`
 * at gps.py:219, and `kernel(x, x).to_dense()` (tests/test_kernels.py:41).
 *   w (Q), mu (Q x d), v (Q x d);  dim_order 0 = prod_d sum_q (GPyTorch),
 *   1 = sum_q prod_d.   noise may be NULL.
 */
int pgm_sm_kernel_f64(const double* x1, int64_t n1, const double* x2, int64_t n2, int d,
                      const double* w, const double* mu, const double* v, int q,
                      const double* noise, double noise_scalar, int dim_order,
                      double* K, int64_t ldk, void* stream);

/*
 * One marginal-log-likelihood evaluation, value (+ gradient when need_grad):
 * the arithmetic behind `output = model(train_x); loss = -mll(output, train_y);
 * loss.backward()` at pgmuvi/trainers.py:179-181, i.e. rows A1+A3+A4+A5 of
 * SURVEY.md section 8a (SM kernel build, + noise, blocked Cholesky, log-det,
 * inverse quadratic form, closed-form gradient) with GPyTorch's
 * fast_computations(False, False, False) semantics (pgmuvi/lightcurve.py:5966).
 *
 *   x (n x d), y (n), mean (n): mean-module output m(x);
 *   noise (n) fixed/heteroscedastic variances (FixedNoiseGaussianLikelihood,
 *   pgmuvi/lightcurve.py:2778-2789) or NULL, plus noise_scalar added to every
 *   diagonal entry (GaussianLikelihood's learned sigma^2, lightcurve.py:2807);
 *   jitter is added to the diagonal too (psd_safe_cholesky retry policy lives in
 *   the caller).
 * Outputs (device):
 *   mll[1]      log N(y | mean, K + noise) / n            (already divided by n)
 *   g_w[q], g_mu[q*d], g_v[q*d]   d mll / d (weights, means, scales)
 *   g_noise[n]  d mll / d noise_i  (sum it for the scalar noise)
 *   g_mean[n]   d mll / d mean_i   (= alpha / n)
 *   info[1]     0, or 1-based index of the first non-positive pivot (then the
 *               other outputs are undefined, as with LAPACK potrf).
 * Gradient outputs may be NULL when need_grad == 0.
 */
int pgm_mll_value_grad_f64(pgm_ws* ws, const double* x, const double* y, const double* mean,
                           const double* noise, double noise_scalar, int64_t n, int d,
                           const double* w, const double* mu, const double* v, int q,
                           int dim_order, double jitter, int need_grad,
                           double* mll, double* g_w, double* g_mu, double* g_v,
                           double* g_noise, double* g_mean, int* info, void* stream);

/*
 * The same for `batch` independent light curves of equal n (config 3 / MCMC
 * chains): every array gains a leading batch dimension (x: batch x n x d, w:
 * batch x q, ..., mll: batch, info: batch); `noise_scalar` may be NULL or a
 * device array of `batch` values.  All problems advance together, one launch
 * per algorithm step with the batch on gridDim.z.
 *
 * Light curves of at most 128 points (d = 1 or 2; pgmuvi's one published workload has 89, paper/paper.md:113, the multiband
 * light curve of its Lomb-Scargle notebook 106 in three bands): the whole
 * evaluation -- factors, matrix, factorisation, inverse, gradient contraction, results, status -- is ONE launch, one
 * workgroup per light curve (k_small); no launch graph, nothing but the results and what pgm_predict_f64 reads later is
 * written to memory.  Same results as the launch sequence of every other size (value: the same bits; gradients: 1e-12),
 * which the environment switch PGM_SMALL=0 (read when a workspace is made) brings back -- and which a call of up to 20 light
 * curves takes by itself where it is the faster (many mixtures on 100 and more points: one CU would do all of a light
 * curve's exponentials; the measured table is small_ok's, pgm_host.inc; PGM_SMALL=2: the one launch whatever the shape).
 */
int pgm_mll_value_grad_batched_f64(pgm_ws* ws, int batch,
                                   const double* x, const double* y, const double* mean,
                                   const double* noise, const double* noise_scalar, int64_t n, int d,
                                   const double* w, const double* mu, const double* v, int q,
                                   int dim_order, double jitter, int need_grad,
                                   double* mll, double* g_w, double* g_mu, double* g_v,
                                   double* g_noise, double* g_mean, int* info, void* stream);

/*
 * The same for `batch` light curves of DIFFERENT lengths -- what a real many-light-curve batch is: every
 * pgmuvi `Lightcurve` carries its own N and is subsampled to its own length (pgmuvi/lightcurve.py:1724-1733,
 * 2150-2181), and each is the unit of work of pgmuvi/trainers.py:179-181 ("for ragged N
 * sort by N^3").
 *   n_host (HOST array, batch entries): points of light curve b, 1 <= n_b <= stride_n, n_b <= the workspace's max_n;
 *   per-point arrays are padded to a common pitch: x is batch x stride_n x d, y / mean / noise / g_noise / g_mean are
 *   batch x stride_n (entries beyond n_b are neither read nor written); w, mu, v, mll, g_w, g_mu, g_v, info as in
 *   the batched call.
 * The light curves are taken in order of their 128-row block count, longest first, and advance in launch sets of at
 * most the workspace's max_batch members (pgm_ragged_plan shows the sets).  Twelve light curves and more (160 block rows
 * in all) form ONE set in which every member stops at its own last block row -- no tile beyond it is built, solved,
 * updated or multiplied; a few light curves run as sets of similar block-row counts whose shorter members end in
 * identity padding, a length joining the next longer set where that is cheaper than a launch set of its own.  Either
 * way no bit of a light curve's value depends on the company it is in: batched == single, bit for bit.  The length
 * table is uploaded when it differs from the previous call's (one host synchronisation, and the launch graphs of
 * the sets are recorded again; nothing of the kind in a fit loop over the same light curves).  Not inside a stream
 * capture (-24).
 */
int pgm_mll_value_grad_ragged_f64(pgm_ws* ws, int batch,
                                  const double* x, const double* y, const double* mean,
                                  const double* noise, const double* noise_scalar,
                                  const int64_t* n_host, int64_t stride_n, int d,
                                  const double* w, const double* mu, const double* v, int q,
                                  int dim_order, double jitter, int need_grad,
                                  double* mll, double* g_w, double* g_mu, double* g_v,
                                  double* g_noise, double* g_mean, int* info, void* stream);
/* Host only (no GPU): the launch sets of such a call.  set_of_host[b] (batch entries, may be NULL) = the set light curve b
 * runs in, nb_of_set_host[s] (up to batch entries, may be NULL) = the set's block rows; returns the number of sets. */
int pgm_ragged_plan(const int64_t* n_host, int batch, int max_batch, int* set_of_host, int* nb_of_set_host);
/* The same for an existing workspace: ITS slot count and the schedule switches frozen when it was made -- exactly the sets
 * pgm_mll_value_grad_ragged_f64 runs on it (pgm_ragged_plan answers for a workspace that would be made now). */
int pgm_ragged_plan_ws(pgm_ws* ws, const int64_t* n_host, int batch, int* set_of_host, int* nb_of_set_host);

/*
 * Posterior prediction at n_test inputs from the factor left in the workspace by
 * the last pgm_mll_value_grad_f64 call with need_grad != 0 (alpha and
 * L^-1 are kept):  mean_out = mean_test + K*^T alpha,
 *                  var_out  = k(x*,x*) - || L^-1 K* ||^2   (latent variance).
 * Replaces `likelihood(model(x_test))` in eval mode at
 * pgmuvi/lightcurve.py:9607-9631, 9862, 9937, 10071.
 */
int pgm_predict_f64(pgm_ws* ws, const double* x_test, const double* mean_test, int64_t n_test,
                    double* mean_out, double* var_out, void* stream);

/*
 * Per-kernel device timing (HIP events on `stream`), for bench.py's roofline
 * block.  While enabled every launch of the named phase is bracketed by events;
 * pgm_profile_read synchronises the stream and returns accumulated milliseconds
 * and launch counts per phase (arrays of pgm_profile_phases() entries; names from
 * pgm_profile_phase_name).  Disabled by default (no overhead).
 */
int pgm_profile_enable(pgm_ws* ws, int on);
int pgm_profile_phases(void);
const char* pgm_profile_phase_name(int phase);
int pgm_profile_read(pgm_ws* ws, double* ms_host, int64_t* launches_host);
/* 128^3 tile products of the inverse pass that the last single-light-curve evaluation with gradient ran inside the
 * factorisation sweep's diagonal-block launches (on CUs those launches would leave idle) instead of in the
 * inverse/gradient launch; 0 when that does not apply (batches, small or very large N).  For flop accounting. */
int64_t pgm_profile_early_inverse_products(const pgm_ws* ws);

/*
 * Device-resident fit: the optimiser loop of pgmuvi/trainers.py:177-195 for a
 * constant-mean spectral-mixture exact GP, one hipGraph replay per iteration and no host work in between.
 * Raw parameter vector (P = nmean + q + 2 q d (+1)): [mean | w | mu | v | (learned scalar noise)] where the mean block is the
 * constant (nmean = 1) or, with linear_mean, the d weights then the bias (nmean = d + 1); each entry with a
 * GPyTorch constraint: ckind 0 none, 1 softplus(raw) + ca (Positive / GreaterThan), 2 ca - softplus(-raw) (LessThan),
 * 3 ca + cb * sigmoid(raw) (Interval: ca = lower bound, cb = upper - lower).  optimizer 0 SGD, 1 Adam, 2 AdamW (torch
 * semantics).  x, y, noise are device pointers that must outlive the handle; raw0 / ckind / ca / cb are host arrays.
 * pgm_fit_run enqueues `iters` more iterations; pgm_fit_read synchronises and returns the iterations done, the loss
 * -mll per iteration, the raw parameters after each step ([iters][P]), the current raw parameters and the last
 * factorisation status (the device writes this log to host-mapped memory as it goes: a read is a stream synchronisation
 * and host copies).  For n <= 128 an iteration is ONE launch -- constraint transforms, the evaluation, the chain
 * rule and the optimiser step inside k_small -- replayed 25 iterations per graph.
 * pgm_fit_set_priors (optional, before the first pgm_fit_run): MAP instead of maximum likelihood -- per raw-vector entry
 * a prior on the CONSTRAINED value, kind 0 none, 1 Normal(loc, scale), 2 LogNormal(loc, scale) (the priors
 * pgmuvi/lightcurve.py:3273-3322 registers); their log densities are added to N * mll before the division by N, as
 * gpytorch's ExactMarginalLogLikelihood does, and the logged loss includes them.  Host arrays of P entries.
 */
typedef struct pgm_fit pgm_fit;
int pgm_fit_create(pgm_fit** out, pgm_ws* ws, const double* x, const double* y, const double* noise, int64_t n, int d, int q,
                   int dim_order, int linear_mean, const double* raw0, const int* ckind, const double* ca, const double* cb, int has_noise_param,
                   int optimizer, double lr, double beta1, double beta2, double eps, double weight_decay, int max_iter);
int pgm_fit_set_priors(pgm_fit* fit, const int* kind, const double* loc, const double* scale);
int pgm_fit_run(pgm_fit* fit, int iters, void* stream);
int pgm_fit_read(pgm_fit* fit, void* stream, int* iters_done, double* loss_hist, double* raw_hist, double* raw, int* info);
int pgm_fit_destroy(pgm_fit* fit);

/*
 * The sampler's potential on the device (BASELINE config 5; the model the reference's disabled
 * Lightcurve.mcmc describes, pgmuvi/lightcurve.py:5964-6003, with the priors of set_default_priors, :3235-3330).  For `batch`
 * chains, each on its own light curve (x batch x n x d, y batch x n, noise batch x n or NULL = a learned noise variance; device
 * pointers that must outlive the handle), with the per-chain unconstrained vector
 *     z = [c, log w (q), log mu (q d), log v (q d) (, log sigma^2)],   P = 1 + q + 2 q d (+ 1),
 * pgm_pot_eval returns   U(z) = -[ N mll(theta(z)) + sum_p log Normal(z_p; loc_p, scale_p) ]   and dU/dz  (a LogNormal(loc,
 * scale) prior on a positive site plus the Jacobian of theta = exp z IS the Normal(loc, scale) density of z; the mean constant
 * carries a Normal prior on its value).  loc_host / scale_host: batch x P host arrays (scale > 0).  z_host, u_host (batch),
 * g_host (batch x P), info_host (batch, may be NULL) are host arrays: positions go in and results come back through host-mapped
 * memory, one hipGraph replay per call (parameters from z, the fused evaluation, potential and gradient) -- no copy launches,
 * no stream synchronisation; the call returns when the results have arrived.  A failed factorisation or a non-finite value
 * gives U = +inf and a zero gradient (the sampler rejects the step).
 */
typedef struct pgm_pot pgm_pot;
int pgm_pot_create(pgm_pot** out, pgm_ws* ws, int batch, const double* x, const double* y, const double* noise, int64_t n, int d, int q,
                   int dim_order, const double* loc_host, const double* scale_host);
int pgm_pot_eval(pgm_pot* pot, const double* z_host, double* u_host, double* g_host, int* info_host, void* stream);
int pgm_pot_destroy(pgm_pot* pot);

/*
 * Dense back-end (the reference's non-spectral-mixture models, pgmuvi/gps.py:915-1342:
 * quasi-periodic, Matern, RBF, RQ, separable products, sums): the caller supplies a = K + noise as a dense symmetric
 * matrix ([batch][n][lda], device) and r = y - mean ([batch][n]); mll[batch] is the per-datum log marginal likelihood
 * (as above), g_a ([batch][n][ldg], symmetric) = dmll/da = (alpha alpha^T - a^-1) / 2n, g_r = dmll/dr = -alpha / n.
 * Same factorisation sweep as the spectral-mixture path; gradient tiles are accumulated with fp64 atomics
 * (reproducible to round-off).  Replaces `-mll(model(train_x), train_y)` + `.backward()` (pgmuvi/trainers.py:179-181)
 * for those models, with autograd pulling g_a back through the torch-built kernel matrix.
 */
int pgm_mll_dense_f64(pgm_ws* ws, int batch, const double* a, int64_t lda, const double* r, int64_t n, double jitter,
                      int need_grad, double* mll, double* g_a, int64_t ldg, double* g_r, int* info, void* stream);

/* Posterior prediction after pgm_mll_dense_f64(need_grad != 0): k_star = K(x_train, x_test) ([n][ldk]), k_ss[n_test] the
 * prior variances at the test inputs.  Same outputs as pgm_predict_f64. */
int pgm_predict_dense_f64(pgm_ws* ws, const double* k_star, int64_t ldk, const double* k_ss, const double* mean_test,
                          int64_t n_test, double* mean_out, double* var_out, void* stream);

/*
 * Lomb-Scargle periodogram for the seeding of the mixture frequencies: the
 * floating-mean ("generalised") periodogram, standard normalisation, exact fp64 sums, of `batch` light curves
 * (t, y, dy: [batch][n]; dy NULL = unit weights) on one frequency grid freq[nf]; power: [batch][nf];
 * scratch: [batch][2n+1] doubles of caller-owned device memory.  fit_mean = 1 is astropy's default.
 * Replaces `LombScargle(t, y, yerr).power(freq)` at pgmuvi/lightcurve.py:4320, 4504-4511 (fit_LS), which
 * Lightcurve.fit() calls at :5516-5541.
 */
int pgm_lomb_scargle_f64(const double* t, const double* y, const double* dy, int64_t n, int batch,
                         const double* freq, int64_t nf, int fit_mean, double* scratch, double* power,
                         void* stream);

/*
 * Factorisation status of the last pgm_mll_value_grad*_f64 / pgm_mll_kernel_value_grad_f64 call on `ws`, as soon as it is
 * final: the last diagonal block of the sweep writes the status to host-mapped memory and stamps it with the evaluation's
 * number; this call polls for that number (it does NOT wait for the inverse/gradient pass that follows on the stream, and
 * the evaluation needs no event in its middle: it replays as one graph) and returns 0 (every problem factored) or 1, with
 * the per-problem LAPACK-style codes in
 * info_host[batch] (host memory, may be NULL).  This is the host synchronisation GPyTorch's psd_safe_cholesky performs inside
 * `mll(output, y)` (pgmuvi/trainers.py:180) to decide on a jitter retry; the caller's Python work between forward and
 * backward then overlaps the rest of the evaluation.  <0: nothing to report (no evaluation yet, or it ran inside a stream
 * capture of the caller).
 */
int pgm_factorisation_status(pgm_ws* ws, int* info_host, int batch);
/*
 * The same for one particular evaluation: pgm_last_evaluation(ws) right after a call names the evaluation that call enqueued
 * (a counter per workspace); pgm_factorisation_status_of returns -1 when another evaluation has been enqueued on `ws`
 * since -- the host-visible status then belongs to that one, and the caller falls back to the `info` array its own call
 * filled on the device.  (A workspace is shared by every request it covers: a second model, a chunk of a batch.)
 */
int64_t pgm_last_evaluation(const pgm_ws* ws);
int pgm_factorisation_status_of(pgm_ws* ws, int64_t evaluation, int* info_host, int batch);

/*
 * The reference's other exact-GP models (pgmuvi/gps.py:915-1342: QuasiPeriodicGPModel, MaternGPModel,
 * PeriodicPlusStochasticGPModel, SeparableGPModel, ...) compose GPyTorch's stationary kernels with ScaleKernel /
 * ProductKernel / AdditiveKernel; `covar_module(x)` + `mll(output, y)` + `loss.backward()` (pgmuvi/trainers.py:179-181) of
 * such a model is this call.  The kernel is a sum of products of leaf kernels with scale factors,
 *     K = sum_t (prod of theta[tscale[t][..]]) * prod_{l in tmask[t]} k_l(x, x'; theta[par[l]..]),
 * leaves (parameters in theta order, constrained values): 1 RBF [l], 2-4 Matern 1/2, 3/2, 5/2 [l], 5 periodic [p, lambda],
 * 6 rational quadratic [l, alpha], 7 cosine [p], 8 linear [v], 9 constant [c]; `dims[l]` is the bit mask of the input
 * dimensions leaf l sees (its active_dims).  theta: [batch][nparam] device values; g_theta: d mll / d theta, same shape.
 * Everything else (noise, mean, jitter, mll, info, workspace) as in pgm_mll_value_grad_batched_f64.  The workspace must
 * have been created with max_q >= nparam.
 *
 * PARITY STATUS OF THE LEAF FORMULAS: UNPINNED.  They are GPyTorch's published forms as restated in
 * oracle/sm_mll_oracle.py (rbf, matern, periodic, rq, cosine) -- RBF exp(-r^2 / 2 l^2); Matern nu with r / l;
 * periodic exp(-2 sin^2(pi r / p) / lambda) with lambda = GPyTorch's `lengthscale` entering UNSQUARED; rational
 * quadratic (1 + r^2 / (2 alpha l^2))^-alpha; cosine cos(pi r / p).  The reference holds no recorded number that
 * exercises any of them (its tests check shapes and symmetry only, tests/test_kernels.py; no notebook fits these
 * models), gpytorch is not installed in the build container, so the HIP path equals the restatement to 1e-9 and the
 * restatement is checked against nothing: the periodic kernel's lengthscale convention in particular is a reading of
 * GPyTorch's documentation, not a verified fact.  The spectral-mixture path above is the pinned one.
 */
typedef struct pgm_kernel_program {
  int nleaf, nterm, nparam;
  unsigned char kind[6], dims[6], par[6];
  unsigned char tmask[4], tnscale[4], tscale[4][3];
} pgm_kernel_program;
int pgm_mll_kernel_value_grad_f64(pgm_ws* ws, int batch, const double* x, const double* y, const double* mean,
                                  const double* noise, const double* noise_scalar, int64_t n, int d,
                                  const pgm_kernel_program* prog /* host */, const double* theta, double jitter, int need_grad,
                                  double* mll, double* g_theta, double* g_noise, double* g_mean, int* info, void* stream);

/*
 * The same periodogram by the FFT approximation astropy's default `method='auto'` takes on a regular grid of more than
 * 200 frequencies -- i.e. what `LS.power(freq, assume_regular_frequency=True)` at pgmuvi/lightcurve.py:4514 (and the
 * per-band periodograms behind `LombScargleMultiband.power(method='fast')`, pgmuvi/multiband_ls_significance.py:202)
 * actually returns: the trigonometric sums on f_k = f0 + k df, k < nf, come from inverse FFTs of the samples spread onto
 * a regular grid of 2^ceil(log2(oversampling nf)) points with 4-point Lagrange weights (Press & Rybicki 1989; astropy's
 * defaults: oversampling 5).  It differs from the exact sums by up to 1e-2 in the power at the high-frequency end.
 * scratch: [batch][pgm_lomb_scargle_fast_scratch_doubles(n, nf, oversampling)] doubles of caller-owned device memory.
 */
int64_t pgm_lomb_scargle_fast_scratch_doubles(int64_t n, int64_t nf, int oversampling);
int pgm_lomb_scargle_fast_f64(const double* t, const double* y, const double* dy, int64_t n, int batch,
                              double f0, double df, int64_t nf, int fit_mean, int oversampling,
                              double* scratch, double* power, void* stream);

/* fp64 MFMA issue-rate probe (TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64 on
 * every CU); used once by bench.py to report the measured peak beside the
 * datasheet figure.  Synchronises. */
int pgm_probe_mfma_f64(int device, double* tflops_host);

#ifdef __cplusplus
}
#endif
#endif /* PGMUVI_HIP_H */
