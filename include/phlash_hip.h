/* libphlash_hip.so -- C ABI of the MI355X (gfx950) PSMC likelihood engine.
 *
 * Drop-in boundary for the *kernel plugin* of jthlab/phlash (v1.0.6).  Each entry point names the
 * reference interface it replaces (paths relative to the reference repo):
 *
 *   phk_create / phk_destroy   <->  _PSMCKernelBase.__init__ / __del__      src/phlash/gpu.py:101-174
 *                                   (validate + upload the int8 het matrix once, own it for life)
 *   phk_loglik                 <->  _PSMCKernelBase.__call__ + the two CUDA entry points
 *                                   `loglik` / `loglik_grad`                 src/phlash/gpu.py:182-325,
 *                                                                           529-538, 575-586
 *   phk_device_count           <->  PSMCKernel._initialize_devices           src/phlash/gpu.py:369-384
 *   phk_last_error             <->  CudaError / ASSERT_DRV                   src/phlash/gpu.py:23-46
 *
 * Conventions
 *   - plain C, no exceptions: every call returns PHK_OK (0) or a PHK_E* code; the message of the
 *     last failure on the calling thread is phk_last_error().
 *   - `params`, `inds`, `ll`, `grad` are DEVICE pointers (e.g. torch tensors' data_ptr()): nothing
 *     bounces through the host.  `data` of phk_create may be host or device memory.
 *   - calls on one handle are stream-ordered on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream) and return without synchronising.  A handle is not re-entrant (it owns
 *     scratch buffers), exactly like a reference kernel object (gpu.py:222-237).
 *   - parameter block layout [B, S or 1, 7, K], rows b,d,u,v,emis0,emis1,pi (gpu.py:189, 496-501),
 *     element type float (double_precision = 0) or double (1) (gpu.py:133-136).
 */
#ifndef PHLASH_HIP_H
#define PHLASH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHK_OK 0
#define PHK_EINVAL 1   /* bad argument (the reference raises AssertionError: gpu.py:106-113,197-214) */
#define PHK_ENOMEM 2   /* device allocation failed (the reference raises MemoryError: gpu.py:117-124) */
#define PHK_EHIP 3     /* HIP runtime error (the reference raises CudaError/RuntimeError) */
#define PHK_EUNSUPPORTED 4 /* K / variant not compiled in */
#define PHK_EOVERRUN 5 /* a kernel loop ran out of the iteration budget the host derived from the row length: the kernel
                          returned early, the evaluation is invalid; phk_last_error() names kernel, sequence and block.
                          (No counterpart in the reference, whose kernels loop over ``L`` alone: gpu.py:541, 617.) */

typedef struct phk_handle phk_handle;

/* ABI version of this library (major*1000 + minor). */
int phk_version(void);

/* Message of the last failure on this thread ("" if none).  Never NULL. */
const char* phk_last_error(void);

/* Number of visible HIP devices. */
int phk_device_count(int* n);

/* Create a kernel object for K hidden states over the observation matrix data[N][L]
 * (int8, values -1 missing / 0 hom / >=1 het; values above 1 are clipped to 1 and values below -1
 * rejected, as gpu.py:106-110; rows that are entirely missing are rejected as gpu.py:111-113).
 * The matrix is re-packed on the device to a private 2-bit layout; `data` is not referenced after
 * the call returns.  K in {4,8,16,32,64}. */
int phk_create(phk_handle** out, int K, const int8_t* data, int64_t N, int64_t L,
               int data_on_device, int double_precision, int device);

int phk_destroy(phk_handle* h);

/* One evaluation over B particles x S chunks.
 *   params     device, [B, S, 7, K] with element strides (pstride_b, pstride_s); pstride_s = 0
 *              broadcasts one [7,K] block of a particle over all S chunks.
 *   inds       device int64 [S]: row of `data` for each chunk.  0 <= inds[s] < N (gpu.py:197-199) is
 *              checked on the device: an index outside the range is clamped to row 0 and raises
 *              bit 1 of the flag word; the next phk_underflow_risk then returns PHK_EINVAL.
 *   W          number of leading sites of every row that are run but NOT scored (the reference's
 *              warm-up prefix, model.py:52-55).  W = 0 reproduces `loglik`/`loglik_grad` exactly:
 *              pi is the state law one transition before site 0.
 *   ll         device double [B, S]: log-likelihood of sites W..L-1 of chunk s under particle b
 *   grad       device [B, S, 7, K] (float or double as the handle) or NULL for the no-gradient
 *              kernel.  grad_dlog = 0: d ll / d theta.  grad_dlog = 1: theta * d ll / d theta, the
 *              quantity the reference kernel returns (gpu.py:647-653,686-691), every row in natural
 *              index order (i.e. after the reference's np.roll of the v row, gpu.py:307-309).
 */
int phk_loglik(phk_handle* h, const void* params, int64_t pstride_b, int64_t pstride_s,
               const int64_t* inds, int64_t B, int64_t S, int64_t W, double* ll, void* grad,
               int grad_dlog, void* stream);

/* float64 parameter blocks -> what a float32 kernel object runs on, one launch.  The reference rounds its float64
 * PSMCParams to the kernel's float type field by field on the host (src/phlash/gpu.py:200-214, ``astype(float_type)``);
 * the float32 kernels here additionally run on the model written with hom emission 1 ("folded"):
 *   params       device double [nblocks, 7, K]  rows b,d,u,v,emis0,emis1,pi (gpu.py:189)
 *   params_f32   device float  [nblocks, 7, K]  the same block rounded to float32
 *   prefold_f32  device float  [nblocks, 5, K]  rows fl(emis0 b), fl(emis0 d), fl(emis0 v), fl(emis1 / emis0), fl(1 / emis0),
 *                each formed in float64 and rounded ONCE.  Handed to phk_loglik_prefolded beside params_f32 the kernels
 *                take these factors as they are; without them (phk_loglik) they fold the float32 rows themselves, which
 *                rounds every folded factor three times -- an error that is the same at every site of a row and adds up
 *                along it.
 *   crel         device double [nblocks, 7, K] or NULL: coefficients of the first-order correction phk_ll_first_order applies */
int phk_prefold(int device, int K, const double* params, int64_t nblocks, float* params_f32, float* prefold_f32, double* crel,
                void* stream);

/* First-order correction of a float32 gradient call's log-likelihoods for the rounding of the model to float32:
 *     ll[b, s] += sum_j theta[b, s, j] * (d ll / d theta)[b, s, j] * crel[b, s, j]
 * with the gradient that call returned (grad_dlog as it was passed to phk_loglik*), the float64 blocks and the coefficients
 * of phk_prefold (params and crel share the element strides pstride_b, pstride_s; 0 broadcasts over the chunks).  The rounding of
 * a float32 model is the same at every site of a row, so its effect on ll grows with the row length (about 1e-8 per site
 * absolute: the reference's own float32 kernels lose the same); it is linear in the rounding residuals
 * to first order, and the derivative it multiplies is the gradient the call has just computed.  What is left is the
 * arithmetic's error, which does not add up coherently.  Not available to a no-gradient call. */
int phk_ll_first_order(int device, int K, double* ll, const float* grad, int grad_dlog, const double* params, const double* crel,
                       int64_t pstride_b, int64_t pstride_s, int64_t B, int64_t S, void* stream);

/* phk_loglik for a float32 kernel object with the blocks' pre-folded factors (phk_prefold): ``prefold`` is laid out like
 * ``params`` with five rows per block instead of seven (element strides pstride_b / 7 * 5, pstride_s / 7 * 5).  Same outputs,
 * same flags; gradients are with respect to the rows of ``params`` as ever. */
int phk_loglik_prefolded(phk_handle* h, const void* params, int64_t pstride_b, int64_t pstride_s, const float* prefold,
                         const int64_t* inds, int64_t B, int64_t S, int64_t W, double* ll, void* grad, int grad_dlog,
                         void* stream);

/* Particle -> PSMCParams for a whole population in one launch, float64, with its Jacobian.
 * Replaces, for B particles at once: MCMCParams.to_dm (src/phlash/params.py:94-127),
 * SizeHistory.ect / .pi (src/phlash/size_history.py:123-138,170-193), transition_matrix + _expQ
 * (src/phlash/transition.py:9-85) and PSMCParams.from_dm (src/phlash/params.py:33-55), which the
 * reference evaluates per particle under vmap and differentiates with jax.grad.
 *   K                 hidden states (3..64); P epochs; epoch_of_state[K] (host) maps each state to its
 *                     epoch (the expansion of the PSMC pattern, src/phlash/util.py:35-37)
 *   x        device   [B, P+3] = t_tr(2), c_tr(P), rho_over_theta_tr  (ravel order of params.py:58-66)
 *   params   device   [B, 7, K] rows b,d,u,v,emis0,emis1,pi
 *   jac      device   [B, 7*K, P+3] d params / d x, or NULL */
int phk_param_map(int device, int K, int P, const int32_t* epoch_of_state, double theta, const double* x,
                  int64_t B, double* params, double* jac, void* stream);

/* The same, and the [B, 7, K] block once more rounded to float32 (params_f32, may be NULL): what a float32 kernel
 * object takes in phk_loglik, without a conversion launch in between. */
int phk_param_map_rounded(int device, int K, int P, const int32_t* epoch_of_state, double theta, const double* x,
                      int64_t B, double* params, double* jac, float* params_f32, void* stream);

/* The tail of a sampler step between the likelihood kernels and the SVGD update, two launches.
 * phk_reduce_chunks: the sum over the minibatch that model.py:57 takes (``.sum()``), for value and gradient, and the
 *   hand-over of the kernel object's flags, in the layout that is all-reduced over the ranks:
 *     ll   device double [B, S], grad device [B, S, 7, K] (float or double as the handle): outputs of phk_loglik
 *     buf  device double [B + 1, 1 + 7K]: row b = [sum_s ll, sum_s grad]; row B = [underflow flag, bad-index flag,
 *          0...] as 0.0 / 1.0 (the handle's flag word is cleared, as by phk_take_flags_async)
 * phk_chain_rule: log density of every particle and its gradient in particle space -- what the reference obtains
 *   with jax.grad through log_density (src/phlash/model.py:24-73) and PSMCParams.from_dm(MCMCParams.to_dm()):
 *     logp[b] = c_prior * log_prior(x_b) + c_hmm * buf[b, 0] + c_extra * extra_val[b]
 *     grad[b] = c_prior * d log_prior / d x_b + c_hmm * jac[b]^T buf[b, 1:] + c_extra * extra_grad[b]
 *   x [B, P+3], jac [B, 7K, P+3] (phk_param_map), buf as above after the all-reduce; extra_val [B] / extra_grad
 *   [B, P+3] may be NULL (the AFS term, evaluated elsewhere).  A particle whose logp is not finite gets -inf and a
 *   zero gradient (model.py:73 under jax.grad). */
int phk_reduce_chunks(phk_handle* h, const double* ll, const void* grad, int64_t B, int64_t S, double* buf, void* stream);
int phk_chain_rule(int device, int K, int P, double alpha, double beta, const double* x, const double* buf,
                   const double* jac, int64_t B, double c_prior, double c_hmm, const double* extra_val,
                   const double* extra_grad, double c_extra, double* logp, double* grad, void* stream);

/* The AFS term of the objective with its gradient w.r.t. the particles, one launch.  Replaces, under vmap + jax.grad,
 *   l3 = xlogy(T afs, T esfs).sum(),  esfs = etbl / etbl.sum(),  etbl = W etjj          (src/phlash/model.py:58-68,
 *   size_history.py:212-226: etjj_k = int_0^inf exp(-k(k-1)/2 R(t)) dt, k = 2..n, through JaxPPoly.exp_integral,
 *   jax_ppoly.py:44-84; W = _W_matrix(n), size_history.py:350-369),
 * a function of the particle through t = [0, geomspace(t1, tM, K-1)] and c = softplus(c_tr) by epoch only (params.py:94-127).
 *   x      device [B, P+3];  n: sample size (n - 1 spectrum entries), 3 <= n <= 128 (n = 2: the term is 0, do not call)
 *   tw     device [m, n-1] = T W,  w1 device [n-1] = column sums of W,  y device [m] = T afs   (constants of a run;
 *          T = identity, m = n - 1, when the run has no afs_transform)
 *   value  device [B];  grad device [B, P+3]: the arguments extra_val / extra_grad of phk_chain_rule */
int phk_afs_term(int device, int K, int P, const int32_t* epoch_of_state, const double* x, int64_t B, int n, int m,
                 const double* tw, const double* w1, const double* y, double* value, double* grad, void* stream);

/* log_prior of the whole population with its gradient, one launch.  Replaces log_prior
 * (src/phlash/model.py:11-21) under vmap + jax.grad:
 *   value[b] = logN(log(rho/theta); 0, 1) - alpha * sum_i (log c_{i+1} - log c_i)^2 - beta * |x_b|^2
 * with rho/theta = 0.1 + 9.9 sigmoid(x[P+2]) and c = softplus(x[2 .. 2+P)) per epoch (params.py:106-118).
 *   x      device [B, P+3];  value device [B];  grad device [B, P+3] d value / d x, or NULL */
int phk_log_prior(int device, int P, double alpha, double beta, const double* x, int64_t B, double* value,
                  double* grad, void* stream);

/* The SVGD / AMSGrad update of the sampler's inner step for the whole population on the device:
 *   phi_j = (1/B) sum_i [ -k_ij g_i + (2/h)(x_i - x_j) k_ij ],  k_ij = exp(-|x_i - x_j|^2 / h);
 *   AMSGrad (b1, b2, eps, bias correction with `count` = the step number starting at 1, running max of the
 *   corrected second moment), x_out = x - lr mu_hat / (sqrt(nu_max) + eps);
 *   h_out = median(pairwise distances of x_out)^2 / log B  (median as torch.quantile(., 0.5)).
 * Replaces what the reference delegates to blackjax.svgd(..., optax.amsgrad(lr)) per iteration
 * (src/phlash/mcmc.py:178-199, 279; blackjax 1.2.5 / optax 0.2.6).  All arrays device float64:
 *   x, grad_logp, mu, nu, nu_max, x_out [B, D] (mu, nu, nu_max updated in place; x_out must not alias x);
 *   h_in, h_out scalars (may alias); dist_ws workspace of phk_svgd_workspace_doubles(B) doubles, whose first
 *   B (B - 1) / 2 hold the pairwise distances of x_out on return (strict lower triangle).  The median is an exact
 *   bucket select: by one workgroup up to 256 particles, over the whole chip beyond (the reference's default is
 *   500).  h_out = NULL skips it.  B <= 4096, D <= 72. */
int64_t phk_svgd_workspace_doubles(int64_t B);
int phk_svgd_step(int device, int64_t B, int D, const double* x, const double* grad_logp, double* mu, double* nu,
                  double* nu_max, const double* h_in, double* h_out, double* x_out, double* dist_ws, int64_t count,
                  double lr, double b1, double b2, double eps, void* stream);

/* Tuning / introspection (no reference counterpart).
 * R = lanes per sequence (1,2,4,8,16; must divide K, K/R <= 16), T = checkpoint block (8, or 16
 * where K/R <= 4).
 * 0 = choose automatically from B*S. */
int phk_set_variant(phk_handle* h, int R, int T);
int phk_get_variant(phk_handle* h, int64_t B, int64_t S, int* R, int* T);
/* With autotuning on (default; environment PHK_AUTOTUNE=0 turns it off) the first phk_loglik for a
 * new batch shape times every compiled (R, T) on the first 2,048 sites of that batch (scratch
 * outputs, a few tens of ms, synchronises the stream once) and keeps the fastest; otherwise a
 * static rule picks R from the sequence count. */
int phk_set_autotune(phk_handle* h, int on);
/* How the gradient is evaluated: 0 = serial (forward kernel, then one backward sweep per sequence:
 * best when B*S sequences fill the chip), 1 = segmented (forward kernel and an independent
 * beta-recursion kernel run concurrently on two streams, then every 512-site segment of every
 * sequence is swept in parallel, each unit storing its partial sums in a slot of its own that a
 * finalize kernel adds up in unit order: best for small batches such as the reference's 500
 * particles x 5 chunks), -1 = automatic (default; the autotuner times both where the batch is small). */
int phk_set_backward_mode(phk_handle* h, int mode);
/* Force a complete plan (segmented = -1 returns to automatic).  Serial: (R, T).  Segmented: R for
 * the segment sweep, R_forward for the forward kernel, R_scan for the beta scan; T (8 or 16) is
 * shared by the forward kernel and the sweep. */
int phk_set_plan(phk_handle* h, int segmented, int R, int T, int R_forward, int R_scan);
/* The plan the last phk_loglik ran with. */
int phk_get_plan(phk_handle* h, int* segmented, int* R, int* T, int* R_forward, int* R_scan);
/* Hybrid form of the serial plan (chosen by the tuner where the serial sweep would leave wave slots
 * empty): sequences [0, first) of the launch are swept serially, the rest by segments (R_sweep lanes
 * per sequence, seeds from a beta scan with R_scan) at the same time.  first = 0: not hybrid. */
int phk_get_plan_hybrid(phk_handle* h, int64_t* first, int* R_sweep, int* R_scan);
/* Force the hybrid part of a plan forced with phk_set_plan(h, 0, R, T, R_forward, 0) (first = 0: plain serial).
 * Together the two calls reproduce what phk_get_plan + phk_get_plan_hybrid report, so that one rank's tuned
 * plan can be installed on its peers (the reference's GPUs all run the same kernel, gpu.py:386-438). */
int phk_set_plan_hybrid(phk_handle* h, int64_t first, int R_sweep, int R_scan);
/* The scaled forward state is brought back to [0.5,1) by an exact power of two after every nrm-th
 * site (1, 2 or 4; 0 = library default).  nrm = 1 is the reference's per-site normalisation
 * (hmm.py:77-79); larger intervals do the same arithmetic with fewer rescales and are safe while
 * nrm consecutive sites cannot shrink the total mass below the float range. */
int phk_set_rescale_interval(phk_handle* h, int nrm);
/* With nrm > 1 the forward kernel raises a sticky flag when a rescale finds the total mass below
 * 2^-64 (float) / 2^-600 (double), i.e. the parameters are extreme enough that the unscaled sites in
 * between could have lost precision.  Reads (and clears) the flag; synchronises with the work
 * enqueued before.  The Python host re-evaluates such a call with nrm = 1.
 * The same word carries two failure bits: a chunk index outside [0, N) (returns PHK_EINVAL) and, round 6, a kernel
 * loop that ran out of its iteration budget (returns PHK_EOVERRUN; the message names kernel, sequence and block). */
int phk_underflow_risk(phk_handle* h, int* flag);
/* Test hook: scale the iteration budgets the kernels of this handle are given (default 1 / 1; the budget is twice the
 * row's block count, an upper bound on the iterations any loop of a kernel can legitimately make).  kernels: bit 0 forward
 * kernel, bit 1 backward kernel, bit 2 beta scan.  A scale below ~1/2 makes a healthy evaluation overrun, which is how
 * tests/test_plans_and_modes.py exercises the PHK_EOVERRUN path. */
int phk_set_loop_budget_scale(phk_handle* h, int kernels, int num, int den);
/* Developer builds (-DPHK_ASM_RUN=1) only: the K = 16 float32 sweeps with two lanes per sequence run their hot blocks through a
 * hand-written instruction sequence (csrc/sweep_run_k16r2.inc, generated by scripts/gen_sweep_asm.py: the C++ body's arithmetic,
 * operation for operation, an all-hom block without its per-site tests) when on = 1.  It returns the same bits as the C++ body
 * (tests/test_hip_parity.py::test_asm_block_run_equals_the_cxx_body) and measured 1-2 % slower (profiles/r06_ab_experiments.txt
 * item 8), so the shipped library is built without it and answers on = 1 with PHK_EUNSUPPORTED. */
int phk_set_asm_run(phk_handle* h, int on);
/* The same flag word handed over WITHOUT a host synchronisation: a one-thread kernel on `stream`
 * writes dst[0] = 1.0 if the underflow-risk bit is set (else 0.0), dst[1] = 1.0 if a chunk index was
 * outside [0, N) (else 0.0) -- `dst` is a device array of two doubles -- and clears the word.  The
 * multi-rank host puts `dst` inside the buffer it all-reduces (SUM) anyway, so that every rank sees the same flags and takes the same redo branch
 * (the reference has no counterpart: its kernel objects are driven by threads of one process,
 * gpu.py:386-438). */
int phk_take_flags_async(phk_handle* h, double* dst, void* stream);
/* Deterministic mode (also environment PHK_DETERMINISTIC=1): the plan comes from the static rule,
 * never from a timing, and every reduction runs in a fixed order, so two calls with the same inputs
 * return the same bits.  (Without it the tuner may pick different variants on different runs, and
 * float32 variants differ in their last digits.) */
int phk_set_deterministic(phk_handle* h, int on);
/* Upper bound for the checkpoint workspace; larger problems are run in particle / chunk slabs. */
int phk_set_workspace_limit(phk_handle* h, int64_t bytes);
int64_t phk_workspace_bytes(phk_handle* h);
/* The (particles x chunks) slab the last phk_loglik was cut into (= B x S when it ran as one launch). */
int phk_get_slab(phk_handle* h, int64_t* particles, int64_t* chunks);
/* With profiling on, every phk_loglik records HIP events around its kernels on the call's stream;
 * phk_last_timing waits for them and returns the summed device time of the forward and backward
 * kernels of the last call (ms) and the number of launches of each. */
int phk_set_profiling(phk_handle* h, int on);
int phk_last_timing(phk_handle* h, float* fwd_ms, float* bwd_ms, int* n_launches);
/* Same, summed over every call since the previous phk_timing_totals (or since profiling was
 * switched on); waits once, at query time, so a timed loop needs no per-step synchronisation. */
int phk_timing_totals(phk_handle* h, double* fwd_ms, double* bwd_ms, int* n_launches);

/* Environment variables read by the library (none is needed in normal use):
 *   PHK_AUTOTUNE=0         static plan rule instead of the timing-based tuner
 *   PHK_DETERMINISTIC=1    as phk_set_deterministic(h, 1) for every handle
 *   PHK_TUNE_VERBOSE=1     the tuner's timings and decision on stderr
 *   PHK_HYBRID=R:Rf:first:R3:R2   developer override of the hybrid plan (tests)
 *   PHK_SIDE_PRIO=high     second stream at the highest instead of the lowest priority (A/B runs)
 *   PHK_POISON=<byte>      diagnostic: fill the scratch buffers with that byte before every launch sequence
 *                          (255: NaN patterns); PHK_POISON_MASK selects buffers */

#ifdef __cplusplus
}
#endif
#endif
