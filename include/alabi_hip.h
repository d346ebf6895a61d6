/*
 * alabi_hip.h -- C ABI of libalabi_hip.so, the MI355X (gfx950) implementation of the
 * GP-surrogate + ensemble-MCMC hot path of jbirky/alabi.
 *
 * The reference has no FFI: it reaches this arithmetic through the Python object
 * protocols of two third-party packages (george.GP, emcee.EnsembleSampler).  Each entry
 * point below names the reference call site (path:line under /root/reference) whose work
 * it replaces; INTEGRATION.md shows the ctypes binding a maintainer adds on the alabi side.
 *
 * Conventions
 *   - All array arguments are DEVICE pointers to float64 / int32 / int64 unless marked
 *     "host".  They are owned by the caller (PyTorch-ROCm tensors' data_ptr()).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Work is
 *     enqueued on it; only the calls documented as synchronising wait for it.
 *   - Every call returns an int status; no exception crosses the boundary.
 *   - A handle is not thread-safe; distinct handles are independent.
 *   - Everything computes in IEEE fp64.
 */
#ifndef ALABI_HIP_H
#define ALABI_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define ALABI_OK 0
#define ALABI_NOT_POSITIVE_DEFINITE 1 /* Cholesky met a pivot <= 0 or NaN (see alabi_gp_last_pivot) */
#define ALABI_BAD_ARGUMENT 2
#define ALABI_HIP_ERROR 3
#define ALABI_NOT_COMPUTED 4          /* predict / set_y before a successful compute */
#define ALABI_TIMEOUT 5               /* persistent ensemble kernel: a bounded spin ran out (state is invalid; re-run
                                         with ALABI_ENS_STREAM=0) */

#define ALABI_UTILITY_BAPE 0
#define ALABI_UTILITY_AGP 1
#define ALABI_UTILITY_JONES 2

#define ALABI_MAX_DIM 64

typedef struct alabi_gp alabi_gp;
typedef struct alabi_ens alabi_ens;

/* Library / device --------------------------------------------------------------------- */
int alabi_abi_version(void);
const char* alabi_status_string(int status);
const char* alabi_last_error(void);            /* text of the last HIP error on this thread */
int alabi_device_info(int* n_cu, int* lds_bytes, char* arch /* >=64 bytes, host */);

/* GP object: replaces george.GP(kernel=var(y)*ExpSquaredKernel(metric, ndim), mean=,
 * white_noise=)  -- alabi/core.py:1000, :1141; alabi/gp_utils.py:230-233. ---------------- */
int alabi_gp_create(int n_cap, int d, alabi_gp** out);
int alabi_gp_destroy(alabi_gp* gp);

/* gp.set_parameter_vector(p) -- alabi/core.py:705-733, :1157.  log_M is a host array[d];
 * all three log_* values are natural logs (george convention). */
int alabi_gp_set_hyper(alabi_gp* gp, double mean, double log_white_noise, double log_amp,
                       const double* log_M);
/* Kernel family: 0 ExpSquaredKernel (default), 1 Matern32Kernel, 2 Matern52Kernel, 3 RationalQuadraticKernel
 * (log_alpha used by 3 only) -- the four choices of init_gp(kernel=), alabi/core.py:1000-1014. */
int alabi_gp_set_kernel(alabi_gp* gp, int kernel_type, double log_alpha);

/* gp.compute(X) -- alabi/core.py:1158, :1430, :1577; gp_utils.py:243.  Assembles
 * K = k(X,X) + exp(log_white_noise) I and factorises it (lower Cholesky).  X is [N,d]
 * row-major on the device.  SYNCHRONISES the stream (it returns the factorisation status).
 * Returns ALABI_NOT_POSITIVE_DEFINITE where LAPACK's potrf would report info > 0. */
int alabi_gp_compute(alabi_gp* gp, const double* X, int N, void* stream);
int alabi_gp_last_pivot(alabi_gp* gp, int* pivot /* host, 1-based like LAPACK info */);

/* Batched fit + held-out mean: the k-fold cross-validation search of init_gp / _opt_gp -- alabi/gp_utils.py:511-637
 * (_evaluate_candidate_worker: per fold deepcopy(gp), set_parameter_vector, compute(train), log_likelihood, predict(val)) mapped
 * over the candidates by a process pool at gp_utils.py:640-700; alabi/core.py:1287-1305.  ONE call evaluates `njobs`
 * (hyper-parameter vector, training rows, validation rows) triples that share the resident inputs X [n,d] and y [n] (device):
 * every kernel matrix is assembled, all are factorised in one launch of the task queue (the chains of different matrices run side
 * by side), alpha / log-likelihood / held-out mean follow in two launches, one read-back.
 *   hyper      host [njobs][4 + d]: mean, log white noise, log amplitude, log alpha (RationalQuadratic only), log M_1..d
 *   train_idx  device int32, the training rows of job b are train_idx[train_off[b] .. train_off[b+1]) (offsets: host int64)
 *   val_idx    the same for the validation rows (may be empty); mu_val [val_off[njobs]] receives the held-out means
 *   nll        host [njobs]: -log-likelihood of the training rows (+inf where K is not positive definite)
 *   status     host [njobs]: 0, or LAPACK's 1-based pivot index where the factorisation broke down (mu_val is then NaN)
 * Jobs are processed in chunks whose matrices fit `workspace_bytes` (0: ALABI_BATCH_BYTES or 16 GiB, at most half of the free memory).  Factors are bit-identical
 * to alabi_gp_compute on the same rows.  Synchronises with `stream`.  N <= 12288 per job. */
typedef struct alabi_gp_batch alabi_gp_batch;
int alabi_gp_batch_create(int d, int kernel_type, long long workspace_bytes, alabi_gp_batch** out);
int alabi_gp_batch_destroy(alabi_gp_batch* batch);
int alabi_gp_batch_fit_predict(alabi_gp_batch* batch, const double* X, const double* y, int n, int njobs,
                               const double* hyper /* host */, const int* train_idx, const long long* train_off /* host */,
                               const int* val_idx, const long long* val_off /* host */, double* mu_val,
                               double* nll /* host */, int* status /* host */, void* stream);
/* The row lists of a k-fold split for `ncand` candidates at once (alabi/gp_utils.py:538: sklearn KFold(shuffle=True).split yields
 * ascending train and validation rows): fold_of [ncand, n] int8 = the fold of every row under each candidate's shuffle (-1: unused
 * row); job c k + f gets fold f's rows in val_idx[val_off[job] ..) and the other used rows in train_idx[train_off[job] ..).  All
 * arrays on the device (offsets int64 [ncand k + 1], computed by the caller from the fold sizes). */
int alabi_cv_fold_lists(const signed char* fold_of, int ncand, int n, int k, const long long* train_off,
                        const long long* val_off, int* train_idx, int* val_idx, void* stream);
/* Factor [N,N] (zeros above the diagonal) / alpha [N] of a job of the LAST chunk processed (tests: bit-identity with the
 * single-matrix path); ALABI_BAD_ARGUMENT for a job of an earlier chunk. */
int alabi_gp_batch_get_factor(alabi_gp_batch* batch, int job, double* L_out, void* stream);
int alabi_gp_batch_get_alpha(alabi_gp_batch* batch, int job, double* alpha_out, void* stream);
/* Chunks whose queue launch ran into a bounded wait and were redone on the launch-per-step path (expected: 0). */
int alabi_gp_batch_timeouts(alabi_gp_batch* batch, int* count /* host */);

/* Extend the factorisation by ONE training point x_new[d] (device) with the hyper-parameters unchanged: the refit after every
 * active-learning iteration (alabi/core.py:1780 -> _fit_gp -> gp.compute at :1158) in O(N^2) through the cached L^-1 instead
 * of O(N^3).  Needs a free padding row (N not a multiple of 64 and N < n_cap): ALABI_BAD_ARGUMENT otherwise -- call
 * alabi_gp_compute then.  ALABI_NOT_POSITIVE_DEFINITE leaves the factor as it was.  Invalidates alpha (call alabi_gp_set_y).
 * SYNCHRONISES the stream. */
int alabi_gp_append(alabi_gp* gp, const double* x_new, void* stream);

/* Change the constant mean only (it does not enter K): keeps the factorisation, invalidates alpha. */
int alabi_gp_set_mean(alabi_gp* gp, double mean);

/* alpha = K^-1 (y - mean): george _compute_alpha inside gp.predict -- alabi/core.py:85. */
int alabi_gp_set_y(alabi_gp* gp, const double* y, void* stream);

/* gp.predict(y, Xs, return_var=) -- alabi/core.py:85, :95, :1441, :1486, :1601, :1812.
 * Xs is [M,d] row-major; mu[M]; var[M] or NULL (mean only).  White noise is not added
 * to var.  var may come out slightly negative from cancellation, as in the reference.
 * The first variance request after a factorisation builds and caches L^-1 (Npad^2 doubles,
 * about 0.25 ms at N = 2000); var = amp - |L^-1 k*|^2 is then a product on the matrix cores.
 * Without room for the cache the blocked forward substitution is used (same values to ~1e-14 amp). */
int alabi_gp_predict(alabi_gp* gp, const double* Xs, long long M, double* mu, double* var,
                     void* stream);

/* One held-out evaluation of a hyper-parameter vector: gp.compute(X) + gp.log_likelihood(y) + gp.predict(y, Xs) as the k-fold
 * cross-validation of the reference does per fold (alabi/gp_utils.py:568-600) -- the three calls in one, so that a search of
 * 875 folds does not pay the binding's per-call overhead three times.  nll_out is a host double (-log likelihood); mu[M] on
 * the device; M may be 0.  Same status codes as the three calls (NOT_POSITIVE_DEFINITE leaves the handle uncomputed). */
int alabi_gp_fit_predict(alabi_gp* gp, const double* X, int N, const double* y, const double* Xs, long long M,
                         double* mu, double* nll_out, void* stream);

/* Prediction with gradients with respect to the query point, for the acquisition optimiser:
 * grad_gp_mean_prediction / grad_gp_var_prediction -- alabi/utility.py:558-623, which difference the kernel numerically
 * (utility.py:511-555, step 1e-6) and form K^-1 explicitly (solver.get_inverse(), utility.py:610).  Here
 * dmu[M,d] = (dk / dx)^T alpha and dvar[M,d] = -2 (dk / dx)^T K^-1 k with the closed-form kernel derivative and
 * K^-1 k = W^T (W k), W = cached L^-1; queries are processed 16 at a time.  mu[M] / var[M] may be NULL.
 * All pointers are device pointers.  ALABI_HIP_ERROR when there is no room for the L^-1 cache. */
int alabi_gp_predict_grad(alabi_gp* gp, const double* Xs, long long M, double* mu, double* var,
                          double* dmu, double* dvar, void* stream);
/* The same for ONE point with host buffers on both sides -- the evaluation the polish step of find_next_point makes ~30 times per
 * active-learning iteration (alabi/utility.py:1030-1163 calls the objective and its gradient point by point): x [d] in, out [2 + 2 d] =
 * mu, var, dmu[d], dvar[d]; pinned staging buffers of the handle, one synchronisation. */
int alabi_gp_predict_grad_point(alabi_gp* gp, const double* x /* host [d] */, double* out /* host [2 + 2 d] */, void* stream);

/* solver.log_determinant and -gp.log_likelihood(y) -- alabi/core.py:1248; gp_utils.py:139.
 * Both SYNCHRONISE the stream and write one host double. */
int alabi_gp_logdet(alabi_gp* gp, double* out, void* stream);
int alabi_gp_nll(alabi_gp* gp, double* out, void* stream);

/* d logL / d p of the log marginal likelihood at the current hyper-parameters, analytic:
 * 0.5 tr((alpha alpha^T - K^-1) dK/dp) and sum(alpha) for the mean.  Replaces george's
 * gp.grad_log_likelihood(y) (reference call sites alabi/core.py:1261, alabi/gp_utils.py:165).
 * grad_out (HOST, d + 4 doubles): [mean, log white noise, log amplitude, log alpha (0 unless the kernel is the
 * rational quadratic), log_M_0 .. log_M_{d-1}].  Needs alabi_gp_compute and alabi_gp_set_y.  Synchronises. */
int alabi_gp_grad_log_likelihood(alabi_gp* gp, double* grad_out, void* stream);

/* Inspection (tests, GP-protocol adapter: gp._alpha, utility.py:577; solver factor). */
int alabi_gp_get_alpha(alabi_gp* gp, double* alpha_out /* [N] */, void* stream);
int alabi_gp_get_factor(alabi_gp* gp, double* L_out /* [N,N] row-major, upper part zero */,
                        void* stream);
/* gp.solver.get_inverse() -- alabi/utility.py:610 (the finite-difference acquisition gradient): K^-1 [N,N] row-major, both
 * triangles, = W^T W on the matrix cores from the cached W = L^-1 of the current factor (built on demand). */
int alabi_gp_get_inverse(alabi_gp* gp, double* Kinv_out, void* stream);
int alabi_gp_n(alabi_gp* gp, int* n /* host */);
/* How the last alabi_gp_compute factorised (tests, soak runs): 0 = not computed, 1 = launch per step, 2 = the one-launch task queue,
 * 3 = the queue's wait ran out and the matrix was assembled and factorised again step by step. */
int alabi_gp_last_factor_path(alabi_gp* gp, int* path /* host */);

/* kernel.get_value(x1, x2) -- alabi/utility.py:549, :607.  K_out is [n1,n2] row-major,
 * no white noise. */
int alabi_kernel_matrix(const double* X1, int n1, const double* X2, int n2, int d,
                        int kernel_type, double log_alpha, double log_amp,
                        const double* log_M /* host [d] */, double* K_out, void* stream);

/* Acquisition scan: utility.bape_utility / agp_utility / jones_utility evaluated on M
 * candidates + the arg-min that utility.minimize_objective takes over restarts --
 * alabi/utility.py:729-810, :629-701, :853-946, :1149-1163; alabi/core.py:1601-1621.
 * Xs [M,d]; bounds host [d,2]; u [M] or NULL; best_val / best_idx are HOST outputs
 * (SYNCHRONISES).  Candidates outside the open box get +inf. */
int alabi_utility_scan(alabi_gp* gp, int algo, const double* Xs, long long M,
                       const double* bounds, double y_best, double* u, double* mu_out,
                       double* var_out, double* best_val, long long* best_idx, void* stream);
/* Continuous polish of an acquisition optimum: projected limited-memory BFGS from x0 inside the (open) box, value and gradient of the
 * acquisition function from ONE alabi_gp_predict_grad_point per evaluation, the optimiser itself on the host beside it.  Stands for the
 * local optimisation of alabi/utility.py:1030-1163 (scipy L-BFGS-B around one prediction per objective call) on top of the batched
 * scan; never worse than x0.  bounds [d][2], x0 / x_out [d], all host; *nevals = evaluations made. */
int alabi_utility_polish(alabi_gp* gp, int algo, const double* x0, const double* bounds, double y_best, int maxiter,
                         double* x_out, double* u_out, int* nevals, void* stream);
/* The epilogue alone on caller-supplied (mu, var): used by the parity tests. */
int alabi_utility_eval(int algo, const double* Xs, long long M, int d, const double* bounds,
                       double y_best, const double* mu, const double* var, double* u,
                       void* stream);

/* Ensemble sampler: replaces emcee.EnsembleSampler(W, d, sm.lnprob).run_mcmc(p0, nsteps)
 * with the StretchMove -- alabi/core.py:2319-2325, lnprob = surrogate mean + box prior
 * (alabi/core.py:2073-2100, utility.py:218-275).  n_ensembles >= 1 independent ensembles of
 * W walkers each ("independent chains") share every launch; walker ids are global
 * (ensemble e owns rows [e*W, (e+1)*W) of coords / logp / chain) and walkers only interact
 * inside their ensemble.  bounds is a host array [d,2] in the GP's (scaled) coordinates. */
int alabi_ens_create(alabi_gp* gp, int W, int d, int n_ensembles, const double* bounds,
                     unsigned long long seed, alabi_ens** out);
int alabi_ens_destroy(alabi_ens* ens);
/* Independent normal priors on selected coordinates on top of the uniform box: the reference's lnprior_normal
 * (alabi/utility.py:370-378; passed to run_emcee as prior_fn, alabi/core.py:2108).  mean[d], std[d] on the HOST in the
 * sampler's coordinates; a non-finite mean or std <= 0 means "no normal prior on this coordinate".  Adds
 * sum_k norm.logpdf(x_k, mean_k, std_k) to the log-probability.  Takes effect from the next call on. */
int alabi_ens_set_normal_prior(alabi_ens* ens, const double* mean, const double* std);

/* Log-probability inside the box = scale * (GP mean) + shift.  Default (1, 0) is the reference's lnprob with identity
 * scalers; an affine y_scaler (alabi/core.py:1483-1502: y = y_scaler.inverse_transform(gp.predict(...))) sets its slope and
 * offset here, an affine theta_scaler is absorbed by running the ensemble in scaled coordinates (the stretch move is
 * affine-invariant).  scale > 0.  Affects alabi_ens_lnprob / run / half_step from the next call on. */
int alabi_ens_set_logp_affine(alabi_ens* ens, double scale, double shift);

/* Non-affine y scalers the reference ships (alabi/utility.py:62-71; un-scaling per prediction at alabi/core.py:1483-1502):
 * the log-probability becomes map(scale * GP mean + shift) with kind 0 identity (default), 1 nlog_scaler's inverse -10^x,
 * 2 log_scaler's inverse 10^x.  Applied inside every ensemble kernel, once per proposal. */
int alabi_ens_set_logp_map(alabi_ens* ens, int kind);

/* Enable / disable the persistent dataflow kernel for alabi_ens_run on this handle (default: enabled when the
 * ensemble fits one workgroup per CU).  Returns ALABI_BAD_ARGUMENT when enabling is impossible. */
int alabi_ens_set_stream(alabi_ens* ens, int enabled);
/* Which path the last alabi_ens_run took: 1 = persistent dataflow kernel with the training set in one workgroup's
 * registers (ens_stream_kernel: N <= 2048, small d), 3 = persistent group kernel (ens_group_kernel: training set
 * partitioned over groups of workgroups, kernel sums on the matrix cores; d <= 30), 0 = one launch per half step. */
int alabi_ens_last_path(alabi_ens* ens, int* path /* host */);
/* Blocking of the last group-kernel launch of alabi_ens_run (zeros before the first one), so a parity test can pin WHICH
 * instantiation of ens_group_kernel it compared with the oracle: out[0] Q (16-proposal tiles per group), [1] G (members per
 * group), [2] NG (groups per ensemble), [3] RT (point tiles per wave held in registers), [4] tpm (point tiles per member),
 * [5] ltw (point tiles per wave staged in LDS), [6] KS (MFMA k-steps = ceil((d + 2) / 4)), [7] LDS bytes per workgroup. */
int alabi_ens_group_plan(alabi_ens* ens, int* out /* host [8] */);

/* log-probability of every walker (surrogate mean + box prior): coords [E*W,d] -> logp [E*W]. */
int alabi_ens_lnprob(alabi_ens* ens, const double* coords, double* logp, void* stream);

/* The surrogate part alone, y_scaler^-1(GP mean), at M arbitrary points [M,d] in the sampler's coordinates -- no box
 * gate, no prior (alabi/core.py:1446-1508 surrogate_log_likelihood as lnprob's like_fn, :2073-2100). */
int alabi_ens_surrogate(alabi_ens* ens, const double* points, int M, double* like, void* stream);

/* Run nsteps full stretch-move steps on one GPU.  coords [E*W,d] and logp [E*W] are updated in
 * place; chain [nsteps/thin_by, E*W, d] and chain_logp [nsteps/thin_by, E*W] receive every
 * thin_by-th state (either may be NULL); n_accept [E*W] int64 is ADDED to.  step0 is the
 * global index of the first step (counter-based RNG: draws depend on (seed, step, global
 * walker id) only).  Enqueues on `stream`; does not synchronise with the kernels. */
int alabi_ens_run(alabi_ens* ens, double* coords, double* logp, long long step0,
                  long long nsteps, int thin_by, double a, double* chain, double* chain_logp,
                  long long* n_accept, void* stream);

/* sampler.get_autocorr_time(tol=0) -- alabi/mcmc_utils.py:45, alabi/core.py:2387 (emcee.autocorr.integrated_time): the FFT part.
 * chain [n_t, n_w, n_d] (device, as alabi_ens_run stores it) -> acf_mean [n_d, n_t] (device): for every dimension the mean over
 * the walkers of acf = IFFT(|FFT(x - mean(x), 2 n)|^2)[:n_t] / acf[0], n = the next power of two >= n_t.  Hand-written four-step
 * transforms in LDS (alabi_amd/csrc/chain_acf.hip): no run-time kernel compilation per transform length.  n_t <= 2^21.  The Sokal
 * window and tau = 2 cumsum(acf) - 1 follow on the host (alabi_amd/mcmc_utils.py).  Synchronises the stream. */
int alabi_chain_autocorr(const double* chain, long long n_t, int n_w, int n_d, double* acf_mean, void* stream);

/* Multi-GPU building blocks for ONE sharded ensemble (n_ensembles == 1; alabi_amd/dist.py
 * drives them around an RCCL all-gather): draw the proposal records of steps
 * [step0, step0+nsteps) into the handle, then apply one HALF step (split 0 or 1 of local
 * step `t`) to the walkers whose position in that half's list lies in [part_begin, part_end). */
int alabi_ens_draw(alabi_ens* ens, long long step0, int nsteps, double a, void* stream);
int alabi_ens_half_step(alabi_ens* ens, double* coords, double* logp, int t, int split,
                        int part_begin, int part_end, long long* n_accept, void* stream);
/* Generic log-probability (n_ensembles == 1): the reference's lnprob = like_fn(theta) + prior_fn(theta) accepts ANY
 * Python callables (alabi/core.py:2073-2100, :2253-2280; docstring example :2236-2239).  A half step of drawn local step
 * `t` is split around the host call: alabi_ens_propose writes the proposals of that half in LIST order to q [nS,d]
 * (nS = ceil(W/2) for split 0, floor(W/2) for split 1) and, if `like` is not NULL, the surrogate part
 * y_scaler^-1(GP mean) at each proposal (-inf outside the box when gate_box != 0); the caller forms
 * lp_new [nS] = like (or its own like_fn) + prior_fn(q); alabi_ens_accept applies emcee's accept test
 * (d-1) ln z + lp_new - logp > ln u' with the step's draws and updates coords / logp / n_accept in place.
 * A NaN lp_new rejects. */
int alabi_ens_propose(alabi_ens* ens, const double* coords, int t, int split, int gate_box,
                      double* q, double* like, void* stream);
int alabi_ens_accept(alabi_ens* ens, double* coords, double* logp, int t, int split,
                     const double* q, const double* lp_new, long long* n_accept, void* stream);
/* ONE ensemble sharded over the GPUs of a node with the whole step loop in the library (n_ensembles == 1): each rank
 * applies every half step to its slice of the active list and the updated (coords, logp) rows are exchanged with one
 * all-gather per half step, enqueued on `stream` (RCCL ncclAllGather over xGMI; librccl.so is loaded with dlopen on first
 * use).  Replaces the reference's process pool under emcee (alabi/core.py:2300, :2322).  Every rank passes the same
 * initial coords / logp / step0 and ends with the same coords / logp / chain; n_accept [W] is ADDED to (summed over ranks).
 * alabi_dist_unique_id: rank 0 obtains 128 bytes and distributes them (e.g. torch.distributed broadcast), then every rank
 * calls alabi_dist_comm_create with them (nranks == 1 needs no id).  The *_callback form carries a host function instead
 * of RCCL: the test rig for several ranks on one GPU (it must complete the exchange before returning). */
typedef struct alabi_comm alabi_comm;
typedef int (*alabi_allgather_fn)(const double* send_dev, double* recv_dev, long long count_per_rank, void* user, void* stream);
int alabi_dist_unique_id(void* id_out /* host, 128 bytes */);
int alabi_dist_comm_create(const void* id /* host, 128 bytes */, int rank, int nranks, alabi_comm** out);
int alabi_dist_comm_create_callback(alabi_allgather_fn fn, void* user, int rank, int nranks, alabi_comm** out);
int alabi_dist_comm_destroy(alabi_comm* comm);
/* Counters of alabi_ens_run_sharded on this communicator (host int64[4]): [0] full chunks replayed from the captured hipGraph,
 * [1] chunks enqueued launch by launch, [2] graph captures, [3] 1 once a run failed on this rank with peers possibly inside the
 * collective (the communicator is then dead: every later run returns ALABI_HIP_ERROR; end the process). */
int alabi_dist_comm_stats(alabi_comm* comm, long long* out /* host [4] */);
int alabi_ens_run_sharded(alabi_ens* ens, alabi_comm* comm, double* coords, double* logp, long long step0,
                          long long nsteps, int thin_by, double a, double* chain, double* chain_logp,
                          long long* n_accept, void* stream);
/* copy of the walker lists of drawn local step t: order_out[W] int32 (device), n0 (host). */
int alabi_ens_step_lists(alabi_ens* ens, int t, int* order_out, int* n0, void* stream);

/* Test entry (n_ensembles == 1): one full step from caller-supplied draws keyed by WALKER id
 * (bit-exact index-arithmetic fixtures).  order[W] int32 lists set 0 then set 1; partner[W]
 * int32 indexes the complementary list.  Out-of-range indices make that proposal a no-op. */
int alabi_ens_step_with_randoms(alabi_ens* ens, double* coords, double* logp,
                                const int* order, int n0, const double* u_z,
                                const int* partner, const double* u_acc, double a,
                                long long* n_accept, void* stream);
/* Test entry: the device's counter-based draws for one step, copied out in LIST order
 * (position p of ensemble e at [e*W + p]): order = global walker id, partner = index drawn
 * into the complementary list, u_z / u_acc raw uniforms, cw (nullable) = partner's global
 * walker id, zz (nullable) = stretch factor.  n0 is a host int. */
int alabi_ens_export_draws(alabi_ens* ens, long long step, double a, int* order, int* n0,
                           double* u_z, int* partner, double* u_acc, int* cw, double* zz,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ALABI_HIP_H */
