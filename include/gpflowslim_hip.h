/*
 * gpflowslim_hip.h -- C ABI of libgpflowslim_hip.so
 *
 * MI355X (gfx950) implementation of GPflow-Slim's exact-GP hot path:
 *   kernels.K -> Cholesky -> triangular solves -> log-det / posterior.
 *
 * The reference has no FFI of its own (it is pure Python over TensorFlow 1.x
 * builtins), so every entry point below cites the reference *call site* whose
 * TensorFlow ops it replaces.  Paths are relative to the reference checkout
 * (gpflowSlim/...).
 *
 * Conventions
 *   - all matrices are C-contiguous row-major fp64 (numpy default);
 *   - "host" pointers are ordinary process memory owned by the caller;
 *     device buffers live inside the handle and are never returned;
 *   - return value 0 = ok, <0 = argument / HIP error (text: gps_last_error),
 *     LAPACK-style "not positive definite" is reported through *info > 0
 *     (order of the first non-positive pivot), the analogue of TensorFlow
 *     raising InvalidArgumentError from tf.cholesky (models/gpr.py:70);
 *   - a handle is bound to one GPU and is NOT thread-safe; calls return after
 *     the handle's stream has been synchronised unless stated otherwise.
 */
#ifndef GPFLOWSLIM_HIP_H
#define GPFLOWSLIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPS_MAX_DIMS   32   /* active dims per primitive kernel            */
#define GPS_MAX_NODES  64   /* instructions per kernel program             */
#define GPS_MAX_STACK   4   /* evaluation-stack depth of a kernel program  */

#define GPS_OK              0
#define GPS_ERR_ARG        -1
#define GPS_ERR_HIP        -2
#define GPS_ERR_STATE      -3
#define GPS_ERR_UNSUPPORTED -4

/* Kernel program: the covariance function as a reverse-Polish program.
 * Mirrors Combination.__init__'s flattening + Sum.K / Product.K left folds
 * (kernels.py:1009-1037, 1071-1084).  A primitive pushes k(x_i, x_j); ADD /
 * MUL pop two values and push the result (left operand = deeper element).   */
enum gps_kern_op {
  GPS_K_RBF      = 1,  /* RBF.K            kernels.py:436-439 */
  GPS_K_MATERN12 = 2,  /* Matern12.K       kernels.py:573-577 */
  GPS_K_MATERN32 = 3,  /* Matern32.K       kernels.py:589-594 */
  GPS_K_MATERN52 = 4,  /* Matern52.K       kernels.py:605-610 */
  GPS_K_PERIODIC = 5,  /* Periodic.K       kernels.py:806-819 */
  GPS_K_WHITE    = 6,  /* White.K          kernels.py:332-338 */
  GPS_K_CONSTANT = 7,  /* Constant.K       kernels.py:345-350; also the
                          scalars of Combination.const_list :1026-1027      */
  GPS_K_EXPONENTIAL = 8, /* Exponential.K  kernels.py:560-565 */
  GPS_K_SQDIST   = 9,  /* Stationary.square_dist  kernels.py:408-421: variance * max(0, |a|^2 + |b|^2 - 2 a.b), a = x / l (the
                          callable form of the distance every stationary primitive is built on; not differentiable here) */
  GPS_K_EUCLID   = 10, /* Stationary.euclid_dist  kernels.py:424-426: variance * sqrt(square_dist + 1e-12)               */
  GPS_K_ADD      = 16, /* Sum.K     reduce(tf.add, ...)      :1073          */
  GPS_K_MUL      = 17, /* Product.K reduce(tf.multiply, ...) :1081          */
  /* Neural Kernel Network (neural_kernel_network/neural_kernel_network.py:41-47): a program that
   * contains these ops is "layered": up to 8 primitive nodes first (their values form the input
   * vector), then the layers in order.  active_dims[0] = layer index.                           */
  GPS_K_NKN_LINROW  = 32, /* one output of a Linear layer (neural_kernel_network_wrapper.py:90-120):
                             n_dims = input width, lengthscales[0..n_dims) = weights, variance = bias */
  GPS_K_NKN_PRODUCT = 33, /* Product layer (:134-152): n_dims = step (2, 3 or 4)                  */
  GPS_K_NKN_ACT     = 34  /* Activation layer (:155-173): period = 1 -> exp                       */
};

typedef struct gps_kern_node {
  int32_t op;                           /* enum gps_kern_op                  */
  int32_t n_dims;                       /* active dims of a primitive        */
  int32_t active_dims[GPS_MAX_DIMS];    /* column indices into X (Kernel._slice, kernels.py:217-253) */
  double  variance;                     /* constrained value                 */
  double  period;                       /* Periodic only                     */
  double  lengthscales[GPS_MAX_DIMS];   /* per active dim (ARD) or repeated  */
} gps_kern_node_t;

typedef struct gps_handle_s* gps_handle_t;

/* ---- life cycle ------------------------------------------------------- */
int  gps_create(int device_id, gps_handle_t* out);
int  gps_destroy(gps_handle_t h);
/* Give the handle's device buffers back (K / L of a large problem is N^2 x 8 bytes and is otherwise kept for re-use);
 * the resident data set and factor are dropped -- call gps_gpr_set_data again.  Streams and options stay.  No
 * reference counterpart: TensorFlow's allocator owns the reference's device memory.                              */
int  gps_release_buffers(gps_handle_t h);
const char* gps_last_error(gps_handle_t h);
/* name: >=256 bytes.  Returns CU count, HBM bytes, and the gfx arch string. */
int  gps_device_info(gps_handle_t h, char* name, int name_len, int* n_cu,
                     int64_t* hbm_bytes, char* arch, int arch_len);

/* ---- kernels.K(X, X2) -------------------------------------------------
 * Replaces Stationary.square_dist / euclid_dist + RBF/Matern/Periodic.K +
 * Sum.K / Product.K (kernels.py:408-439, 569-610, 806-819, 1071-1084) and the
 * "+ eye(N) * variance" of models/gpr.py:69 (diag_add, symmetric case only).
 * X [n, d_all]; X2 [m, d_all] or NULL (symmetric: m is ignored, out is [n,n]).
 * K_out host [n, m].                                                        */
int gps_kmat(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
             const double* X, int64_t n, const double* X2, int64_t m,
             int64_t d_all, double diag_add, double* K_out);

/* ---- tf.cholesky (models/gpr.py:70,121; conditionals.py:84) ------------
 * A host [n,n] (lower triangle read); L_out host [n,n], upper triangle
 * zero-filled like tf.cholesky.  A and L_out may alias.                     */
int gps_potrf(gps_handle_t h, const double* A, int64_t n, double* L_out,
              int* info);

/* ---- tf.matrix_triangular_solve(L, B, lower=True) ----------------------
 * (densities.py:82; models/gpr.py:122-123; conditionals.py:87,100)
 * L host [n,n] lower; B host [n,nrhs] overwritten by the solution of
 * L X = B (trans=0) or L^T X = B (trans=1).                                 */
int gps_trsm_lower(gps_handle_t h, const double* L, int64_t n, double* B,
                   int64_t nrhs, int trans);

/* ---- GPR: device-resident fused path -----------------------------------
 * gps_gpr_set_data: GPModel.__init__ storing X (models/model.py:111-119).
 * X is uploaded once and stays in HBM; K / L never cross PCIe.              */
int gps_gpr_set_data(gps_handle_t h, const double* X, int64_t n, int64_t d_all);

/* GPR._build_likelihood exact branch (models/gpr.py:69-72) +
 * densities.multivariate_normal (densities.py:73-95):
 *   K = kern.K(X) + noise_var*I ; L = chol(K) ; alpha = L^-1 resid ;
 *   lml = -0.5*n*r*log(2pi) - r*sum(log diag L) - 0.5*sum(alpha^2).
 * resid host [n, r] = Y - mean_function(X).  Leaves L and alpha resident.   */
int gps_gpr_lml(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
                double noise_var, const double* resid, int64_t r,
                double* lml, int* info);

/* Log-marginal likelihood AND its gradient -- what TF autodiff through tf.cholesky supplies to the
 * reference's optimisers (examples/gpr.py:53-54 AdamOptimizer.minimize(objective);
 * models/model.py:172-187 L-BFGS):   d LML/d theta = 1/2 tr((A A^T - r K_y^-1) dK_y/dtheta),
 * A = K_y^-1 resid.  grad_slots: for every primitive node of the program in order, [d/d variance] then
 * stationary kernels one entry per active dim (d/d lengthscale_d; an isotropic kernel sums them),
 * Periodic [d/d lengthscale, d/d period], White / Constant nothing more -- all w.r.t. the CONSTRAINED
 * values; the caller applies the transform's chain rule.  grad_noise = d/d noise_var.
 * kinv_resid (optional) host [n, r] = A = d LML / d resid (chain rule for mean-function parameters).
 * Neural-Kernel-Network programs (GPS_K_NKN_*: neural_kernel_network_wrapper.py:90-173, whose Linear weights and biases the
 * reference trains through autodiff): after the primitives' slots, for every Linear layer in network order and every
 * output o: [d/d W[o][0 .. in-1], d/d bias[o]].  Programs with more than 8 primitive nodes return GPS_ERR_UNSUPPORTED. */
int gps_gpr_lml_grad(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
                     double noise_var, const double* resid, int64_t r, double* lml,
                     double* grad_slots, int n_slots_cap, int* n_slots_out,
                     double* grad_noise, double* kinv_resid, int* info);

/* GPR._build_predict exact branch (models/gpr.py:119-131).
 * refactor != 0: rebuild K, L, V exactly like the reference does on every
 * predict_f call ("cold"); refactor == 0: reuse L / V left by the previous
 * gps_gpr_lml / gps_gpr_predict on this handle ("warm"; prog, noise_var and
 * resid must be unchanged -- the caller vouches for that).
 * mean_out host [n_new, r]  = A^T V          (caller adds mean_function(Xnew))
 * var_out  host [n_new]     = Kdiag - colsum(A*A)       (full_cov == 0)
 *          host [n_new,n_new] = K(Xnew) - A^T A          (full_cov != 0)
 * (the tiling over r, models/gpr.py:127-131, is a host-side broadcast).     */
int gps_gpr_predict(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
                    double noise_var, const double* resid, int64_t r,
                    const double* Xnew, int64_t n_new, int full_cov,
                    int refactor, double* mean_out, double* var_out, int* info);

/* ---- conditionals.conditional / base_conditional ------------------------
 * (conditionals.py:24-66, 80-121; features.py:74-81 for Kuu/Kuf).
 * Device-resident form: Kmm = kern.K(Z) + jitter*I, Kmn = kern.K(Z, Xnew) are
 * built on the GPU.  f host [m, k].  q_sqrt: NULL, or host [m, k]
 * (q_sqrt_ndim == 2), or host [k, m, m] lower-triangular factors
 * (q_sqrt_ndim == 3; i.e. the reference's [m, m, k] transposed to k-major).
 * fmean_out host [n_new, k]; fvar_out host [n_new, k] (full_cov == 0) or
 * [k, n_new, n_new] (full_cov != 0; caller transposes to [n,n,k]).          */
int gps_conditional(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
                    const double* Z, int64_t m, int64_t d_all, double jitter,
                    const double* Xnew, int64_t n_new,
                    const double* f, int64_t k,
                    const double* q_sqrt, int q_sqrt_ndim,
                    int white, int full_cov,
                    double* fmean_out, double* fvar_out, int* info);

/* base_conditional on caller-supplied matrices (conditionals.py:80-121).
 * Kmn host [m, n_new]; Kmm host [m, m]; Knn host [n_new] or [n_new, n_new]. */
int gps_base_conditional(gps_handle_t h, const double* Kmn, const double* Kmm,
                         const double* Knn, int64_t m, int64_t n_new,
                         const double* f, int64_t k,
                         const double* q_sqrt, int q_sqrt_ndim,
                         int white, int full_cov,
                         double* fmean_out, double* fvar_out, int* info);

/* ---- SVGP bound (Hensman et al. 2015) with the Gaussian likelihood: models.SVGP._build_likelihood
 * (models/svgp.py:108-125) = scale * sum variational_expectations (likelihoods.py:186-188) over the conditional
 * q(f) = conditional(X, Z, kern, q_mu, q_sqrt, white) (conditionals.py:24-121, features.py:74-81)
 * - gauss_kl(q_mu, q_sqrt, Kuu + jitter I or None) (kullback_leiblers.py:26-105, models/svgp.py:101-106).
 * One Cholesky of Kuu serves the conditional and the KL; Kuu, Kuf^T [n, m] and the q_sqrt products never leave HBM.
 * Z host [m, d]; X host [n, d]; yres host [n, k] = Y - mean_function(X); q_mu host [m, k]; q_sqrt host [m, k]
 * (ndim 2) or [k, m, m] (ndim 3, lower triangles used); scale = num_data / n (models/svgp.py:123).
 * Outputs: *elbo; optional *kl, *var_exp_sum (unscaled).                                                     */
int gps_svgp_elbo(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
                  const double* Z, int64_t m, int64_t d_all, double jitter,
                  const double* X, int64_t n, const double* yres,
                  const double* q_mu, int64_t k, const double* q_sqrt, int q_sqrt_ndim,
                  int white, double noise_var, double scale,
                  double* elbo, double* kl, double* var_exp_sum, int* info);

/* The same bound AND its gradient -- what TF autodiff through models/svgp.py:108-125 supplies to the optimiser of
 * examples/svgp.py:159-161 (which runs with whiten=False) -- for either parametrisation (white == 0: the whitened
 * gradient at m_w = Lm^-1 q_mu, L_w = Lm^-1 L_q pulled back through that map, including its dependence on Lm):
 *   grad_slots   d/d kernel parameters, slot layout of gps_gpr_lml_grad;  grad_noise  d/d noise_var;
 *   grad_q_mu    host [m, k];  grad_q_sqrt  host, layout of q_sqrt ([m, k], or [k, m, m] with zeros above the diagonals);
 *   grad_mean    (optional) host [n, k] = d/d mean_function(X)  (chain rule for mean-function parameters);
 *   grad_Z       (optional) host [m, d_all] = d/d inducing inputs (features.py:65 makes Z a Parameter and
 *                examples/svgp.py:161 minimises over every variable of the graph, Z included).
 * All with respect to the CONSTRAINED values.                                                                    */
int gps_svgp_elbo_grad(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
                       const double* Z, int64_t m, int64_t d_all, double jitter,
                       const double* X, int64_t n, const double* yres,
                       const double* q_mu, int64_t k, const double* q_sqrt, int q_sqrt_ndim,
                       int white, double noise_var, double scale,
                       double* elbo, double* grad_slots, int n_slots_cap, int* n_slots_out, double* grad_noise,
                       double* grad_q_mu, double* grad_q_sqrt, double* grad_mean, double* grad_Z, int* info);

/* Vector-Jacobian product of the kernel-matrix build -- reverse-mode autodiff through kern.K(X, X2) (kernels.py:408-439,
 * 1071-1084; neural_kernel_network.py:41-47):  grad_slots[s] = sum_ij W[i][j] d k(X_i, X2_j) / d theta_s  for a
 * caller-supplied cotangent W host [n, m] (X2 == NULL: K(X, X), W [n, n] used as given).  Slot layout of
 * gps_gpr_lml_grad.  The building block of gps_svgp_elbo_grad, exported for models assembled by the caller.      */
int gps_kmat_vjp(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* X, int64_t n,
                 const double* X2, int64_t m, int64_t d_all, const double* W, double* grad_slots,
                 int n_slots_cap, int* n_slots_out);
/* The same contraction differentiated in the POINTS: grad_X host [n, d_all] = d/dX sum_ij W[i][j] k(X_i, X2_j)
 * (X2 == NULL: K(X, X), both arguments move).  Reverse mode through kern.K with respect to its first argument --
 * what a trainable InducingPoints.Z receives (features.py:65, 74-81).                                             */
int gps_kmat_input_vjp(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* X, int64_t n,
                       const double* X2, int64_t m, int64_t d_all, const double* W, double* grad_X);

/* gauss_kl(q_mu, q_sqrt, K) (kullback_leiblers.py:26-105): KL[N(q_mu, q_sqrt q_sqrt^T) || N(0, K)], summed over
 * the k independent columns; K host [m, m] or NULL (p = N(0, I)).  tf.cholesky(K) (:51), alpha = Lp^-1 q_mu (:52),
 * tr(K^-1 S_q) through Lp^-T (diagonal q_sqrt: row sums of squares of Lp^-T instead of forming K^-1, :84-90) or
 * through Lp^-1 L_q per latent (:92-94) run on the device.  q_sqrt layouts as in gps_svgp_elbo.                 */
int gps_gauss_kl(gps_handle_t h, const double* K, int64_t m, const double* q_mu, int64_t k,
                 const double* q_sqrt, int q_sqrt_ndim, double* kl, int* info);

/* ---- SGPR (sparse GP regression, Titsias 2009) -------------------------------------------------
 * models/sgpr.py:121-153 (_build_likelihood: the collapsed bound) and :155-189 (_build_predict).
 * Z host [m, d] inducing inputs, X host [n, d], resid host [n, r] = Y - mean_function(X).
 * bound_out (optional): the bound.  If n_new > 0: mean_out host [n_new, r] (caller adds the mean function),
 * var_out host [n_new] (full_cov == 0) or [n_new, n_new] (caller tiles over r, sgpr.py:181-188).          */
int gps_sgpr(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
             const double* Z, int64_t m, const double* X, int64_t n, int64_t d_all,
             double jitter, double noise_var, const double* resid, int64_t r,
             const double* Xnew, int64_t n_new, int full_cov,
             double* bound_out, double* mean_out, double* var_out, int* info);

/* The SGPR bound and its gradient -- what TF autodiff through models/sgpr.py:121-153 supplies to the optimiser (SGPR keeps the
 * inducing inputs trainable: features.py:65 makes Z a Parameter): grad_slots (kernel parameters, slot layout of gps_gpr_lml_grad),
 * grad_noise, grad_mean (optional, host [n, r] = d/d mean_function(X)), grad_Z (optional, host [m, d_all]); all with respect
 * to the constrained values.  Reverse mode at the matrix level (two Cholesky adjoints, kernel-matrix VJPs); every O(M^2 N)
 * product on the fp64 MFMA, resident in HBM.                                                                      */
int gps_sgpr_grad(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
                  const double* Z, int64_t m, const double* X, int64_t n, int64_t d_all,
                  double jitter, double noise_var, const double* resid, int64_t r,
                  double* bound, double* grad_slots, int n_slots_cap, int* n_slots_out, double* grad_noise,
                  double* grad_mean, double* grad_Z, int* info);

/* The same for the FITC log-likelihood (models/sgpr.py:229-290): arguments and outputs as gps_sgpr_grad.          */
int gps_fitc_grad(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
                  const double* Z, int64_t m, const double* X, int64_t n, int64_t d_all,
                  double jitter, double noise_var, const double* resid, int64_t r,
                  double* bound, double* grad_slots, int n_slots_cap, int* n_slots_out, double* grad_noise,
                  double* grad_mean, double* grad_Z, int* info);

/* GP regression with the FITC approximation: models.GPRFITC._build_likelihood / _build_predict
 * (models/sgpr.py:229-318: Luu = chol(Kuu), V = Luu^-1 Kuf, nu = Kdiag - colsumsq(V) + sigma^2,
 * L = chol(I + (V/nu) V^T), gamma = L^-1 V (err/nu)).  Same arguments, layouts and outputs as gps_sgpr;
 * bound_out receives the FITC log-likelihood.  Kdiag must be constant (stationary kernels).          */
int gps_fitc(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes,
             const double* Z, int64_t m, const double* X, int64_t n, int64_t d_all,
             double jitter, double noise_var, const double* resid, int64_t r,
             const double* Xnew, int64_t n_new, int full_cov,
             double* bound_out, double* mean_out, double* var_out, int* info);
/* ---- the sparse models with the data points sharded over ranks (no reference counterpart) ------------
 * Every term of the SGPR / FITC bounds that touches the N data points is a sum over them (models/sgpr.py:138-153,
 * 241-290: A A^T, A err, err^T err, sum log nu, N itself).  With a collective installed, gps_sgpr / gps_fitc take X and
 * resid as THIS RANK'S SHARD of the data, add the partial sums of all ranks up with ONE in-place all-reduce of
 * m_pad^2 + m_pad (r + 1) + 4 doubles in `dev_buf`, and finish redundantly on every rank (Kuu is factored by every
 * rank: M = 4096 -> 7 ms); the bound, and predictions at any Xnew, are then those of the whole data set on every rank.
 * `fn(ctx, dev_ptr, count)` must sum `count` doubles at device address `dev_ptr` (inside `dev_buf`, which the caller
 * allocates so that the collective library knows the memory) over all ranks in place and return 0 once the result is
 * visible to the device (the library has synchronised its stream before the call).  fn == NULL removes it.
 * gps_sgpr_grad / gps_fitc_grad refuse to run while a collective is installed.
 * The SVGP bound (models/svgp.py:108-125) needs no buffer: it is scale * sum_i var_exp_i - KL, so rank p evaluates
 * gps_svgp_elbo(_grad) on its shard with the option "svgp_kl_weight" = 1 / P and the caller adds the P results (bound and
 * every gradient are linear in the per-rank terms).                                                              */
typedef int (*gps_allreduce_fn)(void* ctx, void* dev_ptr, int64_t count);
int gps_set_allreduce(gps_handle_t h, gps_allreduce_fn fn, void* ctx, void* dev_buf, int64_t capacity_doubles);
/* doubles gps_sgpr / gps_fitc reduce for m inducing points and r outputs (size dev_buf at least this large) */
int gps_allreduce_doubles(int64_t m, int64_t r, int64_t* out);

/* ---- native collectives: RCCL called from the library itself (csrc/comm_rccl.hip; no reference counterpart) -----------
 * "ncclBroadcast (RCCL) per panel, root = owner, on a dedicated stream".  librccl is opened with dlopen on
 * first use (gps_comm_load(path): a specific librccl -- e.g. the one a PyTorch wheel bundles, so that a process holds one
 * copy --, NULL: the system's).  One communicator per handle: rank 0 draws a unique id (128 bytes), the caller carries it to
 * the other ranks by whatever side channel it has (a file, MPI, torch.distributed's store), every rank calls gps_comm_init.
 *   gps_comm_exchange(h, dev_buf, count, root, mode, slot)  `count` doubles at dev_buf, complete on `root`, become complete on
 *        every rank; enqueued on the communicator's own stream after everything the handle's stream has done so far, returns
 *        at once.  mode 0: one broadcast; mode 1: scatter of P equal chunks over all of the root's links + in-place all-gather.
 *   gps_comm_wait(h, slot)          the handle's stream waits for the exchange that used `slot` (0..7)
 *   gps_comm_allreduce(h, p, n)     in-place sum of n doubles over the ranks, blocking
 *   gps_comm_install_allreduce      makes that the collective of gps_sgpr / gps_fitc on data shards (see gps_set_allreduce)
 *   gps_comm_abort(h)               gives the communicator up without waiting for the peers (ncclCommAbort): what a rank that
 *        has failed calls so that the others' pending collectives return an error instead of waiting for its part; a collective
 *        that fails inside gps_comm_exchange / gps_comm_allreduce does the same by itself (an open send / receive group is always
 *        closed first).  The handle has no communicator afterwards (gps_comm_init again for a new one).
 * Verified on hardware with the real RCCL at world size 1, and at world size 2 through a stand-in transport behind the same
 * API (tests/fake_rccl; the build box has one GPU); gpflowSlim.distributed.RcclComm drives it.                       */
int gps_comm_load(const char* path);
const char* gps_comm_load_error(void);
int gps_comm_version(int* version);
int gps_comm_unique_id(void* out, int capacity);
int gps_comm_init(gps_handle_t h, int rank, int world, const void* unique_id, int id_len);
int gps_comm_destroy(gps_handle_t h);
int gps_comm_abort(gps_handle_t h);
int gps_comm_exchange(gps_handle_t h, void* dev_buf, int64_t count, int root, int mode, int slot);
int gps_comm_wait(gps_handle_t h, int slot);
int gps_comm_allreduce(gps_handle_t h, void* dev_ptr, int64_t count);
int gps_comm_install_allreduce(gps_handle_t h, void* dev_buf, int64_t capacity_doubles);

/* terms of the last gps_sgpr / gps_fitc call, for SGPRUpperMixin.compute_upper_bound (models/sgpr.py:55-85):
 * out[0] = sum log diag(LB), out[1] = tr(A A^T) with A = L^-1 Kuf (gps_fitc: rows weighted by 1/nu),
 * out[2] = sum c^2, out[3] = Kdiag constant, out[4] = sum log nu (gps_fitc only).                      */
int gps_sparse_last_terms(gps_handle_t h, double* out5);

/* ---- measurement --------------------------------------------------------
 * Per-kernel-class accounting of the calls issued through this handle.
 * gps_profile_enable(h, 1) brackets every launch with HIP events on the
 * handle's stream (adds a few us per launch); counters accumulate until
 * gps_profile_reset.  Classes: "gemm_f64", "potrf_base", "kmat", "trsv",
 * "reduce", "other".                                                        */
int gps_profile_enable(gps_handle_t h, int on);
int gps_profile_reset(gps_handle_t h);
int gps_profile_get(gps_handle_t h, const char* klass, int64_t* launches,
                    double* ms, double* flops, double* bytes);
/* wall-clock (HIP events) of the stages of the last gps_gpr_lml /
 * gps_gpr_predict: out[0]=kmat out[1]=potrf out[2]=trsv+reductions
 * out[3]=predict solve out[4]=total (ms).                                   */
int gps_last_stage_ms(gps_handle_t h, double* out5);

/* ---- multi-GPU: 1-D block-cyclic column Cholesky, one process per GPU ------------------------
 * No reference counterpart (the reference is single-device); the oracle is the single-GPU result.
 * Rank `part` of `nparts` owns block columns c with c % nparts == part (width nb, multiple of 128),
 * builds and updates only those, and receives every factored panel so that L ends up replicated
 * (warm predict_f afterwards needs no exchange).  (Y - m)^T rides along as augmented rows under K: the
 * panel solves and trailing updates turn them into alpha^T = (L^-1 (Y - m))^T block by block
 * (densities.py:82), so no forward substitution over the replicated factor is needed; each panel message
 * ends with that panel's  sum log L_ii,  sum alpha^2  and not-positive-definite info word, and every rank
 * adds them in panel order: LML and info are bit-identical on all ranks.
 * The library supplies the per-step pieces; the caller moves the panel message (RCCL / xGMI through
 * torch.distributed: scatter + all-gather, root = owner) between gps_dist_panel_factor and gps_dist_unpack:
 *
 *   gps_dist_begin(...)                       build owned columns of K + noise I, augmented rows
 *   for j in panels:  owner: gps_dist_panel_factor(j, buf)   -> message in comm buffer `buf`
 *                     exchange(comm[buf][0 : gps_dist_msg_doubles(j)], root = j % nparts)
 *                     others: gps_dist_unpack(j, buf)
 *                     all:    gps_dist_update(j, c_lo, c_hi, lane)  (owned columns in [c_lo, c_hi), c > j)
 *   gps_dist_finish(&lml, &info)              per-panel scalars added in panel order
 *
 * Two lanes: everything runs on the handle's stream (gps_set_stream(h, s, 1) installs the caller's HIP
 * stream s, e.g. a torch stream, so that library kernels and the collective are ordered on it;
 * gps_set_stream(h, NULL, 0) restores the handle's own stream) except gps_dist_update(..., lane = 1),
 * which launches on the stream given to gps_dist_set_bulk_stream: the schedule keeps the bulk of every
 * trailing update there and orders the lanes with events (gpflowSlim/distributed.py).                */
int gps_set_stream(gps_handle_t h, void* hip_stream, int external);
int gps_dist_begin(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, double noise_var,
                   const double* resid, int64_t r, int nparts, int part, int64_t nb,
                   int64_t* n_panels, int64_t* msg_doubles_max);
int gps_dist_msg_doubles(gps_handle_t h, int64_t j, int64_t* out);
int gps_dist_set_comm(gps_handle_t h, void* dev_buf0, void* dev_buf1);
/* Storage of the factor (option "dist_partitioned"; no reference counterpart -- SURVEY 8e "GPU g owns block-columns
 * j = g (mod P) ... 1.07 GB of K/L per GPU"):
 *   1 (default)  partitioned: the handle holds only its own block columns (8 N^2 / P bytes); a panel is read by the
 *                trailing updates from the comm buffer it arrived in, so the caller provides
 *                gps_dist_comm_bufs_needed() (= 3) buffers with gps_dist_set_comm_bufs and moves panel j through
 *                buffer j % count; gps_dist_unpack only keeps the panel's scalars.  Afterwards the factor is
 *                distributed: predictions stream the panels once more (gps_dist_solve_*).
 *   0            replicated: every received panel is copied into an [N, N] buffer on every rank (two comm buffers);
 *                the whole factor is resident everywhere afterwards (warm gps_gpr_predict works).                */
int gps_dist_set_comm_bufs(gps_handle_t h, void* const* dev_bufs, int count);
int gps_dist_comm_bufs_needed(gps_handle_t h, int* count);
/* The whole factorisation from inside the library: gps_dist_begin ... gps_dist_finish with the two-lane look-ahead schedule
 * (gpflowSlim/distributed.py::block_column_schedule ported statement for statement), the handle's native communicator
 * (gps_comm_init) for the panel exchange and HIP streams / events for the lanes -- no host-language call per panel.
 * Collective: every rank calls it with the same arguments; same result bits on every rank; info as gps_dist_finish.
 * exchange_mode: 0 broadcast, 1 scatter + all-gather.                                                            */
int gps_dist_lml(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, double noise_var, const double* resid,
                 int64_t r, int64_t nb, int lookahead, int exchange_mode, double* lml, int* info);
/* predict_f (models/gpr.py:119-131, full_cov == 0) from a partitioned factor: every rank passes its shard of the test
 * points; the owner of panel j packs it again (gps_dist_solve_pack: same message layout as the factorisation's, length
 * gps_dist_msg_doubles(j)), the caller exchanges it, every rank applies it (gps_dist_solve_apply: block column j of
 * A^T = Kx^T L^-T becomes final, the columns to its right take its update, alpha_j is picked up from the augmented rows);
 * gps_dist_solve_finish returns mean [n_new, r] (caller adds the mean function) and var [n_new].  Any comm slot may be
 * used (two alternate).                                                                                          */
int gps_dist_solve_begin(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Xnew, int64_t n_new);
int gps_dist_solve_pack(gps_handle_t h, int64_t j, int buf);
int gps_dist_solve_apply(gps_handle_t h, int64_t j, int buf);
int gps_dist_solve_finish(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, double* mean_out, double* var_out);
/* ... and the whole streamed prediction from inside the library (native communicator; after gps_dist_lml): this rank's
 * shard of the test points (n_new may be 0), mean [n_new, r] without the mean function, var [n_new].               */
int gps_dist_predict(gps_handle_t h, const gps_kern_node_t* prog, int n_nodes, const double* Xnew, int64_t n_new,
                     int exchange_mode, double* mean_out, double* var_out);
/* device bytes held by the handle's own buffers (tests of the 8 N^2 / P + O(N nb) bound) */
int gps_device_bytes(gps_handle_t h, int64_t* bytes);
int gps_dist_set_bulk_stream(gps_handle_t h, void* hip_stream);
int gps_dist_panel_factor(gps_handle_t h, int64_t j, int buf);
int gps_dist_unpack(gps_handle_t h, int64_t j, int buf);
int gps_dist_update(gps_handle_t h, int64_t j, int64_t c_lo, int64_t c_hi, int lane);
int gps_dist_finish(gps_handle_t h, double* lml, int* info);

/* options (diagnostics and A/B switches; the defaults are what bench.py measures).  Round 6 removed every switch of a
 * design that was measured and rejected (one launch per 128 columns of a sweep, cross-level look-ahead, trailing update
 * behind a sweep, joins carried by GEMMs, the substitution following the factorisation, left-looking follower solve) together
 * with its device code: docs/LAB_NOTES.md keeps the measurements and the commits that hold the code.
 *   "gemm_force_tile" pin the GEMM tile edge to 128 / 64 / 32 (0 = automatic: the largest tile that still gives 512 workgroups)
 *   "gemm_tail_split" 1 (default): the tiles of a partial last round of a 128x128 launch are cut into
 *                     K-slices over the idle workgroup slots; 0: one workgroup per tile
 *   "potrf_rl_max"    diagonal blocks of at most this many columns (default 4096) are factored by a right-looking
 *                     sweep over 128-column panels instead of the recursion (tf.cholesky, models/gpr.py:70); 0: recursion only
 *   "potrf_rl_group"  panels per remainder update of that sweep (default 2: K = 256)
 *   "potrf_lookahead" 1 (default): that remainder update runs on a second stream beside the next potrf_base, handed over
 *                     through device counters (never on an external stream; remainders of at least 1024 rows); the panel
 *                     solve of the block below a swept diagonal block follows the sweep on a third stream, right-looking, in
 *                     pieces of 512 columns; a 16384-column node hands the first 4096 columns of its panel solve to its child,
 *                     which runs them beside its second sweep.  0: everything on the handle's stream
 *   "follower_max_wgs" 256 (default): rectangular updates on that third stream are launched in column chunks of at
 *                     most this many 128 x 128 tiles, so that a launch never has workgroups waiting for a slot (which would
 *                     keep taking the slots the chain's short launches need); 0: one launch per update
 *   "leaf_refine"     -1 (default): the 128-column leaves of the triangular solves are refined once against the factor's
 *                     diagonal block (X0 = B W^T; R = B - X0 L11^T; X = X0 + R W^T, W = inv(L11): the accuracy of
 *                     tf.matrix_triangular_solve's substitution, conditionals.py:87,100) wherever the matrix may be ill
 *                     conditioned: gps_conditional / gps_base_conditional / gps_svgp_elbo / gps_gauss_kl / gps_sgpr /
 *                     gps_fitc / gps_potrf / gps_trsm_lower, and the GPR entry points when the bound
 *                     cond_2(K + noise I) <= (N Kdiag + noise) / noise exceeds 2e6 (the plain products are ~7 u cond from
 *                     exact, 1e-8 holds up to cond 1.3e7); 0: plain products with the block inverses; 1: always
 *   "leaf_plain_kappa" (default 1000): in refine mode a leaf whose diagonal block has kappa_2 <= this (estimated per block
 *                     after the factorisation) takes the plain product anyway -- its error eps kappa(L_jj) kappa(L) stays a
 *                     tenth below a backward-stable solve's; 0: refine every leaf
 *   "trsm_panel"      1 (default): every 512-column node of a plain triangular solve is ONE launch (trsm_panel.hip); 0: down to
 *                     128 columns launch by launch
 *   "trsm_panel_rows" 0 (default): 32 rows per workgroup and two workgroups per CU below 64 rows x the number of CUs, one
 *                     persistent workgroup per CU above; 32 / 64 force one of them, 65 = 64 rows per workgroup, not persistent
 *   "trsm_tall_ratio" 16 (default): a solve of m rows against n columns with m >= ratio * n (conditionals.py:87 at config 5's
 *                     shape) goes over its 512-column panels left-looking -- one long-K update and one launch per panel;
 *                     0: the recursive halving always
 *   "predict_inverse_blocks" 1 (default): gps_gpr_predict on at most 8192 test points (N >= 4096 padded, plain leaves) solves
 *                     A^T = Kx^T L^-T (models/gpr.py:122) against the inverses of the factor's 2048-column diagonal blocks -- every
 *                     2048-column node ONE product; the blocks are built once per factor (11 batched launches with triangular
 *                     operands, 1.3 ms at N = 32768) by the first such call; 0: the recursive solve down to 512-column launches
 *   "gpr_aug_rows"    -1 (default): below 6200 points gps_gpr_lml / _predict / _lml_grad store (Y - m)^T as augmented
 *                     rows under K and get alpha = L^-1 (Y - m) (densities.py:82) out of the factorisation itself;
 *                     0 / 1: never / always
 *   "small_n"         1 (default): GPR problems of up to 2048 padded points (16 outputs) are factored by ONE cooperative launch
 *                     (small_n.hip; one stationary primitive: the launch builds K itself); 0: launch by launch.  After four
 *                     give-ups in a row the handle sends its next 256 evaluations launch by launch (gps_profile_get
 *                     "small_n_cooldown" reads what is left of that back-off)
 *   "trsv_wave"       1 (default): L a = y and L^T a = y of the GPR entry points run as ONE wavefront launch over the
 *                     128-row blocks (trsv_wave.hip: 1.2 ms at N = 32768); 0: recursive substitution (4 N / 128 launches)
 *   "trsv_wave_refine" 1 (default): also where the leaves are refined (one refinement step per block inside the wavefront)
 *   "kmat_fast"       1 (default): one-primitive stationary programs and Sum / Product chains of primitives (programs
 *                     "p0 p1 op p2 op ...") use the stack-free kernel-matrix kernels; 0: the stack interpreter for all
 *   "kmat_mfma"       1 (default): chains of primitives compute their feature products on the matrix pipe; 2: one-primitive
 *                     programs too; 0: never
 *   "svgp_kl_weight"  weight of the KL term of gps_svgp_elbo(_grad) (default 1; 1 / P on every rank of a data-sharded run)
 *   "dist_partitioned" 1 (default): a rank of the block-column factorisation stores only its own block columns (8 N^2 / P bytes)
 *   "la_fault_inject" / "wave_fault_inject" / "small_fault_inject"  test hooks: the k-th look-ahead join / wavefront substitution /
 *                     cooperative small-N launch from now takes its give-up path (the evaluation is then re-run once through the
 *                     launch-by-launch forms; gps_profile_get(h, "lookahead_retries" | "trsv_wave_fallbacks" | "small_n_fallbacks",
 *                     &count, ...) counts it)  */
int gps_set_option(gps_handle_t h, const char* key, double value);

/* ---- diagnostics ---------------------------------------------------------
 * fp64-MFMA microbenchmark (v_mfma_f64_16x16x4_f64 issue loop on every CU):
 * measured TFLOP/s, and layout_ok = 1 when the operand / accumulator lane map
 * used by the GEMM kernel reproduces an exact integer product.              */
int gps_diag_mfma_f64(gps_handle_t h, int waves_per_simd, double* tflops,
                      int* layout_ok);
/* raw device GEMM on host matrices, for unit tests:
 * C[m,n] (op)= A[m,k] * B[n,k]^T ; op: 0 -> C -= A B^T, 1 -> C = A B^T, 2 -> C += A B^T, 3 -> C = -A B^T.
 * lower == 1: only tiles on/below the diagonal are computed (m == n); lower == 2 / 3: A (m == k) is upper / lower
 * triangular, lower == 4: B (n == k) is lower triangular -- the zero part of a triangular operand is never read.   */
int gps_diag_gemm_nt(gps_handle_t h, int op, int lower, int64_t m, int64_t n,
                     int64_t k, const double* A, const double* B, double* C);
/* ... as ONE launch over a batch of equal problems stacked row-wise: A [batch * m, k], B [batch * n, k], C [batch * m, n];
 * tri: 0 none, 1 A upper, 2 A lower, 3 B lower triangular (the form that builds the wide inverse blocks of predict_f from
 * the 128-column ones: gps_gpr_predict)                                                                             */
int gps_diag_gemm_nt_batched(gps_handle_t h, int op, int tri, int64_t batch, int64_t m, int64_t n, int64_t k,
                             const double* A, const double* B, double* C);
/* device-resident GEMM of the given shape on pseudo-random operands (lower == 1: syrk form, B = A): average
 * launch time over `reps`, plus one instrumented launch whose workgroups record
 * stamps_out[6*b + {0..5}] = start, end (100 MHz ticks), HW_ID, XCC_ID, K-loop start, K-loop end.               */
int gps_diag_gemm_timeline(gps_handle_t h, int op, int lower, int64_t m, int64_t n, int64_t k, int reps,
                           long long* stamps_out, int64_t cap_blocks, int64_t* nblocks,
                           double* ms_per_launch);
/* replace the handle's stream by one restricted to the CUs set in mask[0 .. n_words) (experiments with
 * concurrent streams; hipExtStreamCreateWithCUMask)                           */
int gps_diag_set_cu_mask(gps_handle_t h, const uint32_t* mask, int n_words);
/* phase timestamps (us) of one 128-block potrf_base launch; out7[0] = shader clock in MHz */
int gps_diag_potrf_base_stamps(gps_handle_t h, int factor, double* out7);

/* Diagnostics: the 512-column triangular solve of m rows (m a multiple of 128) against a synthetic block, timed;
 * panel = 1: as one launch (csrc/trsm_panel.hip), 0: launch by launch (blocked.hpp::trsm_rec).  backward: X L = B.
 * maxdiff_out (optional): largest |difference| between the two forms on the same input. */
int gps_diag_trsm512(gps_handle_t h, int64_t m, int backward, int panel, int reps, double* us_per_solve,
                     double* maxdiff_out);
/* the one-launch form with phase stamps of every row block (32 or 64 rows, as the launcher picks or option trsm_panel_rows
 * says): stamps_out [min(number of row blocks, cap_blocks)][32] 100 MHz ticks --
 * wave 0 in [0, 14), HW_ID / XCC_ID in [14], [15], wave 7 in [16, 30): 0 start, 1 rows loaded, then per 128-column block j:
 * 2 + 3 j its product with the block inverse done, 3 + 3 j its rows stored, 4 + 3 j the updates of the later blocks done */
int gps_diag_trsm512_stamps(gps_handle_t h, int64_t m, int backward, int reps, double* us_per_solve, long long* stamps_out,
                            int64_t cap_blocks);
/* one 128-column leaf of the triangular solves on m rows (device-resident synthetic block), average microseconds per
 * launch over `reps`: mode 0 = product with the explicit block inverse, 1 = refined against the factor's diagonal
 * block (what tf.matrix_triangular_solve's substitution delivers, conditionals.py:87,100); upper: the X L = B form;
 * resid_out: scaled residual of the solve, checked on the host                                                  */
int gps_diag_trsm_leaf(gps_handle_t h, int64_t m, int mode, int upper, int reps, double* us_per_launch,
                       double* resid_out);

#ifdef __cplusplus
}
#endif
#endif /* GPFLOWSLIM_HIP_H */
