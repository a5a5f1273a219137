/*
 * gvi_hip.h -- C ABI of the MI355X (gfx950) natural-gradient Gauss-Hermite hot path.
 *
 * Drop-in boundary for hzyu17/GaussianVI's NGD-GH path.  The reference has no FFI: its boundary is
 * two C++ contracts -- the per-factor operator surface of
 * gvibase/GVIFactorizedBase.h:128-168 / ngd/NGDFactorizedBaseGH.h:37-129 and the joint surface of
 * gvibase/GVI-GH.h:41-46,106-144 / ngd/NGD-GH.h:25-95.  Its own CUDA variant calls the device ONCE
 * PER PASS WITH ALL FACTORS (gvibase/GVI-GH-Cuda-impl.h:177-192 -> helpers/CudaOperation.h:424-436);
 * this ABI keeps that batched shape.  Every entry point cites what it replaces.
 *
 * Conventions
 *   - plain C, opaque context, int status (0 = GVI_OK), no global state, one HIP stream per context;
 *   - fp64 (GVI_F64).  GVI_F32 is declared for BASELINE config 5 and currently returns
 *     GVI_ERR_UNSUPPORTED;
 *   - all matrices ROW-MAJOR, factor-major batches: mu[K][d], Sigma[K][d][d], Vddmu[K][d][d];
 *     chain blocks D[T][n][n] (diagonal), U[T-1][n][n] (block (t, t+1)), vectors g[T][n];
 *   - functions without suffix take caller-owned HOST buffers (copy in, run, copy out, sync);
 *     the `_dev` twins take DEVICE pointers and are asynchronous on the context stream;
 *   - there is NO CPU fallback: every compute entry point needs a gfx950 device and reports
 *     GVI_ERR_HIP otherwise.
 */
#ifndef GVI_HIP_H
#define GVI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gvi_ctx gvi_ctx;
typedef int gvi_status;

enum {
  GVI_OK = 0,
  GVI_ERR_ARG = 1,         /* bad argument / shape */
  GVI_ERR_HIP = 2,         /* HIP runtime error or no device */
  GVI_ERR_UNSUPPORTED = 3, /* valid request this build does not implement */
  GVI_ERR_NOTABLE = 4,     /* (d, p) outside the tabulated 1-D rules (p <= 25) */
  GVI_ERR_STATE = 5        /* call order (e.g. ngd_* before ngd_init) */
};
enum { GVI_F64 = 0, GVI_F32 = 1 };

/* psi kinds with device implementations.  Parameter block per factor:
 *   GVI_PSI_RANGE_1D      [y, mu_p, f*b, sig_r_sq, sig_p_sq]                d = 1
 *       psi(x) = (x-mu_p)^2/(2 sig_p_sq) + (y - f b / x)^2/(2 sig_r_sq)     src/1d_example.cpp:25-35
 *   GVI_PSI_QUAD_PRIOR    [Phi (n x n) | Qinv (n x n)]                      d = 2n
 *       psi(x) = 1/2 (Phi x1 - x2)^T Qinv (Phi x1 - x2)   gp/minimum_acc_prior.h:103-106, gp/LTV_prior.h:223-226
 *   GVI_PSI_FIXED_PRIOR   [mu0 (d) | Kinv (d x d)]                          d = n
 *       psi(x) = (x-mu0)^T Kinv (x-mu0)                                     gp/fixed_prior.h:28-30
 *   GVI_PSI_HINGE_SDF_2D  [sigma, epsilon, radius]                          d >= 2 (pose = x[0:2])
 *       psi(x) = sigma * max(0, epsilon + radius - sdf(x0, x1))^2, sdf = bilinear interpolation of a 2-D
 *       signed-distance grid shared by the set (gvi_factors_set_sdf2d): the reference's planar point-robot
 *       obstacle cost (helpers/CudaOperation.h:491-523 with PlanarSDF :21-131) -- SURVEY 8(f)1
 *   GVI_PSI_HINGE_SDF_2D_BODY [sigma, epsilon, radius, slope, n_balls, L]   d >= 3 (pose = (x, z, phi) = x[0:3])
 *       planar quadrotor body: n_balls check points along the body axis, psi = sum_i sigma (slope hinge_i)^2
 *       (CudaOperation_Quad::cost_obstacle_planar / vec_balls, helpers/CudaOperation.h:565-606); same 2-D grid call
 *   GVI_PSI_HINGE_SDF_3D  [sigma, epsilon, radius]                          d >= 3 (pose = x[0:3])
 *       3-D point robot on a trilinear signed-distance field (CudaOperation_3dpR, helpers/CudaOperation.h:650-683;
 *       SignedDistanceField :133-322); grid by gvi_factors_set_sdf3d
 *   GVI_PSI_HINGE_SDF_3D_ARM [sigma, epsilon]                               any d >= ndof (joint angles = x[0:ndof])
 *       arm in DH form with collision spheres on its frames (gvi_factors_set_arm) against a 3-D field
 *       (gvi_factors_set_sdf3d): CudaOperation_3dArm::cost_obstacle + ForwardKinematics
 *       (helpers/CudaOperation.h:325-399, 686-771); as there, n_balls = the factor dimension and the DH matrices
 *       are built from single-precision cosf / sinf
 *   GVI_PSI_HOST_CALLBACK no parameters: psi is an opaque host function (the reference's
 *       std::function, ngd/NGDFactorizedBaseGH.h:30,46-48); use gvi_expand + gvi_moments_from_psi. */
enum { GVI_PSI_RANGE_1D = 0, GVI_PSI_QUAD_PRIOR = 1, GVI_PSI_FIXED_PRIOR = 2, GVI_PSI_HOST_CALLBACK = 3,
       GVI_PSI_HINGE_SDF_2D = 4, GVI_PSI_HINGE_SDF_2D_BODY = 5, GVI_PSI_HINGE_SDF_3D = 6, GVI_PSI_HINGE_SDF_3D_ARM = 7 };

const char* gvi_version(void);
/* Message of the last failing call on this context (never NULL). */
const char* gvi_last_error(const gvi_ctx* ctx);

/* ---- context: replaces CudaOperation Cuda_init / Cuda_free (helpers/CudaOperation.h:424-436) ---- */
gvi_status gvi_ctx_create(int device, int dtype, gvi_ctx** out);
gvi_status gvi_ctx_destroy(gvi_ctx* ctx);
/* Run on a caller-owned hipStream_t (e.g. the framework's current stream); NULL = own stream. */
gvi_status gvi_ctx_set_stream(gvi_ctx* ctx, void* hip_stream);
gvi_status gvi_ctx_sync(gvi_ctx* ctx);

/* ---- quadrature table, host side: replaces nwspgr('GQN', d, p, 1) + the cereal table lookup
 *      (quadrature/GH/SparseGH/nwspgr.m:32-134; quadrature/SparseGaussHermite.h:138-166) ----
 * Z [N][d] rows ascending lexicographic, w [N], idx [N][d][3] = (level, node index, sign) of every
 * coordinate (0 is canonicalised to (1,0,0)) -- the bit-exact "sigma-point index".  Any of Z/w/idx
 * may be NULL.  Needs no GPU. */
gvi_status gvi_spgh_count(int d, int p, int64_t* N);
gvi_status gvi_spgh_nodes(int d, int p, int64_t N, double* Z, double* w, int8_t* idx);

/* ---- the reference's table FILE (QuadratureWeightsMap through cereal::BinaryOutputArchive:
 *      quadrature/saveSparseGHWeightMap.h:14-51, helpers/SerializeEigenMaps.h:195-224; loaded at
 *      quadrature/SparseGaussHermite.h:80-117).  Host only, no GPU.
 * _list : number of entries; the (dim, deg, rows) of the first `cap` entries in file order (arrays may be NULL).
 * _read : the entry with key (d, p) -> Z [N][d] row-major, w [N]; with Z = w = NULL only *N is returned.
 *         GVI_ERR_NOTABLE when the key is absent (the reference only prints, SparseGaussHermite.h:150-160).
 * _write: generate every (dims[e], degs[e]) with the in-tree nwspgr restatement and write them in that order
 *         (save_pointweightmaps, saveSparseGHWeightMap.h:14-51). */
gvi_status gvi_table_file_list(const char* path, int64_t cap, int64_t* count, double* dims, double* degs, int64_t* rows);
gvi_status gvi_table_file_read(const char* path, int d, int p, int64_t* N, double* Z, double* w);
gvi_status gvi_table_file_write(const char* path, int n_entries, const int32_t* dims, const int32_t* degs);

/* ---- problem definition ---- */
/* Chain of T states of size n (joint dimension T n): GVIGH(vec_factors, dim_state, num_states, ...)
 * gvibase/GVI-GH.h:41-64.  Drops previously added factor sets. */
gvi_status gvi_chain_set(gvi_ctx* ctx, int T, int n);
/* Add a homogeneous set of K factors (same d, GH degree p, psi kind): the batched counterpart of
 * constructing K NGDFactorizedBaseGH objects (ngd/NGDFactorizedBaseGH.h:37-44).  start[k] is the
 * reference's start_index (first state of factor k); d must be n or 2n.  temperature may be NULL
 * (= 1).  Generates the (d,p) table and uploads it unless a table for (d,p) is already resident. */
gvi_status gvi_factors_add(gvi_ctx* ctx, int K, int d, int p, const int32_t* start, int psi_kind,
                           const double* psi_params, int64_t params_per_factor,
                           const double* temperature, int* set_id);
/* gvi_factors_add with a caller-supplied quadrature table instead of the generated one: the counterpart of passing
 * weight_sigpts_map_option (a shared QuadratureWeightsMap) to the factor constructors
 * (ngd/NGDFactorizedBaseGH.h:41, ngd/NGDFactorizedLinearGH.h:36, gvibase/GVIFactorizedBaseGH.h:37) whose
 * SparseGaussHermite then looks (d, p) up in that map (quadrature/SparseGaussHermite.h:138-166).  Z [N][d], w [N],
 * host; nothing is generated (p is only recorded).  The table is private to the set. */
gvi_status gvi_factors_add_table(gvi_ctx* ctx, int K, int d, int p, const int32_t* start, int psi_kind,
                                 const double* psi_params, int64_t params_per_factor,
                                 const double* temperature, int64_t N, const double* Z, const double* w,
                                 int* set_id);
/* Replace the set's quadrature table by a caller-supplied one (e.g. read from the reference's
 * cereal file quadrature/SparseGHQuadratureWeights_cereal.bin): Z [N][d], w [N], host. */
gvi_status gvi_factors_set_table(gvi_ctx* ctx, int set_id, int64_t N, const double* Z, const double* w);
/* Signed-distance grid of a GVI_PSI_HINGE_SDF_2D set: PlanarSDF(origin, cell_size, data)
 * (helpers/CudaOperation.h:39-43); data is COLUMN-major rows x cols like Eigen's MatrixXd
 * (data[r + c * rows], :130), row = y cell, col = x cell; queries are clamped to the grid (:61-80). */
gvi_status gvi_factors_set_sdf2d(gvi_ctx* ctx, int set_id, double origin_x, double origin_y, double cell_size,
                                 int rows, int cols, const double* data);
/* 3-D field of a GVI_PSI_HINGE_SDF_3D set: SignedDistanceField(origin[3], cell_size, data) with
 * data[r + c * rows + z * rows * cols] (helpers/CudaOperation.h:148-158, 304-306): row = y, col = x, slice = z. */
gvi_status gvi_factors_set_sdf3d(gvi_ctx* ctx, int set_id, const double* origin, double cell_size, int rows, int cols,
                                 int nz, const double* data);
/* Arm model of a GVI_PSI_HINGE_SDF_3D_ARM set: ForwardKinematics(a, alpha, d, theta_bias, num_spheres, frames, centers)
 * + radii (helpers/CudaOperation.h:345-357, 688-716): DH parameters [ndof], sphere q on frame frames[q] (non-decreasing)
 * with centre centers[q][3] in that frame and radius radii[q]; nspheres >= the factor dimension (the reference reads
 * n_balls = theta.size() spheres, :753). */
gvi_status gvi_factors_set_arm(gvi_ctx* ctx, int set_id, int ndof, const double* a, const double* alpha, const double* d,
                               const double* theta_bias, int nspheres, const int32_t* frames, const double* centers,
                               const double* radii);
/* Closed-form route for a QUAD_PRIOR / FIXED_PRIOR set: NGDFactorizedLinear::calculate_partial_V and
 * fact_cost_value (ngd/NGDFactorizedLinear.h:93-129) instead of quadrature -- the factors the reference's
 * classify_factors sends to its linear branch (gvibase/GVI-GH-Cuda-impl.h:31-38).  No sigma points are
 * evaluated; the Gaussian 4th-moment contraction (its O(d^4) loop, :108-118) is done in the whitened
 * space where it collapses to (u0^2 + |h|^2) I + 2 h h^T per residual row.  on = 0 returns to quadrature. */
gvi_status gvi_factors_set_closed_form(gvi_ctx* ctx, int set_id, int on);
/* factor_switch_to_high_temperature (gvibase/GVIFactorizedBase.h:212-214), batched. */
gvi_status gvi_factors_set_temperature(gvi_ctx* ctx, int set_id, const double* temperature);
gvi_status gvi_factors_info(const gvi_ctx* ctx, int set_id, int* K, int* d, int* p, int64_t* N);

/* ---- per-pass factor operators (one call = all K factors of the set) ---- */
/* calculate_partial_V for every factor (ngd/NGDFactorizedBaseGH.h:53-74): symmetric sqrt of Sigma_k,
 * sigma-point expand, psi, the three GH integrals, Vdmu_k = Lam_k E[(x-mu)psi]/T_k and
 * Vddmu_k = sym(Lam_k E[(x-mu)(x-mu)^T psi] Lam_k - Lam_k E[psi])/T_k with Lam_k = Sigma_k^-1.
 * Ephi[K] receives E[psi] (untempered).  Any output may be NULL. */
gvi_status gvi_moments(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma,
                       double* Ephi, double* Vdmu, double* Vddmu);
gvi_status gvi_moments_dev(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma,
                           double* Ephi, double* Vdmu, double* Vddmu);
/* The raw integrals E_Phi / E_xMuPhi / E_xMuxMuTPhi (gvibase/GVIFactorizedBaseGH.h:54-64). */
gvi_status gvi_raw_moments(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma,
                           double* E_phi, double* E_xmuphi, double* E_xxphi);
/* fact_cost_value for every factor (ngd/NGDFactorizedBaseGH.h:122-129): cost[k] = E[psi]/T_k. */
gvi_status gvi_costs(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma, double* cost);
gvi_status gvi_costs_dev(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma, double* cost);
/* Generic-psi route (any kind): sigma points X [K][d][N] in the reference CUDA path's layout
 * (gvibase/GVI-GH-Cuda-impl.h:177-189: [factor][dim][point]); the host evaluates psi [K][N];
 * the device reduces.  SparseGaussHermite::update_sigmapoints / Integrate
 * (quadrature/SparseGaussHermite.h:197-243). */
gvi_status gvi_expand(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma, double* X);
gvi_status gvi_moments_from_psi(gvi_ctx* ctx, int set_id, const double* mu, const double* Sigma,
                                const double* psi, double* Ephi, double* Vdmu, double* Vddmu);

/* ---- joint (block-tridiagonal) operators ---- */
/* local2joint_dmu_insertion / local2joint_dprecision_insertion + the sum of
 * NGDGH::compute_gradients (ngd/NGDFactorizedBaseGH.h:91-106; ngd/NGD-GH-impl.h:39-55):
 * g, D, U are overwritten with the ordered (set after set, factor index ascending) sum.
 * Vdmu[s] / Vddmu[s] belong to set_ids[s]. */
gvi_status gvi_bt_assemble(gvi_ctx* ctx, int nsets, const int* set_ids, const double* const* Vdmu,
                           const double* const* Vddmu, double* g, double* D, double* U);
/* x = A^-1 rhs for the symmetric block-tridiagonal A = (D, U): replaces the ConjugateGradient of
 * ngd/NGD-GH-impl.h:59-60 by a direct block elimination (partial pivoting inside blocks). */
gvi_status gvi_bt_solve(gvi_ctx* ctx, const double* D, const double* U, const double* rhs, double* x);
/* sum(log pivots)/2 of the natural-order LDL^T (gvibase/GVI-GH-impl.h:192-196).  NaN when a pivot is
 * not positive, so the reference's "NaN < cost is false -> reject the step" rule survives. */
gvi_status gvi_bt_logdet(gvi_ctx* ctx, const double* D, const double* U, double* half_logdet);
/* Tridiagonal blocks of A^-1: inverse_GBP / EigenWrapper::inv_sparse
 * (gvibase/GVI-GH-GBP-impl.h:246-305; helpers/EigenWrapper.h:282-381). */
gvi_status gvi_bt_marginals(gvi_ctx* ctx, const double* D, const double* U, double* SigD, double* SigU);
/* update_mu_from_joint / extract_cov_from_joint for every factor of the set
 * (gvibase/GVIFactorizedBase.h:104-122): mu_k [K][d], Sigma_k [K][d][d]. */
gvi_status gvi_gather_marginals(gvi_ctx* ctx, int set_id, const double* mu, const double* SigD,
                                const double* SigU, double* mu_k, double* Sigma_k);

/* ---- device-resident NGD iteration: GVIGH::optimize body (gvibase/GVI-GH-impl.h:39-118) with
 *      NGDGH::compute_gradients / onestep_linesearch / update_proposal (ngd/NGD-GH-impl.h:21-63,
 *      130-156).  State (mu, Lambda, marginals, per-factor (mu_k, Sigma_k)) never leaves HBM. ---- */
gvi_status gvi_ngd_init(gvi_ctx* ctx, const double* mu, const double* D, const double* U); /* host */
/* cost_value() at the current proposal (gvibase/GVI-GH-impl.h:176-197). */
gvi_status gvi_ngd_cost(gvi_ctx* ctx, double* cost);
/* factor_cost_vector() for one set (gvibase/GVI-GH-impl.h:147-170), host buffer [K]. */
gvi_status gvi_ngd_factor_costs(gvi_ctx* ctx, int set_id, double* costs);
/* compute_gradients(): moments of every set, assemble, dprecision = Vddmu - Lambda, solve dmu. */
gvi_status gvi_ngd_gradients(gvi_ctx* ctx);
/* onestep_linesearch(step): trial = current + step * (dmu, dprecision); returns its cost. */
gvi_status gvi_ngd_trial(gvi_ctx* ctx, double step, double* new_cost);
/* update_proposal: the last trial becomes the current proposal (marginals are reused). */
gvi_status gvi_ngd_accept(gvi_ctx* ctx);
/* One iteration with the reference's backtracking rule: step = base; repeat step *= 0.75, accept the
 * first trial with cost < current cost, give up after max_backtrack + 1 trials.  cost_iter is the
 * cost at entry; accepted/new_cost/ntrials report the outcome. */
gvi_status gvi_ngd_step(gvi_ctx* ctx, double step_size_base, int max_backtrack, double* cost_iter,
                        int* accepted, double* new_cost, int* ntrials);
/* The iteration loop of GVIGH::optimize (gvibase/GVI-GH-impl.h:38-112) without its recorder and temperature schedule:
 * up to max_iters calls of gvi_ngd_step in one C call, so that no host-language overhead sits between the decision of
 * one iteration and the launches of the next.  Stops early after an iteration whose backtracking was exhausted
 * (accepted = 0): optimize() then switches the temperature or declares convergence (:96-108) -- the caller's move.
 * Outputs are per iteration, any may be NULL; *iters_done counts the iterations run.  Same numbers as the same sequence
 * of gvi_ngd_step calls.  With option "pipeline" = 1 (the default; GVI_PIPELINE=0 switches it off) the launches of iteration
 * i + 1 are queued -- predicated on a device-side accept word -- before the host has read the cost of iteration i (single
 * process, chain pattern on the sign-orbit kernel; otherwise the plain loop). */
gvi_status gvi_ngd_run(gvi_ctx* ctx, int max_iters, double step_size_base, int max_backtrack, double* cost_iter,
                       int* accepted, double* new_cost, int* ntrials, int* iters_done);
/* Scheduling of gvi_ngd_step (results are identical in every mode):
 *   speculate  1 (default): the next iteration's gradients are queued behind the first trial, so the
 *              device does not idle while the host reads the cost; 0: strictly trial-then-decide;
 *   fuse_trial 1: the first trial's cost is taken from a FULL moments pass at the trial point, which is
 *              also the next iteration's gradient pass (one psi pass per accepted iteration instead of
 *              the reference's cost pass + gradient pass); 0: separate passes as the reference;
 *              2 (default): adaptive -- fused while first trials keep being accepted, the cost-only pass for the
 *              step after a rejected first trial (a rejected fused trial wastes the moment accumulation). */
gvi_status gvi_ngd_set_mode(gvi_ctx* ctx, int speculate, int fuse_trial);
/* psi passes launched by the resident iteration since the last reset: full (all moments) and cost-only (m0);
 * one pass = every factor of every set once.  bench.py derives its evaluation counts from these. */
gvi_status gvi_ngd_counters(gvi_ctx* ctx, int64_t* full_passes, int64_t* cost_passes, int reset);

/* ---- proximal (JKO / Bures-Wasserstein) update rule: ProxGVIGH + ProxGVIFactorizedBaseGH
 *      (proxgd/ProxGVI-GH-impl.h:24-60, 121-202; proxgd/ProxGVIFactorizedBaseGH.h:64-113, 152-160).  Same quadrature
 *      moments as the natural-gradient path; per factor b = Lam E[(x-mu)psi], S = Lam E[(x-mu)(x-mu)^T psi] Lam - Lam E[psi],
 *      Sig_half = (I - h S) Sigma (I - h S)^T, Sigma_new = Sig_half/2 + h I + sqrtm(Sig_half (Sig_half + 4hI))/2,
 *      Vdmu = -b, Vddmu = (Sigma_new^-1 - Lam)/h; the joint dmu / dprecision are their plain scattered sums (no solve).
 *      The reference's prox classes never divide by the temperature: selecting the rule switches every set to unit
 *      temperature.  Single process only.
 * gvi_ngd_set_update_rule: GVI_RULE_NGD (default) or GVI_RULE_PROX_JKO; state set by gvi_ngd_init is kept.
 * gvi_prox_gradients(h): increments at the current proposal for step h (read back with gvi_ngd_get_gradients:
 *      dmu = gq, dprecision = (VD, VU)).   gvi_prox_trial(step): cost at mu + step dmu, Lam + step dprecision.
 * gvi_prox_step: one optimize() iteration -- gradients at h = base, trial B at base^B, first decreasing trial accepted,
 *      the last one accepted anyway after max_backtrack failures (decreased = 0 then). */
enum { GVI_RULE_NGD = 0, GVI_RULE_PROX_JKO = 1 };
gvi_status gvi_ngd_set_update_rule(gvi_ctx* ctx, int rule);
gvi_status gvi_prox_gradients(gvi_ctx* ctx, double h);
gvi_status gvi_prox_trial(gvi_ctx* ctx, double step, double* new_cost);
gvi_status gvi_prox_step(gvi_ctx* ctx, double step_size_base, int max_backtrack, double* cost_iter, int* decreased,
                         double* new_cost, int* ntrials);
/* Split forms for sharded factors (one process per GPU): *_local does the rank's factors and leaves
 * the partial sums in the exchange buffer; the caller all-reduces gvi_ngd_exchange() over the ranks
 * (RCCL); *_finish does the replicated chain work.  Single GPU: local; finish. */
gvi_status gvi_ngd_gradients_local(gvi_ctx* ctx);
gvi_status gvi_ngd_gradients_finish(gvi_ctx* ctx);
gvi_status gvi_ngd_cost_local(gvi_ctx* ctx);                  /* cost_value() at the current proposal, */
gvi_status gvi_ngd_cost_finish(gvi_ctx* ctx, double* cost);   /* split the same way (exchange 1)        */
gvi_status gvi_ngd_trial_local(gvi_ctx* ctx, double step);
gvi_status gvi_ngd_trial_finish(gvi_ctx* ctx, double* new_cost);
/* which = 0: packed [g | D | U] partial sums (count = T n + (2T-1) n^2) written by gradients_local;
 * which = 1: the trial's partial sum of factor costs (count = 1) written by trial_local;
 * which = 2: the packed partial sums written by spec_gradients_local (the other gradient buffer). */
gvi_status gvi_ngd_exchange(gvi_ctx* ctx, int which, void** dev_ptr, int64_t* count);

/* ---- sharded factors, exchange INSIDE the library (one process per GPU; no reference counterpart -- the
 *      reference is single-process OpenMP, ngd/NGD-GH-impl.h:31-52).  Every rank adds only ITS contiguous range of each
 *      factor set (gvi_factors_add with global start indices; an empty range is K = 0) and calls the ordinary
 *      gvi_ngd_init / gvi_ngd_cost / gvi_ngd_gradients / gvi_ngd_trial / gvi_ngd_step: after gvi_dist_init_* those run
 *      the two exchange steps of a pass on the context stream themselves -- the host issues ONE call per iteration:
 *        exchange 0  each rank's partial [g | D | U] is non-zero only on its state range [lo, hi]: the ranks ALL-GATHER
 *                    those state records (n + 2 n^2 doubles per state, padded to the longest range) and fold them in rank
 *                    order (the ranges of neighbours overlap in one state) -- (N-1)/N of an all-reduce's zeros never move;
 *        exchange 1  the partial factor-cost sums: all-gather of one double per rank, summed in rank order.
 *      Both are deterministic and give bit-identical results on every rank, which the replicated chain recursions need.
 * Transport: RCCL (dlopen of librccl, ncclAllGather on the context stream) or a host-supplied all-gather. */
/* 128-byte ncclUniqueId from rank 0, to be distributed by the host (any channel) before gvi_dist_init_rccl. */
gvi_status gvi_dist_unique_id(void* id128);
gvi_status gvi_dist_init_rccl(gvi_ctx* ctx, int rank, int world, const void* id128);
/* All-gather callback: send [count] doubles -> recv [world][count] doubles, both DEVICE pointers; the implementation
 * must order itself after the work already queued on hip_stream and either complete or queue its result on that stream
 * before returning.  Returns 0 on success. */
typedef int (*gvi_allgather_fn)(void* user, const void* send, void* recv, int64_t count, void* hip_stream);
gvi_status gvi_dist_init_callback(gvi_ctx* ctx, int rank, int world, gvi_allgather_fn fn, void* user);
/* rank / world of the context (1 process: 0 / 1) and the last exchange's geometry: records sent per rank. */
gvi_status gvi_dist_info(const gvi_ctx* ctx, int* rank, int* world, int* records_per_rank);
/* Speculative pipeline of the sharded driver (what gvi_ngd_step does inside one process): after trial_local and the
 * all-reduce of exchange 1, trial_publish queues the publish of the trial cost WITHOUT waiting; spec_gradients_local /
 * _finish (all-reduce exchange 2 in between) queue the next iteration's gradients at the trial state behind it;
 * trial_wait returns the cost; on acceptance accept_spec makes the trial current AND its gradients the current ones
 * (the next iteration starts without a gradient pass).  A rejected trial simply leaves the speculative buffer unused. */
gvi_status gvi_ngd_trial_publish(gvi_ctx* ctx);
gvi_status gvi_ngd_trial_wait(gvi_ctx* ctx, double* new_cost);
gvi_status gvi_ngd_spec_gradients_local(gvi_ctx* ctx);
gvi_status gvi_ngd_spec_gradients_finish(gvi_ctx* ctx);
gvi_status gvi_ngd_accept_spec(gvi_ctx* ctx);
/* Copy state to host; any pointer may be NULL.  mu[T][n], D, U, SigD[T][n][n], SigU[T-1][n][n]. */
gvi_status gvi_ngd_get_state(gvi_ctx* ctx, double* mu, double* D, double* U, double* SigD, double* SigU);
/* Last gradients: dmu[T][n], dD, dU (dprecision) and the assembled Vdmu g / Vddmu (VD, VU). */
gvi_status gvi_ngd_get_gradients(gvi_ctx* ctx, double* dmu, double* dD, double* dU, double* g,
                                 double* VD, double* VU);

/* ---- measurement hooks (bench.py): HIP-event time of the last moments / cost kernel launch of a
 *      set, in milliseconds, measured on the context stream; enable before the launches.
 *      on = 1: only the dominant launch (set 0, full moments pass) is bracketed -- an event pair costs
 *      ~14 us of queue gaps; on = 2: every moments / cost launch of every set; on = 3: like 1 but only every 8th
 *      dominant launch is bracketed (sampling keeps the measurement out of the measured iteration time). ---- */
gvi_status gvi_profile_enable(gvi_ctx* ctx, int on);
gvi_status gvi_profile_last(gvi_ctx* ctx, int set_id, int what /*0 moments kernel, 1 cost kernel*/, float* ms);
/* Stage timing of the resident iteration: while on, the launches of the three stages -- 0 the chain operations (trial
 * factorisation || gradient solve), 1 the factor pass (products, psi, epilogue), 2 the assemble -- are bracketed by event
 * pairs.  mean_us / counts (each [3], optional): mean device time per bracket and the number of brackets since the last
 * call; the records are then cleared and the mode set to `on`.  A pair costs ~14 us of queue gaps: keep it out of timed runs. */
gvi_status gvi_profile_stages(gvi_ctx* ctx, int on, float* mean_us, int* counts);
/* Launch geometry of the set's last moments/cost launch: variant (0 closed form, 1 generic, 2 register, 3 split = four waves per factor, d = 16/20/24,
 * 5 register kernel fused with the chain's other set in one launch, 6 sign-orbit kernel, 7 sign-orbit kernel for a non-polynomial psi), chunks. */
gvi_status gvi_profile_geometry(gvi_ctx* ctx, int set_id, int* variant, int* nchunk, int64_t* chunk);
/* Stress-test hook of the fence-free hand-over between the epilogue tail and the host (publish_to_host, option
 * "safe_publish"): entries > 0 (a power of two) starts recording the cost every tail publishes at log[(int)sequence & (entries - 1)]
 * on the DEVICE; entries == 0 with out != NULL copies that ring after a stream sync; entries < 0 stops recording.  *seq_now =
 * sequence number of the last publish issued. */
gvi_status gvi_debug_cost_log(gvi_ctx* ctx, int entries, double* out, double* seq_now);
/* Kernel variant override for A/B runs: 0 = auto (sum-of-squares sets with m = 6 / 12 on a table that decomposes into sign
 * orbits take the sign-orbit kernel; otherwise 5 / 2 / 1 as instantiated), 1 = generic LDS kernel, 2 = register kernel
 * (psi operands in LDS), 5 = register kernel with psi operands in SGPRs, 6 = sign-orbit kernel where supported, 7 = auto, with
 * the non-polynomial psi kinds (hinge-on-SDF, range) on the sign-orbit kernel also where a register kernel is instantiated
 * (an A/B leg and parity cross-check: for these kinds the lane-per-point kernels are faster).  3 and 4
 * (round-1 A/B variants, removed) return GVI_ERR_ARG. */
gvi_status gvi_set_variant(gvi_ctx* ctx, int variant);
/* ---- A/B switches: the complete list ----
 * Environment variables are read ONCE, at gvi_ctx_create (GVI_SPGH_EXTENDED at table generation, GVI_RCCL_PATH when
 * librccl is loaded); everything else is a run-time option of gvi_set_option.  Every switch below is exercised by a test
 * or by an A/B leg of tools/gpu_legs.sh; results are identical in every setting unless the last column says otherwise.
 *
 *   environment          option              default  what                                                   results
 *   GVI_ORBIT            orbit               1        sign-orbit psi kernel (0: lane-per-point kernels)      agree to 1e-10
 *   GVI_FUSED            fused               1        one-launch factor pass (0: prep -> psi -> epilogue)    bit-identical
 *   GVI_ASM_ON_LOAD      assemble_on_load    1        assemble inside the chain's first pass                 bit-identical
 *   GVI_PIPELINE         pipeline            1        gvi_ngd_run queues iteration i + 1 ahead of cost i     bit-identical
 *   GVI_CHAIN_WAVE       chain_wave          1        lane-per-node chain kernel for T <= 65, n <= 2         bit-identical
 *   GVI_CHAIN_MERGE      chain_merge         1        top pass + first backward pass of the chain in one     bit-identical
 *                                                     launch (0: one launch per pass)
 *   GVI_FUSE_TRIAL       (gvi_ngd_set_mode)  2        0 reference pass order, 1 fused trial, 2 adaptive      identical iterates
 *   GVI_SIDE_SOLVE       side_solve          1        gradient solve beside the trial factorisation          bit-identical
 *   GVI_NO_PAIR          pair_fuse           0 / 1    both chain sets in one psi launch (lane-per-point)     bit-identical
 *   GVI_NO_FUSE_GATHER   fuse_gather         0 / 1    gather inside the prep launch                          bit-identical
 *   GVI_NO_SCOST         no_scost            0        cost pass on the full kernel instead of the cost one   bit-identical
 *   GVI_SREG_PIPE        sreg_pipe           1        hand-pipelined loads of the lane-per-point kernel      bit-identical
 *   GVI_MIRROR           mirror              1        +- pairing of the lane-per-point kernel                agree to 1e-11
 *   GVI_SAFE_PUBLISH     safe_publish        0        checked publish + release / acquire arrival counters   bit-identical (slower)
 *   GVI_SPIN_MS          --                  2        wall-time bound of the host's spin on the publish word --
 *   GVI_SPGH_EXTENDED    --                  auto     long-double merge of the Smolyak weights (0 / 1 force)  weights
 *   GVI_RCCL_PATH        --                  --       the only librccl candidate gvi_dist_* tries            --
 *   --                   chol_sqrt           1        Cholesky factor for sum-of-squares psi                 agree to 1e-10
 *   --                   trust_table_degree  0        gvi_factors_add_table: the table is the rule of degree p  --
 *   --                   jacobi_tol_exp      -34      stopping threshold of the symmetric-root solve         rounding
 *   --                   split_flush, target_waves, orbit_waves, orbit_min_tiles, orbit_stack, orbit_copies, dual_chain,
 *                        warm_start: launch geometry / summation order of individual kernels (tests/test_gpu_parity.py A/Bs)
 * Build-time only: GVI_BUILD_DEFINES=GVI_FUSED_TIMING (phase stamps; GVI_FUSED_DBG=8 prints them), GVI_ORBIT_SPLIT12,
 * GVI_EXP_* (timing experiments with WRONG results, tools/build_variant.py).
 *
 * Runtime form of the A/B environment switches read at gvi_ctx_create.  Results are identical in every
 * setting of the scheduling switches; the switches that select another summation order or another (mathematically
 * equivalent) route agree to rounding: "split_flush" (d = 16 / 20 / 24 kernel; 0 = plain recursive sums), "mirror",
 * "orbit" / "orbit_waves" / "orbit_copies" (sign-orbit kernel, its chunking and its private accumulator copies),
 * "trust_table_degree" (1: a table passed to gvi_factors_add_table is the Smolyak rule of the degree p it is added under --
 * e.g. the generator's table, made once and handed to the other ranks -- and takes the routes of the generated table;
 * default 0: a caller's table keeps the symmetric square root, as the reference maps the nodes),
 * "safe_publish" (1: the trial cost reaches the host in the checked four-word form and the arrival counters of the epilogue
 * tail carry release / acquire order instead of the fence-free protocol; same numbers, slower: the A/B of DESIGN section 4.2),
 * "chol_sqrt" (1: sum-of-squares sets take S = chol(Sigma) instead of the symmetric root -- the quadrature is exact there),
 * "assemble_on_load" (1: on chain-structured graphs the ordered assemble of (g, V) is done by the first pass of the chain
 * operations that consume it instead of a launch of its own; same sums in the same order),
 * "fused" (1: the full moments pass of the resident iteration is ONE launch per pass -- gather, per-pass products, psi walk, chunk
 * sum, cost and back-transform of a factor in one workgroup; 0: the three launches prep -> psi -> epilogue),
 * "jacobi_tol_exp" (stopping threshold 10^value of the symmetric-root solve, on SQUARED off-diagonal / diagonal mass; values
 * above -20 return GVI_ERR_ARG -- the environment form GVI_JACOBI_TOL_EXP clamps to -20 instead).
 * chain_wave (default 1; process-wide, environment GVI_CHAIN_WAVE): chains of T <= 65 states of size n <= 2 run on the
 * lane-per-node kernel (one wave per chain operation) instead of the generic block-cyclic-reduction kernels.
 * chain_merge (default 1; environment GVI_CHAIN_MERGE): a chain of more than one pass runs its top pass and the backward
 * recursion of the last segmented pass in ONE launch (the backward workgroups wait for a device word the top pass's workgroup
 * releases); 0: one launch per pass.  The wait is bounded (1 s); on a time-out the waiting workgroups take NaN for what they
 * would have read, so the affected marginals / solution are NaN and the step is rejected -- value 2 is the test of that path
 * (the word is stored wrong on purpose).
 * Names: split_flush, sreg_pipe, mirror, pair_fuse, fuse_gather, side_solve, dual_chain, warm_start, no_scost, target_waves,
 * orbit, fused, assemble_on_load, orbit_waves, orbit_min_tiles, orbit_stack, orbit_copies, chol_sqrt, jacobi_tol_exp, pipeline, chain_wave, chain_merge,
 * trust_table_degree, safe_publish. */
gvi_status gvi_set_option(gvi_ctx* ctx, const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* GVI_HIP_H */
