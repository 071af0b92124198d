/*
 * geobipy_amd.h -- C ABI of libgeobipy_amd.so: the MI355X (gfx950) implementation of GeoBIPy's
 * per-sounding hot path (1-D layered-earth FDEM forward solve, its Jacobian, Gaussian data misfit
 * and log-likelihood), batched over soundings.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every function returns gbp_status, 0 = ok.
 *   - pointers marked [dev] are DEVICE pointers (HBM, fp64/int32, C-contiguous); pointers marked
 *     [host] are host pointers.  The library never allocates or frees caller memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Launches are
 *     stream-ordered; the library never synchronises behind the caller's back.
 *   - the acquisition system (filter tables, per-frequency constants) is uploaded once into an
 *     opaque handle; the caller keeps ownership of the arrays passed to gbp_fdem_system_create.
 *
 * Reference interfaces replaced (paths under /root/reference/geobipy/src/classes/):
 *   nbFdem1dfwd            forwardmodelling/Electromagnetic/FD/fdem1d_numba.py:24-68
 *   nbFdem1dsen            forwardmodelling/Electromagnetic/FD/fdem1d_numba.py:71-121
 *   fdem1dfwd / fdem1dsen  forwardmodelling/Electromagnetic/FD/fdem1d.py:10-52, 87-129
 *   DataPoint.std          data/datapoint/DataPoint.py:268-282
 *   EmDataPoint.active     data/datapoint/EmDataPoint.py:44-56
 *   DataPoint.data_misfit  data/datapoint/DataPoint.py:502-525
 *   DataPoint.likelihood   data/datapoint/DataPoint.py:491-500 (-> statistics/MvNormalDistribution.py:201-216)
 */
#ifndef GEOBIPY_AMD_H
#define GEOBIPY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int gbp_status;
enum {
    GBP_OK = 0,
    GBP_ERR_INVALID_ARG = -1,   /* NULL pointer, non-positive size, nF > GBP_MAX_FREQ ...            */
    GBP_ERR_UNSUPPORTED_TID = -2, /* tensor id outside {1, 3, 7, 9}: the reference leaves H undefined
                                     for those (fdem1d_numba.py:57-66); we refuse                     */
    GBP_ERR_BAD_SYSTEM = -3,    /* non-finite / non-positive frequency or separation                  */
    GBP_ERR_HIP = -4,           /* a HIP runtime call failed; see gbp_last_error()                    */
    GBP_ERR_NO_DEVICE = -5      /* no gfx950 device visible                                           */
};

#define GBP_MAX_FREQ 128  /* frequencies per system (TDEM: components x spline nodes) */
#define GBP_NC0 120       /* J0 filter length  (fdem1d_numba.py:18) */
#define GBP_NC1 140       /* J1 filter length  (fdem1d_numba.py:19) */

typedef struct gbp_fdem_system gbp_fdem_system;

/* Library / device info ------------------------------------------------------------------ */
const char *gbp_version(void);
const char *gbp_last_error(void);          /* thread-local text of the last failure            */
gbp_status gbp_device_count(int *count);   /* [host] out                                        */

/*
 * Acquisition system: the per-system arguments of nbFdem1dfwd (fdem1d_numba.py:25), i.e. exactly
 * what FD/fdem1d.py:31-49 extracts from an FdemSystem, all [host], length nF unless noted:
 *   tid          tensor_id = 1 + 3*rx_orient + tx_orient       (system/FdemSystem.py:199-203)
 *   frequencies  Hz
 *   tx_z, rx_z   vertical loop offsets: tHeight = altitude + tx_z, rHeight = -tHeight + rx_z
 *   tx_moment    `moments` argument;  scale = tx_moment * rx_moment
 *   rx_off       loop_offsets[0, :] (x separation), separation = |rx - tx|
 *   w0[120], lamda0[nF,120], w1[140], lamda1[nF,140]  filter weights / abscissae
 *                                                               (system/FdemSystem.py:67-101, 279-337)
 * The handle owns device copies of derived tables; destroy with gbp_fdem_system_destroy.
 */
gbp_status gbp_fdem_system_create(int nF, const int32_t *tid, const double *frequencies,
                                  const double *tx_z, const double *rx_z, const double *tx_moment,
                                  const double *scale, const double *rx_off, const double *separation,
                                  const double *w0, const double *lamda0, const double *w1,
                                  const double *lamda1, gbp_fdem_system **out);
/*
 * Generic Hankel-kernel system from caller-built tables (used by the TDEM path, whose frequency-domain
 * stage is the same layered-earth recursion at quasi-static spline-node frequencies; replaces the
 * frequency-domain part of gatdaem1d's forwardmodel, call site TD/tdem1d.py:89-96).  All [host]:
 *   npts[nF]      abscissa points of each frequency (>= 64)
 *   wmu[nF]       omega * mu0;  hd0[nF]: exponent height offset, hDiff = hd0 - 2 * altitude
 *   g[2 nF]       complex output scale per frequency (re, im)
 *   tables        7 arrays of P = sum(npts) doubles: a = lambda^2 | u0.re | u0.im | coef.re | coef.im |
 *                 ue.re | ue.im   (term = rTE * exp(ue * hDiff) * coef, summed per frequency)
 * The resulting handle is used with gbp_fdem_forward: pred[b] = [Re(out_f), Im(out_f)].
 */
gbp_status gbp_hankel_system_create_raw(int nF, const int32_t *npts, const double *wmu, const double *hd0,
                                        const double *g, const double *tables, gbp_fdem_system **out);
/*
 * Same system with an accuracy-budgeted abscissa window (opt-in): abscissae whose contribution to ANY output
 * is provably (|rTE| <= 1) below eps_ppm in total, for every sounding at altitude >= min_altitude, are left
 * out of the tables (typically half of the 120 at eps_ppm = 1e-12).  eps_ppm <= 0 = all abscissae, i.e.
 * gbp_fdem_system_create.  The Jacobian kernels use the same window (see gbp_fdem_system_create_binned for the bound).
 */
gbp_status gbp_fdem_system_create_windowed(int nF, const int32_t *tid, const double *frequencies,
                                           const double *tx_z, const double *rx_z, const double *tx_moment,
                                           const double *scale, const double *rx_off, const double *separation,
                                           const double *w0, const double *lamda0, const double *w1,
                                           const double *lamda1, double eps_ppm, double min_altitude,
                                           gbp_fdem_system **out);
/*
 * The same window chosen PER SOUNDING: n_bins table sets, set i windowed for altitudes >= first_altitude_m + i metres; the
 * forward kernels look up a sounding's set from its own altitude (below first_altitude_m: all abscissae; above the last bin:
 * the last set, whose bound still holds).  What is evaluated for a sounding therefore depends on that sounding alone --
 * results do not change with the batch a sounding is evaluated in -- and every output stays within eps_ppm of the full
 * 120 / 140-point sums (geobipy_amd's default 1e-10 ppm: two orders below the rounding error of those sums themselves, three
 * below the parity bar).  This is the handle geobipy_amd.FdemBatch and DeviceChains use by default.  The Jacobian kernels and
 * the sampler (gbp_rj_run*) use the same per-sounding sets: rTE is analytic in each layer's conductivity and bounded by 1 on the
 * right half plane, hence |d rTE / d ln sigma_k| <= 2/pi, and the terms dropped from a true-derivative entry obey the same bound.
 */
gbp_status gbp_fdem_system_create_binned(int nF, const int32_t *tid, const double *frequencies,
                                         const double *tx_z, const double *rx_z, const double *tx_moment,
                                         const double *scale, const double *rx_off, const double *separation,
                                         const double *w0, const double *lamda0, const double *w1,
                                         const double *lamda1, double eps_ppm, int first_altitude_m, int n_bins,
                                         gbp_fdem_system **out);
/*
 * Per-sounding windows for ANY handle (gbp_fdem_system_create, gbp_hankel_system_create_raw): builds the n_bins table sets
 * from the handle's own tables, replacing an earlier set.  relative = 0: eps in the units of the output (ppm for gbp_fdem
 * systems; gbp_fdem_system_create_binned = create + this).  relative = 1: eps as a fraction of the value a frequency's sum takes
 * for rTE = 1 at the bin's altitude (the image-source field, the largest the output gets) -- for raw tables whose outputs are not
 * ppm: the time-domain nodal spectra (geobipy_amd.TdemBatch and gbp_tdem_forward use 1e-12).  _clear_bins: back to all abscissae.
 */
gbp_status gbp_hankel_system_add_bins(gbp_fdem_system *sys, double eps, int relative, int first_altitude_m, int n_bins);
gbp_status gbp_hankel_system_clear_bins(gbp_fdem_system *sys);
/*
 * Further table sets for one handle -- the same frequencies, weights and point counts as the tables it was created with, other
 * altitude terms hd0[nF] and points tables[7][P]: e.g. the Hankel tables of other transmitter-receiver offsets of a time-domain
 * system.  The `set_of_row` ARGUMENT of gbp_fdem_forward_rows_ex / gbp_fdem_fm_dlogc_rows_ex (and gbp_td_operator.table_set for
 * the sampler) then makes row b of that launch use set set_of_row[b] (0 = the handle's own tables, k = the k-th added set; [dev]
 * int32, B entries, owned by the caller, read by the kernels; NULL = set 0 for every row), so soundings of different geometry
 * run in ONE launch.  The handle keeps no per-call state: host threads may share one handle, each with its own rows (SURVEY 8b
 * "Threading"; tests/c_abi/two_threads.cpp).  Call gbp_hankel_system_add_bins after the last add_set (eps = 0, n_bins = 0 for "no
 * windows"): it builds every set's descriptors.  Set ids are not range-checked.
 */
gbp_status gbp_hankel_system_add_set(gbp_fdem_system *sys, const double *hd0, const double *tables);
/* abscissa points a sounding at (integer) altitude_m is evaluated with */
gbp_status gbp_fdem_system_bin_points(const gbp_fdem_system *sys, int altitude_m, int *npts);
void gbp_fdem_system_destroy(gbp_fdem_system *sys);
gbp_status gbp_fdem_system_npoints(const gbp_fdem_system *sys, int *npts);  /* abscissa points evaluated per sounding */
gbp_status gbp_fdem_system_nfreq(const gbp_fdem_system *sys, int *nF);
/* free-space field H0 per frequency as (re, im) pairs, [host] out[2*nF] (fdem1d_numba.py:68 denominator) */
gbp_status gbp_fdem_system_h0(const gbp_fdem_system *sys, double *out);

/*
 * Batched forward solve.  Replaces B calls of FdemDataPoint.forward -> fdem1dfwd -> nbFdem1dfwd.
 *   nlayers [dev] int32[B]        layers per sounding (1 <= nlayers[b] <= Lmax); nlayers[b] == 0 skips sounding b:
 *                                 none of its outputs are written (forward, fused and Jacobian entries alike)
 *   sigma   [dev] f64[B, Lmax]    conductivity S/m          (Model.values)
 *   thk     [dev] f64[B, Lmax]    layer thickness m; entry nlayers[b]-1 (the half-space, inf in the
 *                                 reference's mesh.widths) is never read
 *   height  [dev] f64[B]          sensor altitude above the top of the model (DataPoint.z)
 *   pred    [dev] f64[B, 2*nF]    out: [Re(out_0..F-1), Im(out_0..F-1)] in ppm
 *                                 (data/datapoint/FdemDataPoint.py:544-545)
 */
gbp_status gbp_fdem_forward(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                            const double *sigma, const double *thk, const double *height,
                            double *pred, void *stream);

/* Same with an explicit number of waves per workgroup (1..16; 0 = chosen from the batch size, as gbp_fdem_forward does).
 * The partial Hankel sums of a frequency are combined in a fixed order PER wave count, so results are bit-reproducible
 * for a given `waves` and differ in the last digits between wave counts; callers that need results independent of the
 * batch size (sharded surveys) pass the same `waves` everywhere.  There is no hidden per-thread or environment state. */
gbp_status gbp_fdem_forward_ex(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                               const double *sigma, const double *thk, const double *height,
                               double *pred, int waves, void *stream);

/* Same with a table set per row (gbp_hankel_system_add_set): set_of_row [dev] int32[B] or NULL. */
gbp_status gbp_fdem_forward_rows_ex(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                                    const double *sigma, const double *thk, const double *height,
                                    double *pred, const int32_t *set_of_row, int waves, void *stream);

/*
 * Per-sounding status word (SURVEY 8b "Errors"; the reference asserts on the host, FD/fdem1d.py:29,
 * DP/FdemDataPoint.py:541, and lets NaNs run): status [dev] int32[B] out, OR of the GBP_ROW_* bits.  `pred` [dev]
 * f64[B, N] may be NULL (inputs only).  The compute entries never abort a batch for one bad row: a row with
 * nlayers[b] > Lmax (or > max_layers of the Jacobian entries) gets NaN outputs and nothing outside its own rows is
 * touched; nlayers[b] <= 0 rows are skipped; non-positive sigma / thickness propagate to NaN / inf in that row only.
 */
enum {
    GBP_ROW_BAD_NLAYERS = 1,       /* nlayers[b] < 1 or > Lmax                                   */
    GBP_ROW_BAD_SIGMA = 2,         /* a used conductivity is not finite and > 0                  */
    GBP_ROW_BAD_THICKNESS = 4,     /* a used thickness (all but the last layer) is not finite and > 0 */
    GBP_ROW_BAD_HEIGHT = 8,        /* altitude negative or not finite (FD/fdem1d.py:29)          */
    GBP_ROW_NONFINITE_OUTPUT = 16  /* a NaN / inf among the sounding's N predicted values        */
};
gbp_status gbp_fdem_validate(int B, int Lmax, int N, const int32_t *nlayers, const double *sigma, const double *thk,
                             const double *height, const double *pred, int32_t *status, void *stream);

/*
 * Gaussian data misfit + log-likelihood for B soundings of N channels each.
 *   pred, obs [dev] f64[B, N];  rel, add [dev] f64[B] (one relative / additive error per sounding,
 *   DataPoint.py:274);  chi2, logL [dev] f64[B] out.
 *   std_i = sqrt((rel*obs_i)^2 + add^2); channel i active iff obs_i > 0 and not NaN;
 *   chi2 = sum_active ((pred-obs)/std)^2;  logL = -(Na/2) ln 2pi - sum_active ln std - chi2/2.
 */
gbp_status gbp_gauss_loglike(int B, int N, const double *pred, const double *obs, const double *rel,
                             const double *add, double *chi2, double *logL, void *stream);

/* Same with an explicit standard deviation per channel, sd [dev] f64[B, N] (TdemDataPoint.std,
 * data/datapoint/TdemDataPoint.py:329-376, is time-gate dependent). */
gbp_status gbp_gauss_loglike_std(int B, int N, const double *pred, const double *obs, const double *sd,
                                 double *chi2, double *logL, void *stream);

/*
 * Fused forward + misfit + log-likelihood (one launch; what Inference1D.accept_reject evaluates at
 * every proposal, inversion/Inference1D.py:572-597).  pred may be NULL when only chi2 / logL are wanted.
 */
gbp_status gbp_fdem_forward_loglike(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                                    const double *sigma, const double *thk, const double *height,
                                    const double *obs, const double *rel, const double *add,
                                    double *pred, double *chi2, double *logL, void *stream);

gbp_status gbp_fdem_forward_loglike_ex(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                                       const double *sigma, const double *thk, const double *height,
                                       const double *obs, const double *rel, const double *add,
                                       double *pred, double *chi2, double *logL, int waves, void *stream);

/*
 * Batched Jacobian d pred / d ln(sigma_k) (ppm).  Replaces FdemDataPoint.sensitivity -> fdem1dsen ->
 * nbFdem1dsen.   J [dev] f64[B, 2*nF, Lmax] out (columns >= nlayers[b] are set to 0);
 * rows [0,nF) real part, [nF,2nF) imaginary part (FdemDataPoint.py:553-557).
 */
gbp_status gbp_fdem_sensitivity(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                                const double *sigma, const double *thk, const double *height,
                                double *J, void *stream);

/*
 * Same with two knobs:
 *   max_layers  an upper bound of nlayers[] known to the caller (sizes the kernel's LDS working set;
 *               pass Lmax when unknown).  A sounding with nlayers > max_layers gets a NaN row (J and pred).
 *   exact       0: reproduce the reference's M1_1 formula bit-for-tolerance (default of
 *               gbp_fdem_sensitivity); 1: the true derivative of the forward recursion -- the reference's
 *               expression at fdem1d_numba.py:269-274 is not.
 */
gbp_status gbp_fdem_sensitivity_ex(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                                   const double *sigma, const double *thk, const double *height,
                                   double *J, int max_layers, int exact, void *stream);

/*
 * Prediction AND Jacobian of the same models from one pass: FdemDataPoint.fm_dlogc (DP/FdemDataPoint.py:547-551,
 * = forward + sensitivity).  Arguments as gbp_fdem_sensitivity_ex; pred [dev] f64[B, 2*nF] out (may be NULL).
 * Each frequency's Hankel sum is formed by one wave, so `pred` does not depend on the launch shape; it equals
 * gbp_fdem_forward's to rounding (different summation order).
 */
gbp_status gbp_fdem_fm_dlogc(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                             const double *sigma, const double *thk, const double *height,
                             double *pred, double *J, int max_layers, int exact, void *stream);

/* `waves` (0..16): waves per workgroup, a performance hint only -- one wave owns a frequency, so neither J nor pred depend on it. */
gbp_status gbp_fdem_fm_dlogc_ex(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                                const double *sigma, const double *thk, const double *height,
                                double *pred, double *J, int max_layers, int exact, int waves, void *stream);

/* Same with a table set per row (gbp_hankel_system_add_set): set_of_row [dev] int32[B] or NULL. */
gbp_status gbp_fdem_fm_dlogc_rows_ex(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                                     const double *sigma, const double *thk, const double *height,
                                     double *pred, double *J, int max_layers, int exact, const int32_t *set_of_row,
                                     int waves, void *stream);

/* The two row entries with one more per-row argument (round 4): row_scale [dev] f64[B] or NULL.  Row b is evaluated with the points of
 * its table set taken to the horizontal distance rho_set / row_scale[b] -- abscissae x s, coefficients x s^3 (csrc/gbp_fdem_point.h
 * scale_point): exact for raw Hankel handles of DIPOLE sources (time-domain systems without ModellingLoopRadius), whose tables depend on
 * the distance through lam = base / rho only.  How the time-domain sampler evaluates a SAMPLED receiver position (gbp_td_moves) with the
 * chain's own table set instead of building tables per proposal; a changed dz or transmitter height enters through `height`
 * (2 h + dz = 2 (h + (dz - dz_set) / 2) + dz_set).  NULL: the `_ex` entries. */
gbp_status gbp_fdem_forward_rows_scaled(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                                        const double *sigma, const double *thk, const double *height, double *pred,
                                        const int32_t *set_of_row, const double *row_scale, int waves, void *stream);
gbp_status gbp_fdem_fm_dlogc_rows_scaled(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                                         const double *sigma, const double *thk, const double *height,
                                         double *pred, double *J, int max_layers, int exact, const int32_t *set_of_row,
                                         const double *row_scale, int waves, void *stream);

/* NOT part of the product interface -- timing helper for bench.py: average kernel time (ms) of `reps` launches of the
 * fused kernel, measured with hipEvents recorded on `stream` around the launches. */
gbp_status gbp_bench_time_forward_loglike(const gbp_fdem_system *sys, int B, int Lmax, const int32_t *nlayers,
                                         const double *sigma, const double *thk, const double *height,
                                         const double *obs, const double *rel, const double *add,
                                         double *pred, double *chi2, double *logL, void *stream,
                                         int reps, float *avg_ms);

/* Test hook: element-wise evaluation of the device math kernels (geobipy_amd/csrc/gbp_math.h) on
 * [dev] arrays of length n.  op: 0 exp_neg(x), 1 sincos(x) -> (sin, cos), 2 csqrt(x + i y) -> (re, im),
 * 3 rcp(x), 4 raw v_rsq_f64 seed, 5 raw v_rcp_f64 seed, 6 sqrt_rsqrt(x) -> (sqrt, 1/sqrt), 7 log_pos(x) (the sampler's ln),
 * 8 sincos_quadrant(x) -> (sin, cos) (the sampler's Box-Muller angle).
 * y and out1 may be NULL where unused. */
gbp_status gbp_debug_math(int op, int n, const double *x, const double *y, double *out0, double *out1,
                          void *stream);

/* ------------------------------------------------------------------------------------------------
 * Device-resident rjMCMC step: one iteration of Inference1D.accept_reject
 * (inversion/Inference1D.py:537-631) for B soundings at once, FDEM data, Resolve-style option set
 * (layer conductivities + relative / additive error; no height move).  Pieces restated per chain:
 *   structural move      RectilinearMesh1D.perturb            mesh/RectilinearMesh1D.py:1018-1118
 *   value remapping      Model.perturb_structure / insert / delete  model/Model.py:316-366
 *   value proposal       Model.stochastic_newton_perturbation model/Model.py:368-419
 *   error proposals      DataPoint.perturb                    data/datapoint/DataPoint.py:531-573
 *   priors               Model.probability :533-575, DataPoint.probability :454-489
 *   jump proposal ratio  Model.proposal_probabilities         model/Model.py:577-659
 *   accept / bookkeeping Inference1D.accept_reject :597-631, update :705-790
 * Same Markov kernel as the reference; the random numbers come from a counter-based generator
 * (Philox4x32-10 keyed by `seed`, counter = chain, iteration, stream, draw), so a chain is a valid
 * draw of the same sampler, not the reference's own draw for a numpy seed (the host-side
 * geobipy_amd.Inference1D does that).  All arrays are [dev]; [B, K] arrays have row stride K.
 */
typedef struct gbp_rj_options {
    int32_t max_layers;          /* K = maximum_number_of_layers                                      */
    int32_t n_channels;          /* N = 2 * nF                                                         */
    int32_t solve_gradient, solve_value;  /* which model priors enter the probability (solve_gradient / solve_parameter) */
    int32_t solve_relative_error, solve_additive_error, exact_jacobian;
    int32_t n_depth_bins, n_value_bins;   /* posterior grids (interface histogram / hit-map)          */
    int32_t n_error_bins;        /* cells of the error-level histograms (log10 between the prior bounds; reference: 99) */
    int32_t schedule;            /* 0: the caller decides what is accumulated (`accumulate` argument);
                                    1: the reference's per-sounding schedule (Inference1D.update :713-737, infer :641-688):
                                       a chain burns in at the first iteration > burn_in_min_iterations with misfit <
                                       active channels -- its posteriors and best model start over there --, is done
                                       n_markov_chains iterations later, and has failed when it has not burned in after
                                       n_markov_chains iterations; done / failed chains keep their final state          */
    int32_t burn_in_min_iterations, n_markov_chains;
    int32_t forward_waves;       /* waves per sounding of the forward launches (and per chain of the persistent kernel):
                                    0 = chosen from the block size, > 0 = fixed.  A performance hint only: the Hankel sums
                                    are reduced per 64-point pass and added in pass order, so the chains are bit-identical
                                    for any value, block size and sharding of the survey                 */
    double min_edge, max_edge, min_width; /* min_edge already raised to min_width (RectilinearMesh1D.py:358-360) */
    double p_birth, p_death, p_perturb, p_none;
    double value_precision;      /* 1 / ln(1 + factor)^2                                              */
    double value_min, value_max; /* parameter_limits: proposals with a conductivity outside have zero prior; value_max <= 0: none */
    double gradient_precision;   /* 1 / gradient_standard_deviation^2                                 */
    double alpha;                /* covariance_scaling                                                 */
    int32_t n_rel_groups, n_add_groups;   /* error levels per sounding, 1..4 each: one relative level per system x component and
                                             one additive level per system (DataPoint.py:268-282, TdemDataPoint.py:361-365);
                                             frequency-domain data with one system: 1 and 1                              */
    double rel_min[4], rel_max[4], rel_sd[4], add_min[4], add_max[4], add_sd[4];   /* per group; sd = sqrt(proposal variance) */
    double depth_bin_width;      /* interface histogram / hit-map depth cell                           */
    double value_half_width;     /* hit-map spans log10(sigma / prior mean) in [-w, w]                 */
    uint64_t seed;
    uint64_t first_chain;        /* global index of chain 0 of this block: the random streams are keyed by
                                    first_chain + b, so a survey gives the same chains however it is sharded */
    /* The height move of the data point (`solve_z`: Point.perturb / set_priors / set_proposals, pointcloud/Point.py:614-621,
     * 949-983): uniform prior height0 +- height_half_width (maximum_z_change), random walk height + height_scale * N(0, 1)
     * redrawn up to 10 times while outside the prior, then kept.  height_scale is the reference's `z_proposal_variance` AS IS:
     * its NormalDistribution.rng passes the variance to numpy as the scale (statistics/NormalDistribution.py:111).  Needs
     * chains->height_p and ->height0; chains->height is then STATE (written on acceptance).  Lock-step drivers only. */
    int32_t solve_height;
    double height_half_width, height_scale;
    int32_t additive_independent;/* the additive levels as Tempest_datapoint treats its additive-error MULTIPLIERS (data/datapoint/
                                    Tempest_datapoint.py:339-341, 475-487, 503-508): ONE joint draw per iteration from a log-normal
                                    centred on add_centre (where set_proposals put it: the initial multipliers -- the proposal's mean
                                    is never moved to the state, so this is an independence sampler, not a random walk), no redraw
                                    against the prior, and no prior term in the acceptance ratio.  0: the levels of every other data
                                    point (random walk, redrawn while outside the log-uniform prior, prior in the ratio)          */
    double add_centre[4];
    double extra_log_prior;      /* constant added to every proposal's log prior: the densities of the uniform priors of sampled
                                    scalars that live outside the chains struct, in gbp_td_moves; cancels in the acceptance ratio, keeps
                                    the stored prior / posterior values those of the full model                              */
    int32_t trace_every, trace_length; /* per-iteration traces (chains->trace_misfit / trace_accept; Inference1D.data_misfit_v / acceptance_v,
                                    inversion/Inference1D.py:408, 414, 713, 749): every `trace_every`-th entry of the reference's two
                                    arrays is kept, `trace_length` slots per chain -- slot j holds data_misfit_v[j * trace_every] (the misfit
                                    after update j * trace_every + 1) and acceptance_v[j * trace_every] (the decision of update j *
                                    trace_every), updates counted from the chain's (re)start.  trace_every = 1, trace_length = 2 *
                                    n_markov_chains: the reference's arrays in full.  0: no traces                                  */
} gbp_rj_options;

typedef struct gbp_rj_chains {
    int32_t B;
    /* constants */
    const int32_t *rel_group, *add_group;   /* [N] or NULL  group of each channel's relative / additive level; NULL = 0 */
    const double *add_scale;       /* [N] or NULL  per-channel factor of the additive error: std^2 = (rel d)^2 + (add * add_scale)^2
                                      (TdemDataPoint.std, data/datapoint/TdemDataPoint.py:361-365: sqrt(1e-3 / t)); NULL = 1 (FDEM) */
    const int64_t *chain_id;       /* [B] or NULL  global index of each chain (keys its random streams); NULL: first_chain + b */
    const double *data;            /* [B, N]  observed data (<= 0: inactive channel)                   */
    const double *height;          /* [B]     sensor height; with opt->solve_height the chain's CURRENT height: state, written by the
                                      sampler on acceptance (the const is dropped there)                 */
    const double *log_mean_prior;  /* [B]     ln of the best half-space conductivity                   */
    /* chain state */
    int32_t *k;                    /* [B]     layers                                                   */
    double *edges, *sigma;         /* [B, K]  interface depths (k - 1 used, +inf padded), conductivities (1 padded) */
    double *rel, *add;             /* [B, n_rel_groups], [B, n_add_groups]  error levels                                  */
    double *pred, *J;              /* [B, N], [B, N, K]  carried prediction / Jacobian                  */
    double *prior, *like, *misfit; /* [B]                                                              */
    /* proposal scratch (written by the step) */
    int32_t *action, *k_r;         /* [B]     0 none, 1 insert, 2 delete, 3 perturb; layers after the move */
    int32_t *nl_a, *nl_c;          /* [3, B]  k_r of the chains needing the phase A / C kernels, else 0: row 0 all,
                                      rows 1-2 split by layer count (<= 8, more)                         */
    int32_t *nl_b;                 /* [B]     k_r of the chains whose proposal keeps its dimension (fused forward), else 0 */
    double *edges_r, *sigma_r, *thk_r;        /* [B, K] remapped model                                 */
    double *rel_p, *add_p;                    /* [B, n_*_groups]  proposed error levels              */
    double *pred_r, *J_r;                     /* [B, N], [B, N, K] at the remapped model               */
    double *chol;                             /* [B, K, K] lower Cholesky factor of the proposal precision */
    double *log_prop, *sigma_p;               /* [B, K] proposed ln sigma, sigma                       */
    double *pred_p, *misfit_p, *like_p, *J_p; /* proposal: [B, N], [B], [B], [B, N, K]                  */
    double *log_ratio;                        /* [B]   ln acceptance ratio of the last step            */
    /* bookkeeping */
    int64_t *n_accepted;           /* [B]                                                              */
    int32_t *k_hist;               /* [B, K + 1]            posterior of the layer count               */
    int32_t *edge_hist;            /* [B, n_depth_bins]     interfaces with a conductivity contrast > 50 % */
    int32_t *rel_hist, *add_hist;  /* [B, n_*_groups, n_error_bins]  posteriors of the error levels (both or neither; may be NULL) */
    int32_t *hitmap;               /* [B, n_value_bins, n_depth_bins] or NULL (depth fastest: the cells of one layer share a
                                      value bin, so one iteration updates a few contiguous runs)         */
    int32_t *hit_dwell;            /* [B] (with hitmap)  iterations the current model is still owed to the hit map: a model is
                                      added with its dwell time when it is replaced; gbp_rj_flush_posteriors settles the rest */
    int32_t *burned_in_iteration;  /* [B]  schedule 1: -1 until the chain burns in (may be NULL for schedule 0)          */
    int32_t *status;               /* [B]  schedule 1: 0 running, 1 done, 2 failed to burn in                            */
    double *best_posterior;        /* [B]                                                              */
    int32_t *best_k;
    double *best_edges, *best_sigma;          /* [B, K]                                                */
    double *best_rel, *best_add;   /* [B, n_rel_groups], [B, n_add_groups] or NULL: error levels of the highest-posterior state */
    int32_t *iteration0;           /* [B] or NULL (= 0)  schedule 1: the iteration at which the chain (re)started -- the schedule
                                      counts from there (Inference1D.reset :984-999 restarts a chain that accepted nothing over a
                                      whole window; the host does the restart between calls, rjmcmc_gpu.DeviceChains.infer) */
    /* height move (opt->solve_height; all NULL otherwise) */
    double *height_p;              /* [B]  proposed height (scratch of the step)                                          */
    const double *height0;         /* [B]  centre of the uniform prior = the sounding's measured height                   */
    int32_t *height_hist;          /* [B, n_error_bins] or NULL  posterior of the height on the prior's cells (Point.set_z_posterior) */
    double *best_height;           /* [B] or NULL  height of the highest-posterior state                                  */
    int32_t *step_flags;           /* [B] or NULL  what the accept stage did with the chain in the last iteration: bit 0 accepted,
                                      bit 1 the highest-posterior state was replaced, bit 2 the posteriors were reset (burn-in),
                                      bit 3 the post-step state was added to the posteriors; 0 for a frozen chain.  Read by the
                                      stages that carry further per-chain state (gbp_td_moves)                             */
    double *trace_misfit;          /* [B, trace_length] or NULL  decimated misfit trace (opt->trace_every; NaN = not reached)      */
    uint8_t *trace_accept;         /* [B, trace_length] or NULL  decimated accept / reject decisions (both traces or neither)      */
    int32_t *best_iteration;       /* [B] or NULL  the update (1-based, from the chain's (re)start) that produced the highest-posterior
                                      state (Inference1D.best_iteration :733, 743)                                                */
} gbp_rj_chains;

/* The three host-logic stages of one iteration, exposed separately for the tests ... */
gbp_status gbp_rj_propose(const gbp_rj_options *opt, const gbp_rj_chains *c, int64_t iteration, void *stream);
gbp_status gbp_rj_newton(const gbp_rj_options *opt, const gbp_rj_chains *c, int64_t iteration, void *stream);
gbp_status gbp_rj_accept(const gbp_rj_options *opt, const gbp_rj_chains *c, int64_t iteration, int accumulate,
                         void *stream);
/* ... and n_iterations complete iterations (propose, prediction + Jacobian of the remapped models, newton, fused
 * forward + likelihood of the proposals that keep their dimension, prediction + Jacobian of those that change it, accept), all stream-ordered, no host
 * synchronisation.  `accumulate` != 0 adds every post-step state to the posterior histograms. */
gbp_status gbp_rj_run(const gbp_fdem_system *sys, const gbp_rj_options *opt, const gbp_rj_chains *c,
                      int64_t first_iteration, int n_iterations, int accumulate, void *stream);
/* `mode`: 0 = gbp_rj_run's choice, 1 = lock-step driver (five to seven stream-ordered launches per iteration over the whole block:
 * propose, the evaluations at the remapped models, Newton (1 or 2), the evaluations at the proposals, accept (1 or 2)),
 * 2 = persistent kernel (one workgroup owns a chain and loops over all n_iterations in ONE launch; frequency-domain data),
 * 3 = lock-step with one launch per kind of evaluation and layer-count bucket (ten per iteration; what time-domain blocks use),
 * 4 = lock-step as 1, the block cut into 3 contiguous sub-blocks that advance concurrently on streams of their own (a host thread
 * each for the duration of the call) -- the latency-bound per-chain stages of one overlap the physics of the others: +9 % at 2 048
 * chains, +20 % at 4 096, +25 % at 8 192, +12 % at 16 384.
 * All drivers walk bit-identical chains; small blocks (config 5 split over 8 GPUs: 1 024 chains per GPU) are about twice as
 * fast in mode 2, medium ones (2 048 ... ~20 000 chains) fastest in mode 4 (from about four iterations per call: the sub-blocks'
 * host threads and stream joins are per call), large ones in mode 1.  Mode 0 chooses.  In mode 4 the
 * launch masks nl_a / nl_c hold one [3, n] block per sub-block instead of one [3, B] array (scratch of an iteration). */
gbp_status gbp_rj_run_mode(const gbp_fdem_system *sys, const gbp_rj_options *opt, const gbp_rj_chains *c,
                           int64_t first_iteration, int n_iterations, int accumulate, int mode, void *stream);
/*
 * The same for time-domain data (TdemDataPoint): `sys` is the frequency-domain handle of the spline nodes of the system --
 * or of all systems of a multi-moment acquisition merged into one (gbp_hankel_system_create_raw) --, and a constant block
 * matrix turns the nodal spectrum into window values, windows = nodal @ W (geobipy_amd/tdem.py; replaces gatdaem1d's
 * forwardmodel / derivative, TD/tdem1d.py:89-154).  opt->n_channels = the number of windows; chains->add_scale carries the
 * time-dependent additive error, rel_group / add_group the level of each channel.
 */
/* Geometry mixing (transmitter / receiver attitude, azimuth of the receiver offset, X / Y / Z outputs; geobipy_amd/tdem_geometry.py):
 * the frequency-domain kernels then produce the nodal spectra of the BASIS INTEGRALS of the transmitter-receiver frame
 * (n_in values per row) and the spectrum entry m of the output components is the per-row real combination
 *     out[m] = sum_{t < terms} weights[b, col[m, t]] * in[src[m, t]]        (src < 0: no term)
 * formed inside gbp_td_apply_mix / the sampler's window stage.  n_in = 0: no mixing (the kernels' spectra are the outputs'). */
typedef struct gbp_td_mix {
    int32_t n_in;           /* nodal values per row written by the kernels = 2 * nF of `sys`; 0 = no mixing   */
    int32_t terms;          /* T                                                                              */
    int32_t n_weights;      /* weights per row                                                                */
    const int32_t *src;     /* [dev] int32[n_nodal, T]                                                        */
    const int32_t *col;     /* [dev] int32[n_nodal, T]                                                        */
    const double *weights;  /* [dev] f64[B, n_weights]                                                        */
    const double *offset;   /* [dev] f64[B, n_channels] or NULL: added to the windows of every row -- the predicted PRIMARY
                               field of data whose channels hold primary + secondary (Tempest_datapoint.py:106-123);
                               honoured with or without mixing (n_in = 0)                                      */
} gbp_td_mix;
/* Sampled attitude angles of the loop pair (the reference's solve_transmitter_pitch / _roll / _yaw and solve_receiver_pitch / _roll /
 * _yaw: Loop_pair.perturb system/Loop_pair.py:161-164, EmLoop.perturb / set_priors / set_proposals system/EmLoop.py:222-305 -- the
 * receiver pitch is the nuisance parameter of Tempest inversions).  A rotation changes neither the Hankel tables (they depend on the
 * horizontal distance and dz) nor the kernels' nodal spectra of the basis integrals, only the per-row real mixing weights (and the
 * primary field of total-field data): every chain carries its GA-AEM tuple, a proposal stage draws the angles -- uniform prior
 * centre +- half_width, random walk redrawn up to 10 times while outside, scale = the reference's proposal "variance" as is (see
 * solve_height) -- and forms the proposal's weights / offset; evaluations at proposals use those; the accept stage's decision
 * (chains->step_flags) moves them into the state.  Needs mix.n_in > 0 and chains->step_flags.  n_moves = 0: fixed geometry. */
typedef struct gbp_td_moves {
    int32_t n_moves;             /* 0 .. 6                                                                              */
    int32_t entry[6];            /* entry of the GA-AEM tuple a move samples: 1..3 transmitter roll, pitch, yaw; 7..9 receiver;
                                    with `rho_scale` (position moves): 0 transmitter height, 4..6 receiver offset dx, dy, dz           */
    double sign[6];              /* tuple entry = sign * sampled value (Loop_pair.Geometry negates pitch and yaw)        */
    double half_width[6];        /* maximum_<..>_change                                                                  */
    double scale[6];             /* <..>_proposal_variance, used as the standard deviation like the reference does       */
    int32_t n_bins[6];           /* cells of the posterior on the prior's support: 199 for a pitch, 99 otherwise (<= 199) */
    double *geom, *geom_p;       /* [dev] f64[B, 10] GA-AEM tuples: current (state) and proposed (scratch)                */
    const double *geom0;         /* [dev] f64[B, 10] the measured geometry: prior centres                                */
    double *weights;             /* [dev] f64[B, mix.n_weights] = mix.weights, writable (state)                          */
    double *weights_p;           /* [dev] f64[B, mix.n_weights] (scratch)                                                */
    double *offset, *offset_p;   /* [dev] f64[B, n_channels] = mix.offset writable / its proposal; both NULL without total-field data */
    int32_t *hist;               /* [dev] int32[B, n_moves, 199] or NULL: posteriors                                     */
    double *best_geom;           /* [dev] f64[B, 10] or NULL: tuple of the highest-posterior state                       */
    /* how weights and offset follow from a tuple (geobipy_amd/tdem_geometry.py GeometryMix): one block of n_basis weights per
     * (system, output component) */
    int32_t n_blocks, n_basis, loop, on_axis;
    int32_t basis[5];            /* the basis integrals of the layout, indices into (B0L, B1L, B0, B1, BA)               */
    const int32_t *block_comp;   /* [dev] int32[n_blocks]  0 x, 1 y, 2 z                                                  */
    const double *block_scale;   /* [dev] f64[n_blocks]    output sign * output scaling                                   */
    const double *block_primary; /* [dev] f64[n_blocks]    factor of the free-space field in output units (with offset)   */
    const int32_t *block_windows;/* [dev] int32[n_blocks]  windows of the block (with offset)                             */
    /* Position moves (the reference's solve_receiver_x / _y / _z = the pair's offset, solve_transmitter_z; Loop_pair.perturb :161-164,
     * Point.perturb): a moved receiver keeps its chain's TABLE SET -- built for the measured (rho_set, dz_set) -- and is evaluated with
     * the per-chain scalars  rho_scale = rho_set / rho  (gbp_fdem_*_rows_scaled; dipole sources only when dx / dy move) and the effective
     * height  h + (dz - dz_set) / 2  written to chains->height_p (proposal) / chains->height (state; the const is dropped).  All NULL:
     * angles only.  Needs chains->height_p. */
    double *rho_scale, *rho_scale_p;  /* [dev] f64[B] rho_set / rho of the state / of the proposal (scratch)              */
    const double *rho_set, *dz_set;  /* [dev] f64[B] horizontal distance and dz the chain's table set was built for       */
} gbp_td_moves;
typedef struct gbp_td_operator {
    int32_t n_nodal;        /* rows of W (= 2 * nF of `sys` without mixing)                                   */
    const double *W;        /* [dev] f64[n_nodal, n_channels], row-major                                      */
    double *nodal;          /* [dev] scratch f64[B, max(n_nodal, mix.n_in)]                                   */
    double *J_nodal;        /* [dev] scratch f64[B, max(n_nodal, mix.n_in), K]                                */
    gbp_td_mix mix;         /* geometry mixing, or n_in = 0                                                   */
    const int32_t *table_set;  /* [dev] int32[B] or NULL: table set of every chain (gbp_hankel_system_add_set)  */
    gbp_td_moves moves;     /* sampled attitude angles, or n_moves = 0                                        */
} gbp_td_operator;
/* The time-domain stage on its own (what TdemDataPoint.forward / sensitivity add to the frequency-domain kernels): for every
 * sounding with nlayers[b] > 0,  pred[b, :] = nodal[b, :] @ W  and, when J_nodal / J are given,
 * J[b, g, l] = sum_m J_nodal[b, m, l] W[m, g] for l < nlayers[b] (0 beyond).  All [dev]; nodal f64[B, n_nodal] as written by
 * gbp_fdem_forward on the raw Hankel handle, J_nodal f64[B, n_nodal, K] by gbp_fdem_sensitivity_ex, W f64[n_nodal, N],
 * pred f64[B, N], J f64[B, N, K]. */
gbp_status gbp_td_apply(int B, int K, int n_nodal, int N, const int32_t *nlayers, const double *W, const double *nodal,
                        const double *J_nodal, double *pred, double *J, void *stream);
/* Same with geometry mixing: nodal f64[B, mix->n_in], J_nodal f64[B, mix->n_in, K] (mix NULL or n_in = 0: as gbp_td_apply). */
gbp_status gbp_td_apply_mix(int B, int K, int n_nodal, int N, const int32_t *nlayers, const double *W, const double *nodal,
                            const double *J_nodal, double *pred, double *J, const gbp_td_mix *mix, void *stream);
gbp_status gbp_rj_run_td(const gbp_fdem_system *sys, const gbp_td_operator *td, const gbp_rj_options *opt,
                         const gbp_rj_chains *c, int64_t first_iteration, int n_iterations, int accumulate, void *stream);
/* Adds what the chains' current models are still owed to the hit maps (see hit_dwell); call before reading them. */
gbp_status gbp_rj_flush_posteriors(const gbp_rj_options *opt, const gbp_rj_chains *c, void *stream);
/* ------------------------------------------------------------------------------------------------
 * Time-domain systems at the C level (SURVEY 8b "TDEM boundary"): what the reference gets from GA-AEM's
 *   gatdaem1d.TDAEMSystem(stmfile)                     system/TdemSystem_GAAEM.py:8-35
 *   .forwardmodel(Geometry, Earth) -> SX / SZ          forwardmodelling/Electromagnetic/TD/tdem1d.py:89-96
 *   Geometry(tx_height, tx_roll, -tx_pitch, -tx_yaw, txrx_dx, txrx_dy, txrx_dz, rx_roll, -rx_pitch, -rx_yaw)   system/Loop_pair.py:70-77
 * gbp_tdem_system_create parses the TEXT of a .stm file ([host], NUL-terminated) and folds waveform, spline, low-pass
 * filters and windows into one matrix (geobipy_amd/csrc/gbp_tdem.h); w0[120] / w1[140]: the J0 / J1 Hankel filter weights
 * (the same [host] arrays gbp_fdem_system_create takes).
 * gbp_tdem_forward: geometry [host] f64[B, 10] = GA-AEM's tuple above, angles in degrees in GA-AEM's convention (x = flight
 * direction, y = left, z = up; roll "left side up", pitch "nose down", yaw "turn left" positive; body -> earth matrix
 * Rz(yaw) Ry(pitch) Rx(roll)) -- the reference's sign changes of pitch and yaw are the caller's, as they are in Loop_pair.py.
 * Any attitude and any receiver offset per row, all rows in one launch: the handle keeps one table set per (horizontal
 * distance, dz) it has seen (at most 4096 per layout: bin measured offsets, e.g. to 0.1 m) and a row's azimuth and attitude
 * enter as a per-row mixing matrix of the nodal spectra (gbp_td_mix; geobipy_amd/tdem_geometry.py).  The conventions are restated
 * from GA-AEM's published description and held against closed forms (tests/test_tdem_attitude.py), not against gatdaem1d:
 * parity unpinned for non-zero angles.  nlayers / sigma / thk [dev] as in gbp_fdem_forward; out [dev] f64[B, n_components *
 * n_windows], components x, y, z (those with a non-zero output scaling), in the reference's sign convention for
 * predicted_secondary_field (TdemDataPoint.py:1004-1015).
 * gbp_tdem_fm_dlogc: the same plus J [dev] f64[B, n_components * n_windows, Lmax] = d out / d ln sigma (columns >= nlayers[b]
 * are 0) -- what the reference gets from gatdaem1d's fm_dlogc / derivative(CONDUCTIVITYDERIVATIVE, layer) x sigma
 * (TD/tdem1d.py:98-154), here the exact derivative through the same kernels.
 * Stream-ordered and RE-ENTRANT (SURVEY 8b "Threading"; round 4): host threads may share one handle, each calling on a stream of its
 * own.  A call leases its staging vectors and device scratch from a pool in the handle (the workspace it used last on the same stream,
 * else one whose last call has completed, else a new one); the table sets the calls share are looked up -- and, for a geometry the
 * handle has not met, grown: the device tables are re-allocated then, which waits for the launches that still read the old ones --
 * under the handle's lock, together with the enqueue of the call's launches, so the GPU work of concurrent callers overlaps and only
 * their host-side enqueue takes turns.  Results do not depend on what else the handle has seen or is doing
 * (tests/c_abi/two_threads_tdem.cpp: bit-equal to a fresh handle's single-threaded results while another thread grows the table sets).
 * (The geometry rows and per-sounding heights are HOST arrays read during the call; gbp_tdem_system_set_hankel_eps takes the same lock.)
 */
typedef struct gbp_tdem_system gbp_tdem_system;
gbp_status gbp_tdem_system_create(const char *stm_text, const double *w0, const double *w1, gbp_tdem_system **out);
void gbp_tdem_system_destroy(gbp_tdem_system *sys);
gbp_status gbp_tdem_system_info(const gbp_tdem_system *sys, int *n_windows, int *n_components, int *n_nodes, double *loop_radius);
/* accuracy budget of the per-sounding abscissa windows of gbp_tdem_forward, relative to the inductive-limit value of every nodal
 * sum (default 1e-12: about half of the 120 / 140 abscissae at survey altitudes, far below anything the windows resolve); 0 = all */
gbp_status gbp_tdem_system_set_hankel_eps(gbp_tdem_system *sys, double eps);
/* [host] out: window centres [n_windows] (= gatdaem1d windows.centre), spline-node frequencies [n_nodes], the per-component
 * operator W [2 * n_nodes, n_windows]; any may be NULL */
gbp_status gbp_tdem_system_tables(const gbp_tdem_system *sys, double *window_centres, double *node_frequencies, double *W);
gbp_status gbp_tdem_forward(gbp_tdem_system *sys, int B, const double *geometry, int Lmax, const int32_t *nlayers,
                            const double *sigma, const double *thk, double *out, void *stream);
gbp_status gbp_tdem_fm_dlogc(gbp_tdem_system *sys, int B, const double *geometry, int Lmax, const int32_t *nlayers,
                             const double *sigma, const double *thk, double *out, double *J, void *stream);

/* Diagnostics: [host] out[8] = accumulated 100 MHz clock ticks of chain 0 in the persistent kernel's stages (propose, fm_dlogc at
 * the remapped model, newton, forward / fm_dlogc at the proposal, accept), out[5] = iterations counted, out[6] / out[7] = the longest /
 * the mean life of the launches' workgroups in ticks (a persistent launch ends with its slowest chain); synchronises the device.
 * reset: 0 read only, 1 zero the counters and arm the clock (off by default: it costs chain 0 a few global updates per iteration),
 * 2 zero and disarm. */
gbp_status gbp_rj_debug_stage_ticks(int64_t *out, int reset);
/* Test hook: the proposal stage through one of its three independent implementations of the same draws (0 = gbp_rj_propose's:
 * thread per chain, rows staged through LDS; 1 = unstaged thread per chain; 2 = one wave per chain); an explicit argument, no
 * environment variable or other hidden state. */
gbp_status gbp_rj_debug_propose_variant(const gbp_rj_options *opt, const gbp_rj_chains *c, int64_t iteration, int variant, void *stream);
/* Test hook: n uniforms and n standard normals of stream (chain, iteration, stream_id) as the kernels draw them. */
gbp_status gbp_rj_debug_random(uint64_t seed, int64_t chain, int64_t iteration, int stream_id, int n,
                               double *uniforms, double *normals, void *stream);

/* What leaves the device of a block's conductivity-depth hit maps (int32 [B, n_value, n_depth], depth fastest; csrc/gbp_hitmap.h; the
 * reference derives the same on the host from its Histogram2D posterior, classes/statistics/Histogram.py mean / percentile):
 * gbp_hitmap_statistics -- per depth cell the mean and the 5 / 50 / 95 % points of log10 conductivity [B, n_depth] (bin centres
 * ((v + 0.5) / n_value) 2 half_width - half_width + log_mean_prior[b] / ln 10; the q-point is the first bin whose cumulative share is >= q);
 * gbp_hitmap_runs -- the maps' rows (M = n_value * n_depth cells) in run-length form: first call with start == NULL writes the number
 * of runs of every row to counts[B]; the caller forms ptr[B + 1] (exclusive prefix) and calls again with ptr, start[ptr[B]], value[ptr[B]]
 * (run r of row b: value[ptr[b] + r] from cell start[ptr[b] + r] to the next run's start). */
gbp_status gbp_hitmap_statistics(int B, int n_value, int n_depth, const int32_t *hitmap, const double *log_mean_prior, double half_width,
                                 double *mean, double *p05, double *p50, double *p95, void *stream);
gbp_status gbp_hitmap_runs(int B, int64_t M, const int32_t *hitmap, int64_t *counts, const int64_t *ptr, int32_t *start, int32_t *value,
                           void *stream);

/* [host] Results containers (geobipy_amd/h5lite.py; no reference counterpart -- the reference stores its hit maps dense): the rows of
 * a conductivity-depth hit map held as runs (row r owns runs ptr[r] .. ptr[r + 1] - 1; run q holds value[q] from cell start[q] of the row
 * -- the first at 0 -- to the next run's start) -> one zlib stream of the row's dense int32 bytes per row, written from the runs
 * (csrc/gbp_hostpack.h): out[out_ptr[r] .. out_ptr[r + 1]).  Capacity: 16 bytes per run + cells_per_row / 32 + 64 per row is enough. */
gbp_status gbp_runs_to_zlib(int n_rows, int64_t cells_per_row, const int64_t *ptr, const int32_t *start, const int32_t *value,
                            uint8_t *out, int64_t out_capacity, int64_t *out_ptr);

#ifdef __cplusplus
}
#endif
#endif /* GEOBIPY_AMD_H */
