/*
 * magprop_amd.h — C ABI of the MI355X-native magnetar log-posterior hot path.
 *
 * This is the drop-in boundary for the ONE data-parallel path of sgibson91/magprop
 * that this project accelerates: for each of N walkers, lnprior + integrate the
 * (Mdisc, omega) ODEs over the 10 001-point log grid + luminosity light curve +
 * linear interpolation at the observed times + -0.5*chi^2.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   - lnprob(pars, x, y, yerr, fbad)            code/synthetic_datasets/mcmc_eqns.py:52-81
 *   - lnlike(pars, x, y, yerr)                  code/synthetic_datasets/mcmc_eqns.py:5-25
 *   - lnprior(pars)                             code/synthetic_datasets/mcmc_eqns.py:28-49
 *   - model_lum(pars, xdata=None, n, alpha, cs7, k, dipeff, propeff, f_beam)
 *                                               code/synthetic_datasets/funcs.py:146-236
 *   - lnprob(pars, data, GRBtype, custom_lims)  magnetar/mcmc_eqns.py:87-119
 *   - lnlike(pars, data, GRBtype)               magnetar/mcmc_eqns.py:6-37
 *   - lnprior(pars, custom_lims)                magnetar/mcmc_eqns.py:40-84
 *   - model_lc(pars, xdata, GRBtype, ...)       magnetar/funcs.py:105-220
 *   - init_conds / odes / ODEs                  magnetar/funcs.py:17-101,
 *                                               code/synthetic_datasets/funcs.py:51-142
 * The reference is pure Python; its "FFI" for this path is the emcee log_prob_fn
 * callable (code/synthetic_datasets/synth_mcmc.py:180-185).  The ctypes binding a
 * maintainer would add is shown in INTEGRATION.md and shipped in
 * magprop_amd/_capi.py.
 *
 * Conventions
 *   - plain C, no torch / C++ types in any signature; caller owns every buffer.
 *   - every function returning int returns MP_OK (0) or a negative MP_E* code and
 *     records a message retrievable with mp_last_error() (thread-local).
 *   - per-walker physics failures are NOT errors: lnprob = -inf and a status code
 *     (reference: the string 'flag', magnetar/funcs.py:153-154).
 *   - all floating point is IEEE fp64.
 *
 * Threads and streams
 *   - A handle (and the samplers created on it) may be used from several host threads: every entry point
 *     that takes one serialises on a mutex inside the handle for the duration of the call.  The host-buffer
 *     entry points (mp_lnprob_batch, mp_model_lc, mp_rhs_batch, mp_sampler_run ...) run on the handle's own
 *     stream and return when their results are in the caller's buffers.
 *   - mp_lnprob_batch_dev and the mp_sampler_halfstep_* / mp_sampler_step_* calls only enqueue work on the stream
 *     they are given.  Launches of one handle on different streams may overlap on the device: a launch writes nothing
 *     but its own outputs (since ABI 4 also for handles that hold light curves of more than 64 points, which until
 *     ABI 3 owned per-walker scratch rows and were ordered by the library).  A batch that mixes light curves of more
 *     than 64 points with short ones and exceeds two wavefronts per SIMD is launched longest light curves first
 *     (results do not depend on it); its index buffer comes from a ring of eight per handle, so of such launches only
 *     those eight apart are ordered (by an event, on the device).
 *   - Buffers passed to an asynchronous call must stay valid until the work has completed on that stream;
 *     replacing a dataset (mp_set_dataset) waits for the device first.
 *   - All mp_sampler_halfstep_* calls of one sampler must use one stream.
 */
#ifndef MAGPROP_AMD_H
#define MAGPROP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_ABI_VERSION 5

/* return codes */
#define MP_OK 0
#define MP_EINVAL (-1)  /* bad argument (NULL, size, ds_id, ndim ...)            */
#define MP_EHIP (-2)    /* a HIP runtime call failed; see mp_last_error()         */
#define MP_ERANGE (-3)  /* observed time outside the model grid (interp1d raises  */
                        /* ValueError there: magnetar/funcs.py:214-215)           */
#define MP_ENODEV (-4)  /* no usable gfx950 device                                */
#define MP_ESTATE (-5)  /* call order problem (dataset / prior not set)           */

/* per-walker status (reference: success / 'flag' string / non-finite lnlike) */
#define MP_STATUS_OK 0
#define MP_STATUS_FLAG 1      /* break-up limit reached: the reference's LSODA 'flag'   */
#define MP_STATUS_NONFINITE 2 /* state or chi^2 went non-finite                         */
#define MP_STATUS_PRIOR 3     /* outside the prior box, model never evaluated           */
#define MP_STATUS_BADDATASET 4 /* ds_id names no registered light curve (out of range or never set): lnprob = -inf.  The   */
                              /* host-buffer entry rejects such a batch with MP_EINVAL; the device-pointer entries cannot */
                              /* read the ids and report it per walker instead                                            */

/* limits */
#define MP_MAX_NDIM 9        /* 6 physics parameters + up to dipeff, propeff, f_beam */
#define MP_MAX_DATASETS 64

/*
 * The two physics variants of the reference are data, not code forks
 * (SURVEY.md section 2.1).  mp_cfg_synth()/mp_cfg_lib() fill the two presets.
 */
typedef struct mp_model_cfg {
    double inertia_factor;     /* I = f*M*R^2 : 0.8 (magnetar/funcs.py:12) | 0.35 (code/synthetic_datasets/funcs.py:17) */
    double rm_massflow_factor; /* Rm ~ (f*Mdisc/tvisc)^(-2/7): 1 (magnetar/funcs.py:64) | 3 (synth funcs.py:105)        */
    double n_ode;              /* propeller switch-on inside the ODE right-hand side                                    */
    double n_lum;              /* propeller switch-on in the luminosity stage                                           */
    double alpha;              /* sound-speed prescription                                                              */
    double cs7;                /* sound speed, 1e7 cm/s                                                                 */
    double k;                  /* light-cylinder capping fraction                                                       */
    double dipeff;             /* default dipole efficiency   (overridden per walker when ndim is 8 or 9)               */
    double propeff;            /* default propeller efficiency (overridden per walker when ndim is 8 or 9)              */
    double f_beam;             /* default beaming fraction     (overridden per walker when ndim is 7 or 9)              */
    double nacc_lum_threshold; /* luminosity-stage break-up test: 0.27 (synth funcs.py:206) | 0.0 (magnetar/funcs.py:193) */
    int32_t lprop_gm_term;     /* 1: Lprop includes -(GM/Rm)*eta2*Mdisc/tvisc (synth funcs.py:222-223); 0: lib          */
    int32_t max_stride;        /* grid intervals a step of the solver may span: 0 = MP_MAX_STRIDE_DEFAULT (8), or 1, 2, 4, 8. */
                               /* 1 = every grid interval is a step (the serial restatement oracle/mp_oracle.c mode 0)       */
    double sweep_tol;          /* relative change of omega at the step ends that ends the Newton sweeps of a tile of the   */
                               /* time-parallel solver; 0 = MP_SWEEP_TOL_DEFAULT.  What it buys and costs: DESIGN.md 3     */
    double stride_tol;         /* smoothness indicator h |4th difference of (f - lambda omega)| / omega above which a tile   */
                               /* stepping over 2, 4 or 8 grid intervals is cut back to single intervals; 0 =                */
                               /* MP_STRIDE_TOL_DEFAULT                                                                      */
    int32_t dipole_torque;     /* ABI 5.  0: Ndip = -mu^2 omega^3 / (6 c^3), every model of the reference's packages          */
                               /* (magnetar/funcs.py:78; "Piro & Ott" in code/figure_3.py:79).  1: the alternative torque law */
                               /* of code/figure_3.py:105-165 ("Bucciantini"): Ndip = -(2/3) (mu^2 omega^3 / c^3) (Rlc/Rm)^3  */
                               /* with Rm after the light-cylinder cap.  It changes the ODE's torque only, as in that script  */
                               /* (which integrates and plots radii and torques, no luminosity).  Served by the curve kernels */
                               /* (mp_model_lc, mp_rhs_batch, mp_lnprob_batch); the device-resident sampler refuses it        */
    int32_t reserved;          /* 0                                                                                           */
} mp_model_cfg;

/* Stride adaptivity of the solver: where the solution is smooth on the scale of the output grid the
 * order-5 formula steps over 2, 4 or 8 grid intervals at once and the states at the skipped grid points come from the
 * step's dense output (<= 4e-10 relative); tiles that contain a kink of the right-hand side (Alfven-radius cap, torque
 * arm) or a fast transient are cut there and continued finer.  Measured on the 6 256 golden prior-wide points: same
 * maximum deviation from the reference's tight-integrator values as with max_stride = 1 (5e-8), 13.6 instead of 40 tiles
 * per walker (9 near the truths). */
#define MP_MAX_STRIDE_DEFAULT 8
#define MP_STRIDE_TOL_DEFAULT 1.0e-7
/* Coarse tiles that START before this time (seconds; the spin-up transients of heavy discs around strongly magnetised
 * stars live here) are held to a tenth of stride_tol.  A physical time, not a grid index: the reference's "L" grid
 * reaches it at index 1 003, its "S" grid (t0 = 1 ms) at index 4 003. */
#define MP_EARLY_HOLD_SECONDS 4.0

/* The sweeps contract by 1e-2..1e-3 per pass, so a tile whose last correction was <= 1e-7 relative is converged to
 * <= 4.5e-9 relative in lnprob (measured over the golden clouds and the prior-wide scans, tools/tol_scan.py) — 10x below
 * the 6e-8 by which the scheme itself differs from the reference integrated at rtol = atol = 1e-12, and 3000x below the
 * reference's own LSODA noise (1.4e-5).  The sweeps of a tile also end when every lane's correction fell at least tenfold
 * from the previous sweep and the next one -- estimated linearly from that factor -- would be below MP_STOP_FACTOR (a hundredth) of the
 * tolerance: the verification pass of a fast-converging tile is not run (round 4; near the truths 4 of a walker's 25
 * sweeps), and what is left is <= 0.01 x sweep_tol per tile.  MP_SWEEP_TOL_STRICT (1e-11) is what the tests use when they
 * compare kernel variants with each other and with the serial restatement of the scheme. */
#define MP_SWEEP_TOL_DEFAULT 1.0e-7
#define MP_SWEEP_TOL_STRICT 1.0e-11
#define MP_STOP_FACTOR 0.01

typedef struct mp_handle mp_handle;

int mp_abi_version(void);
const char *mp_last_error(void);

void mp_cfg_synth(mp_model_cfg *cfg); /* code/synthetic_datasets/funcs.py:146-147 defaults */
void mp_cfg_lib(mp_model_cfg *cfg);   /* magnetar/funcs.py:105-106 defaults; ODE always n=1 (funcs.py:150-151) */

/*
 * Create an evaluator bound to HIP device `device` (-1: the calling thread's current
 * device).  `tgrid` (host pointer, n_grid >= 2 strictly increasing doubles) is the
 * output/integration grid: np.logspace(0,6,10001) or np.logspace(-3,6,10001)
 * (magnetar/funcs.py:132-137).  The grid is copied to the device.
 */
mp_handle *mp_create(const mp_model_cfg *cfg, const double *tgrid, int n_grid, int device);
int mp_destroy(mp_handle *h);

/*
 * The same evaluator over SEVERAL devices of the node in ONE process (SURVEY.md 8(b)'s `device_mask`; ABI 5): what a
 * plain, single-process emcee user of `EnsembleSampler(..., vectorize=True)` needs to use more than one GPU -- the
 * reference's counterpart is `Pool()` over walkers, code/synthetic_datasets/synth_mcmc.py:178-185.  devices[n_devices]
 * lists HIP device indices (a device may be listed more than once: its share of every batch grows accordingly).
 * mp_set_dataset / mp_set_prior reach every device; mp_lnprob_batch (host buffers) deals the rows out in contiguous blocks
 * of ceil(n / n_devices) -- the partitioning of SURVEY.md 8(e) -- enqueues every device's kernel before it waits for the
 * first, and returns one vector: no torch, no RCCL, no second process.  mp_model_lc / mp_rhs_batch run on the first
 * device.  The device-pointer entry (mp_lnprob_batch_dev) and the device-resident sampler belong to one device and
 * return MP_ESTATE on such a handle; walker-sharded SAMPLING across devices is magprop_amd/distributed.py (one process per
 * GPU, RCCL).  Not measured on several GPUs by the builder (one-GPU boxes): tested with the same device listed twice.
 */
mp_handle *mp_create_multi(const mp_model_cfg *cfg, const double *tgrid, int n_grid, const int *devices, int n_devices);
int mp_n_devices(const mp_handle *h);   /* 1 for mp_create's handles */

/*
 * Register (or replace) observed light curve `ds_id` (0 <= ds_id < MP_MAX_DATASETS):
 * x = times [s], y = luminosity [1e50 erg/s], yerr = 1-sigma errors; host pointers.
 * Returns MP_ERANGE if any x lies outside [tgrid[0], tgrid[n_grid-1]].
 * Registering a NEW slot appends to the device-resident arrays (only the new light curve is uploaded, nothing waits
 * for the device unless the arrays have to grow); REPLACING a slot waits for the device first.
 */
int mp_set_dataset(mp_handle *h, int ds_id, const double *x, const double *y, const double *yerr, int n_obs);

/*
 * Box prior (inclusive), lower/upper[ndim].  Bit i of log_mask set: sampler
 * coordinate i is log10 of the physical parameter and is un-logged before the
 * model is evaluated (code/synthetic_datasets/mcmc_eqns.py:16-17).  ndim = 0
 * disables the prior (lnlike only; pars then taken as given, un-logged per log_mask).
 */
int mp_set_prior(mp_handle *h, const double *lower, const double *upper, int ndim, uint32_t log_mask);

/*
 * Batched log-posterior, host buffers.  pars[n][ndim] row-major (ndim 6..9:
 * B, P, MdiscI, RdiscI, epsilon, delta [, f_beam | dipeff, propeff [, f_beam]],
 * magnetar/mcmc_eqns.py:22-34).  ds_id[n] selects the dataset per walker (NULL:
 * dataset 0).  lnprob_out[n] required; status_out[n] and ltot_out[n][n_grid]
 * (model light curve in 1e50 erg/s on the grid) optional (NULL).
 */
int mp_lnprob_batch(mp_handle *h, const double *pars, const int32_t *ds_id, int n, int ndim,
                    double *lnprob_out, int32_t *status_out, double *ltot_out);

/*
 * Same, every pointer a DEVICE pointer on the handle's device; the kernel is
 * enqueued on `stream` (a hipStream_t passed as void*; NULL is HIP's default
 * stream, mp_stream(h) the handle's own) and the call returns without
 * synchronising.  Rows of d_ltot whose walker did not finish (status != MP_STATUS_OK)
 * are filled with NaN by the kernel, as in the host-buffer form.
 */
int mp_lnprob_batch_dev(mp_handle *h, const double *d_pars, const int32_t *d_ds_id, int n, int ndim,
                        double *d_lnprob, int32_t *d_status, double *d_ltot, void *stream);

/*
 * Full model light curve for one parameter vector in PHYSICAL units (no prior,
 * no un-logging): out[4][n_grid] = tarr, Ltot, Lprop, Ldip (luminosities in
 * 1e50 erg/s) as returned by model_lc/model_lum with xdata=None
 * (magnetar/funcs.py:219-220).  traj (optional) receives [2][n_grid] = Mdisc, omega.
 */
int mp_model_lc(mp_handle *h, const double *pars, int ndim, double *out, double *traj, int32_t *status);

/*
 * The ODE right-hand side itself, batched: replaces calls of `odes(y, t, B, MdiscI, RdiscI, epsilon, delta, ...)`
 * (magnetar/funcs.py:33-101) / `ODEs(...)` (code/synthetic_datasets/funcs.py:75-142).  For point i:
 * pars[i][ndim] PHYSICAL parameters (B, P, MdiscI, RdiscI, epsilon, delta[, ...]; P is not used by the RHS),
 * t[i], y[i] = (Mdisc [g], omega [rad/s])  ->  dydt[i] = (dMdisc/dt, domega/dt).  lam (optional, [n]) receives
 * d(omega_dot)/d(omega), the Jacobian entry the time-parallel solver linearises with.  Host buffers; evaluated by
 * the same device functions as the log-posterior kernels (one point per lane).
 */
int mp_rhs_batch(mp_handle *h, const double *pars, int ndim, const double *t, const double *y, int n, double *dydt,
                 double *lam);

/*
 * Ensemble sampler: emcee's affine-invariant stretch move (Goodman & Weare 2010) with a random red/blue
 * split per step, as driven by code/synthetic_datasets/synth_mcmc.py:175-185
 * (em.EnsembleSampler(Nwalk, Npars, lnprob, ...).run_mcmc(pos, Nstep)).  Positions, log-posteriors,
 * acceptance counters and the chain stay resident on the device; every half-step is ONE kernel launch that
 * proposes, evaluates the log-posterior, accepts/rejects and stores the chain row (small ensembles: a whole step per
 * launch, mp_sampler_set_whole_step).
 *   n_walkers   walkers per ensemble (even, >= 2*ndim recommended as in emcee)
 *   n_ensembles independent ensembles advanced together (e.g. one per GRB dataset); ens_ds_id[e] is the
 *               dataset of ensemble e (NULL: dataset 0 for all)
 *   a           stretch scale (emcee default 2.0)
 *   target      0: the magnetar log-posterior of this handle; 1: isotropic unit Gaussian (tests of the move)
 * Random numbers: Philox4x32-10 keyed by (seed; step, half, walker) for partner / stretch factor / accept;
 * the split permutations are Fisher-Yates shuffles driven by the same generator on the host.  Same seed => same chain.
 */
typedef struct mp_sampler mp_sampler;
mp_sampler *mp_sampler_create(mp_handle *h, int n_walkers, int n_ensembles, int ndim, const int32_t *ens_ds_id,
                              uint64_t seed, double a, int target);
int mp_sampler_destroy(mp_sampler *s);
/* pos[n_ensembles*n_walkers][ndim] (host): sets the state and evaluates its log-posterior */
int mp_sampler_set_positions(mp_sampler *s, const double *pos);
/* n_steps full steps; chain[n_steps][n_total][ndim] and chain_lnprob[n_steps][n_total] (host, both or neither) */
int mp_sampler_run(mp_sampler *s, int n_steps, double *chain, double *chain_lnprob);
/* mp_sampler_run evaluates a whole step in one launch while 3/2 x the walkers (all ensembles) are at most 19/8 x the SIMDs of
 * the device (2 432 evaluations on an MI355X: they fit two wavefronts per SIMD, or nearly): the proposals of the first half and, for every walker of the second half, both proposals it can end up making
 * (its partner of the first half moved or not); a small kernel then takes the decisions in emcee's order.  Same chain, bit
 * for bit, as one launch per half-step (enable = 0; what larger ensembles get anyway).  Default: enabled. */
int mp_sampler_set_whole_step(mp_sampler *s, int enable);
/* any of the outputs may be NULL */
int mp_sampler_get_state(mp_sampler *s, double *pos, double *lnprob, int64_t *n_accepted, int64_t *steps_done);
/*
 * Proposals inside the prior whose model evaluation failed ('flag' / non-finite): what the reference's lnprob appends
 * to its `fbad` file (code/synthetic_datasets/mcmc_eqns.py:72-79).  The kernels collect them in a device-side window of
 * MP_BAD_WINDOW rows which the library drains into a host-side log (after every chunk of mp_sampler_run and in this
 * call), so the log holds every failed proposal unless more than MP_BAD_WINDOW of them arrive between two drains.
 * *n_bad = how many there were since the sampler was created (exact); *n_logged (optional) = rows in the log;
 * rows [first_row, first_row + max_rows) of the log are copied to pars[max_rows][ndim] (sampler coordinates).
 * Returns the number of rows copied (>= 0) or a negative MP_E* code.
 */
#define MP_BAD_WINDOW 65536
int mp_sampler_get_bad(mp_sampler *s, int64_t first_row, double *pars, int max_rows, int64_t *n_bad, int64_t *n_logged);

/*
 * Walker-sharded ensembles (one process per GPU; the reference's counterpart is emcee's pool.map over
 * walkers, code/synthetic_datasets/synth_mcmc.py:178-185).  Every process holds a sampler with the same seed and the
 * full ensemble state.  A half-step has n_slots = mp_sampler_n_slots() proposals (the active half of every ensemble);
 * the process that owns slots [slot_lo, slot_hi) runs
 *     mp_sampler_halfstep_shard(s, half, slot_lo, slot_hi, d_rows + slot_lo*R, stream)    R = mp_sampler_row_doubles()
 * which draws, evaluates and accept-tests exactly what the single-GPU launch would for those walkers (the random numbers
 * are keyed by (seed; step, half, walker)) and writes one outcome row per slot: proposal[ndim], its lnprob, accepted
 * (0/1), status.  After the rows of all processes have been gathered (ONE all-gather per half-step; RCCL through
 * torch.distributed in magprop_amd/distributed.py) every process commits them with
 *     mp_sampler_halfstep_apply(s, half, d_rows, d_chain_row, d_chain_lnp_row, stream)
 * (d_rows[n_slots][R]; the optional d_chain_row[n_total][ndim] / d_chain_lnp_row[n_total] receive this step's entries of
 * the walkers that moved; both NULL or both set).  half = 0 then 1; the apply of half 1 ends the step.  All pointers
 * are device pointers; nothing synchronises with the host.  The resulting chain is the one mp_sampler_run produces
 * (bit for bit while both run the same kernel variant).
 */
int mp_sampler_n_slots(const mp_sampler *s);
int mp_sampler_row_doubles(const mp_sampler *s);
int mp_sampler_halfstep_shard(mp_sampler *s, int half, int slot_lo, int slot_hi, double *d_rows, void *stream);
int mp_sampler_halfstep_apply(mp_sampler *s, int half, const double *d_rows, double *d_chain_row,
                              double *d_chain_lnp_row, void *stream);
/*
 * The same with a WHOLE step per launch and ONE all-gather per step (mp_sampler_set_whole_step describes the launch):
 * a step has mp_sampler_step_blocks() = 3 * n_slots evaluations; every process runs
 *     mp_sampler_step_shard(s, block_lo, block_hi, d_rows, stream)
 * on its share of them, which writes one row of R' = mp_sampler_step_row_doubles() doubles per block to d_rows[b - block_lo]
 * (proposal[ndim], its lnprob, status, (ndim - 1) ln z, ln u, the walker's lnprob before the move, its partner's slot);
 * after the rows of all processes have been gathered every process commits the step with
 *     mp_sampler_step_apply(s, d_rows, d_chain_row, d_chain_lnp_row, stream)
 * (d_rows[3 * n_slots][R'], row index = block index).  Two launches and one collective per step instead of four and two;
 * a third of the evaluations is discarded, so this pays while a rank's share of the blocks fits its device two wavefronts
 * per SIMD (magprop_amd/distributed.py decides).  Same chain as the half-step protocol and as mp_sampler_run.
 */
int mp_sampler_step_blocks(const mp_sampler *s);
int mp_sampler_step_row_doubles(const mp_sampler *s);
int mp_sampler_step_shard(mp_sampler *s, int block_lo, int block_hi, double *d_rows, void *stream);
int mp_sampler_step_apply(mp_sampler *s, const double *d_rows, double *d_chain_row, double *d_chain_lnp_row, void *stream);
/* device pointers of the resident state: pos[n_total][ndim], lnprob[n_total] (read-only for the caller) */
int mp_sampler_state_ptrs(mp_sampler *s, double **d_pos, double **d_lnprob);

/* wait for everything enqueued on the handle's own stream */
int mp_synchronize(mp_handle *h);

/* introspection used by the measurement harness */
int mp_device(const mp_handle *h);
void *mp_stream(const mp_handle *h); /* the handle's own hipStream_t */
int mp_n_grid(const mp_handle *h);
/* mean Newton sweeps per tile of the most recent host-buffer batch (diagnostic) */
double mp_last_mean_sweeps(const mp_handle *h);
/* mean number of tiles solved per walker (kept or redone) in that batch: 40 / 79 with max_stride = 1 (4 / 2 steps per lane) */
double mp_last_mean_tiles(const mp_handle *h);
/* total Newton sweeps of every walker of that batch (all tiles; 0 for walkers that never started); returns the count copied */
int mp_last_sweeps(const mp_handle *h, int32_t *out, int n);
/* Diagnostics of the solver: with mp_tile_log(h, 1) every later host-buffer batch records, per walker, one word per tile it
 * solved (the first MP_TILE_LOG of them): kind (0: 1/8-interval sub-steps, 1 .. 4: steps over 1, 2, 4, 8 grid intervals) |
 * sweeps << 4 | lanes kept << 16 | why << 24 (bits: 1 a branch of the right-hand side changed inside the tile; 2 / 4 / 8 /
 * 32 the smoothness indicator exceeded stride_tol / its 64th / 2048th / 65536th; 16 lanes had not converged when the
 * sweeps were stopped; 64 the tile was given up after its second or third sweep).  mp_last_tile_log copies walker i's
 * words of the most recent batch; returns how many. */
#define MP_TILE_LOG 96
int mp_tile_log(mp_handle *h, int enable);
int mp_last_tile_log(const mp_handle *h, int walker, int32_t *out, int n);
/* tiles solved (kept or redone) by every walker of that batch; returns the count copied */
int mp_last_tiles(const mp_handle *h, int32_t *out, int n);
double mp_sweep_tol(const mp_handle *h); /* the tolerance in force (cfg.sweep_tol or the default) */
/* The solver settings in force for this handle, out[i] for i < min(n, MP_POLICY_COUNT); returns how many were written.
 * The shipped library takes them from cfg and the constants above only; the developer build
 * (`make -C magprop_amd/csrc experiments`) also reads MAGPROP_AMD_* environment overrides and says so in
 * out[MP_POLICY_EXPERIMENTS] (bench.py copies the whole vector into its JSON line). */
enum {
    MP_POLICY_MAX_STRIDE = 0,        /* grid intervals a step may span                                         */
    MP_POLICY_STRIDE_TOL,            /* smoothness bound of coarse tiles                                       */
    MP_POLICY_SWEEP_TOL,             /* end of the Newton sweeps                                               */
    MP_POLICY_EARLY_HOLD_SECONDS,    /* MP_EARLY_HOLD_SECONDS                                                  */
    MP_POLICY_K4_TOL_FACTOR,         /* stride_tol of tiles over 8 intervals, relative to stride_tol           */
    MP_POLICY_COARSE_TOL_FACTOR,     /* sweep_tol of tiles over 2, 4, 8 intervals, relative to sweep_tol       */
    MP_POLICY_COARSE_MAX_SWEEPS,     /* sweeps after which a coarse tile keeps its converged lanes             */
    MP_POLICY_FINE_MAX_SWEEPS,       /* the same for tiles over single intervals                               */
    MP_POLICY_TROUBLE_LIMIT,         /* failed coarse attempts after which stride 8 is no longer tried         */
    MP_POLICY_STOP_FACTOR,           /* MP_STOP_FACTOR: estimated next correction / sweep_tol that ends a tile  */
    MP_POLICY_FORCED_STEPS_PER_LANE, /* 0 = by batch size                                                      */
    MP_POLICY_EXPERIMENTS,           /* 1: developer build that honours MAGPROP_AMD_* environment overrides, or a build */
                                     /*    whose compile-time policy constants (the five below) were overridden with -D  */
    MP_POLICY_LIGHT_TOL,             /* correction below which the next sweep keeps the Jacobian and the weights         */
    MP_POLICY_CUT_BY_RATIO,          /* lanes behind a cut over which a fast feature's excess picks the next stride     */
    MP_POLICY_ABORT_SKIP_RATIO,      /* excess over the bound beyond which a given-up coarse tile skips a stride         */
    MP_POLICY_LOGPRED_MIN_KIND,      /* tile kinds from this one on start from the log-space extrapolation               */
    MP_POLICY_PRE_EARLY_END_FACTOR,  /* 65 536 x 100: the calm-first-tile test of the 128-step kernels' sub-stepped start */
    MP_POLICY_COUNT
};
int mp_get_policy(const mp_handle *h, double *out, int n);
int mp_n_simd(const mp_handle *h);       /* SIMDs of the handle's device: batch-size thresholds of the kernel variants */

#ifdef __cplusplus
}
#endif
#endif /* MAGPROP_AMD_H */
