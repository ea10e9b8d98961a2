// mp_device.h — structures shared by the host C-ABI (mp_capi.cpp) and the gfx950 kernels (mp_kernels.hip).
#pragma once
#include <stdint.h>

#include "../../include/magprop_amd.h"

namespace mp {

constexpr int kTile = 64;    // observation bucket size in grid intervals (the kernels' tiles are 64*SPL steps)
constexpr int kTile64 = 64;  // the same, for code inside the kernels where kTile names their own tile length

// physical constants, magnetar/funcs.py:7-13 (cgs)
constexpr double kG = 6.674e-8;
constexpr double kC = 3.0e10;
constexpr double kR = 1.0e6;
constexpr double kMsol = 1.99e33;

// One observed light curve, pre-digested on the host for the grid it will be interpolated on.
// Observations are sorted by time and bucketed by the 64-step tile whose light-curve values they need.
struct DsDesc {
    int32_t n_obs;
    int32_t obs_off;   // first observation in the packed obs_* arrays
    int32_t tile_off;  // first entry of this dataset's tile_ptr[n_tiles + 1]
    int32_t flags;     // bit 0: every observed time IS a grid point (x - tgrid[g] == 0 throughout: the reference's synthetic sets,
                       // code/synthetic_datasets/generate_data.py:61-67) -- np.interp then returns the model at grid point g itself and
                       // the state at g + 1 is never needed
};
constexpr int kDsOnKnots = 1;

// What a tile of the time-parallel solver needs to know about its step: tiles step over 1/8 of a grid interval (kind 0,
// the first intervals), 1, 2, 4 or 8 intervals (kinds 1 .. 4).  The grid is geometric, so the ratio Q of consecutive step
// end times, the quadrature matrices of the exponential Adams-Moulton formulas and the positions of the skipped grid
// points inside a step are constants of the kind (host-computed, oracle/mp_oracle.c mpo_eam_weights).
constexpr int kKinds = 5;
struct StrideK {
    double lnQ;          // ln of the ratio of consecutive step end times
    double inv_Q;        // 1/Q
    double one_m_invQ;   // 1 - 1/Q: step length = (step end time) * this
};

// The other constants of the kinds live in a device table (DevShared::wtab) that every workgroup copies into LDS once:
// per kind kWtabStride doubles,
//   [6 k + m]      W5[k][m], k, m = 0..4: order-5 quadrature on nodes t_{j+1} .. t_{j-3}; rows padded to 6
//   [32 + i]       theta_i, i = 1..7: time fraction of the i-th skipped grid point inside a step, (q^i - 1)/(Q - 1)
// (25 wave-uniform doubles per kind do not fit the scalar registers next to the walker constants; from LDS they are read
// where they are used, two per broadcast ds_read_b128).
constexpr int kWtabStride = 40;
constexpr int kWtabTheta = 32;
// behind the kinds' blocks: the dense-output weights of Mdisc where the step is longer than the viscous time (cubic Lagrange
// interpolation of Mdisc / (tvisc Mdotfb) through four nodes), [((kind - 2) * 7 + (i - 1)) * 12 + variant * 4 + k]:
// kinds 2 .. 4, skipped grid point i = 1 .. 7, variant = position of the step among the three intervals of its four nodes
// (0: the first, the step is the first of the tile's kept part; 1: the middle; 2: the last), k = node
constexpr int kWtabDense = kKinds * kWtabStride;
constexpr int kWtabSize = kWtabDense + 3 * 7 * 12;
constexpr int kTtabN = 257;

// Everything the kernel reads that is shared by all walkers (resident in HBM, L2-hot).
struct DevShared {
    const double *tgrid;  // [n_grid]
    int32_t n_grid;
    int32_t n_tiles;      // ceil((n_grid - 1) / 64): number of 64-step observation buckets
    const DsDesc *ds;     // [n_ds]
    int32_t n_ds;
    int32_t pad0;
    const int32_t *tile_ptr;  // per dataset: [n_tiles + 1] prefix offsets into the obs arrays (relative to obs_off)
    const int32_t *obs_g;     // grid interval index g: tgrid[g] <= x <= tgrid[g + 1]
    const double *obs_dx;     // x - tgrid[g]
    const double *obs_idt;    // 1 / (tgrid[g + 1] - tgrid[g])
    const double *obs_y;
    const double *obs_yerr;
    int32_t has_long;         // 1: some registered light curve has more than 64 points (the launchers pick the LONG kernel builds)
    int32_t pre_fine;         // grid intervals at the start that are covered with 1/8-interval sub-steps (32, or fewer on a tiny grid)
    // prior
    double lower[MP_MAX_NDIM];
    double upper[MP_MAX_NDIM];
    int32_t n_prior;
    uint32_t log_mask;
    // derived star constants (host-computed once from cfg)
    double GM, inertia, inv_inertia, crot /* 0.5*I/|W| */, sqrtGM, inv_sqrtGM, sqrtR;
    double crm_unit;      // (1e15 R^3)^(4/7) GM^(-1/7) f_Rm^(-2/7): Alfven-radius constant of a 1e15 G field
    // geometric grid t_i = t0 q^i
    double t0, lnq8;      // first grid time; ln(q)/8
    double sweep_tol;     // relative change of the step-end values that ends the Newton sweeps of a tile
    double stop_factor;   // the sweeps end when the next correction, estimated from the contraction, is below stop_factor x the tolerance
    double stride_tol;    // smoothness indicator above which a tile at a coarse stride is cut (cfg.stride_tol)
    double coarse_tol_factor;   // sweep tolerance of tiles over 2, 4 or 8 grid intervals, relative to sweep_tol
    double k4_tol_factor;       // stride_tol of tiles over 8 grid intervals, relative to stride_tol (0.1; oracle/mp_oracle.c)
    int32_t n_simd;       // SIMDs of the device (multiProcessorCount x 4): batch sizes up to this get one wave per SIMD
    int32_t force_spl;    // experiments: 0 = automatic, else steps per lane (2, 4) of the one-wavefront kernels
    int32_t force_waves;  // experiments: 0 = automatic, else wavefronts per walker (1, 2, 4)
    int32_t pad1;
    int32_t max_kind;     // coarsest tile kind allowed: 1, 2, 3, 4 for cfg.max_stride 1, 2, 4, 8
    int32_t coarse_max_sweeps, fine_max_sweeps, trouble_limit;   // sweeps after which a slowly converging tile keeps its converged lanes
    double early_hold_t;  // coarse tiles that start before this TIME (MP_EARLY_HOLD_SECONDS = 4 s: where the spin-up transients
                          // live; a physical time, so the "S" grid, whose index 1 024 is at 8 ms, gets the same protection) are
                          // held to a tenth of stride_tol, like the tile behind the sub-steps
    StrideK sk[kKinds];
    const double *wtab;   // [kWtabSize] quadrature matrices, skipped-point positions and dense-output weights of the tile kinds
    const double *ttab;   // [kKinds - 1][kTtabN] Q^k of the kinds 1 .. 4, k = 0 .. 256 (host-computed once per handle: the kernels
                          // used to spend 16 double-precision exponentials per lane on it at every launch)
    mp_model_cfg cfg;
};

// Per-launch arguments.
struct LaunchArgs {
    const double *pars;     // [n][ndim], device
    const int32_t *ds_id;   // [n] or nullptr
    const int32_t *order;   // [n] or nullptr: workgroup b evaluates walker order[b] (mixed-length batches: the walkers with the
                            // longest light curves first, see order_kernel); nullptr: walker b
    int32_t n;
    int32_t ndim;
    int32_t physical;       // 1: pars are physical, skip prior + un-logging (model_lc)
    int32_t want_chi2;      // 0: skip observations (model_lc only)
    double *lnprob;         // [n]
    int32_t *status;        // [n] or nullptr
    int32_t *sweeps;        // [n] or nullptr: total Newton sweeps over all tiles
    int32_t *tiles;         // [n] or nullptr: tiles solved (kept or not)
    int32_t *tile_log;      // diagnostics, or nullptr: [n][MP_TILE_LOG] one word per tile solved: kind | sweeps << 4 | kept lanes << 16
    double *ltot;           // [n][n_grid] or nullptr   (1e50 erg/s)
    double *lprop;          // [n][n_grid] or nullptr
    double *ldip;           // [n][n_grid] or nullptr
    double *mdisc;          // [n][n_grid] or nullptr
    double *omega;          // [n][n_grid] or nullptr
};

// Arguments of one fused stretch-move half-step (mp_kernels.hip: stretch_kernel).
struct StretchArgs {
    double *pos;            // [n_total][ndim] current positions, updated in place
    double *lnprob;         // [n_total]
    int64_t *n_accepted;    // [n_total]
    const int32_t *perm;    // [n_ensembles][n_walkers]: this step's random split (first n_half entries = half 0)
    const int32_t *ds_id;   // [n_total] or nullptr (dataset 0)
    double *chain;          // [n_rows][n_total][ndim] or nullptr
    double *chain_lnp;      // [n_rows][n_total]
    double *upd;            // nullptr: update in place.  Else outcome rows [slots][ndim + 3] = (proposal, lnprob, accepted,
                            // status): written by stretch_kernel (row = slot - slot_lo), read by stretch_apply_kernel (row = slot)
    double *spec;           // whole-step launches (stretch_step_kernel / stretch_step_commit_kernel): outcome rows
                            // [3][slots][ndim + kSpecExtra], see mp_kernels.hip
    double *bad_log;        // [bad_cap][ndim] or nullptr: proposals inside the prior whose model failed (the reference's fbad file)
    uint32_t *bad_count;    // number of such proposals so far (may exceed bad_cap)
    uint32_t bad_cap;
    int32_t slot_lo;        // first slot of the active half (all ensembles flattened) covered by this launch
    int32_t chain_row;
    int32_t n_walkers;      // walkers per ensemble (even)
    int32_t n_half;         // n_walkers / 2
    int32_t n_ensembles;
    int32_t n_total;        // n_walkers * n_ensembles
    int32_t ndim;
    int32_t half;           // 0 or 1: which half moves
    int32_t target;         // 0: magnetar posterior, 1: isotropic unit Gaussian (move tests)
    uint32_t step;
    uint32_t pad;
    uint64_t seed;
    double a;               // stretch scale (emcee default 2)
    uint64_t ens_order;     // half-step launches: ensemble at position p of the launch = (ens_order >> 4p) & 15; 0 = identity
};

// ---------------------------------------------------------------- compile-time constants of the stride policy
// (used by mp_eval.hpp; reported by mp_get_policy so that an A/B build made with -D overrides cannot pass for the shipped one)
// Tile kinds from this one on start their sweeps from the log-space extrapolation (5: none).  Tiles of 256 steps (4 steps
// per lane) over 2, 4, 8 grid intervals reach 0.6 - 1.2 decades ahead; at 128 steps (2 steps per lane) the quartic in the
// index is as good and cheaper (same-box A/B, profiles/r04_ab_predictor.log: 4 096 near-truth walkers +3 % slower with it).
#ifndef MP_LOGPRED_MIN_KIND
#define MP_LOGPRED_MIN_KIND 2
#endif
#ifndef MP_ABORT_SKIP_RATIO
#define MP_ABORT_SKIP_RATIO 32.0   // excess of the indicator over its bound beyond which a given-up coarse tile skips a stride
#endif
// A coarse tile that is cut at a fast feature of the solution (no kink) is followed by the stride its excess over the bound
// asks for, judged over this many lanes from the cut on (0: always single intervals, as until round 3)
#ifndef MP_CUT_BY_RATIO
#define MP_CUT_BY_RATIO 24
#endif
// 2-steps-per-lane kernels, adaptive stride: the sub-stepped start ends after its first tile (16 grid intervals) when that tile's
// indicator, scaled to a step of a whole interval (8^5 with the margin of 2 = 65 536) and held to a HUNDREDTH of the bound
// (MP_PRE_EARLY_END_FACTOR = 65 536 x 100; a tenth raised the soak's maximum from 5.6e-8 to 8.0e-8 for 1 % more speed:
// profiles/r04_ab_substeps_spl2.log), stays below the bound everywhere and no kink lies in it.  (Simply halving the sub-stepped stretch put one
// golden point -- a spin-up transient at t = 1 s -- at 1.01 of the tight bound in the serial restatement: tests/test_oracle.py.)
// Corrections below this (relative) let the next sweep keep the Jacobian, e^{h lambda} and the weights ("light" sweep).  1e-4
// until the tiles ended on the contraction estimate; scanned then on one box (profiles/r04_ab_light_tol.log): 3e-4 / 1e-3 / 3e-3 /
// 1e-2 speed up the passes near the truths by 0.7 / 1.3 / 1.3 / 2 %, and from 3e-3 on the slowest prior-wide walkers need
// more sweeps (chord iterations stall where the Jacobian changes from step to step): 0.228 -> 0.220 -> 0.239 ms.
#ifndef MP_LIGHT_TOL
#define MP_LIGHT_TOL 1.0e-3
#endif
#ifndef MP_PRE_EARLY_END_FACTOR
#define MP_PRE_EARLY_END_FACTOR 6553600.0
#endif
// 1 if any of the above differs from its shipped value (mp_get_policy()[MP_POLICY_EXPERIMENTS] is then set as for MP_EXPERIMENTS)
#define MP_POLICY_MACROS_MODIFIED (MP_LOGPRED_MIN_KIND != 2 || MP_ABORT_SKIP_RATIO != 32.0 || MP_CUT_BY_RATIO != 24 || MP_LIGHT_TOL != 1.0e-3 || MP_PRE_EARLY_END_FACTOR != 6553600.0)

// Steps per lane of the kernel variant used for a batch of n walkers (tiles are 64*spl steps): see launch_lnprob.
// Up to one wave per SIMD (256 CUs x 4 on MI355X) four steps per lane; beyond, two resident waves win (tools/spl_scan.sh).
inline int kernel_spl(const DevShared &sh, int n) { return n <= sh.n_simd ? 4 : 2; }
// The same for the kernels that write light curves to HBM (mode B), whose time is rounds of resident workgroups.  The
// 2-steps-per-lane build keeps 22.6 KB of LDS per workgroup: 7 workgroups per CU, not 8 -- 1 792 resident walkers on an MI355X,
// so that 2 048 walkers take TWO rounds (0.49 ms) where the 4-steps-per-lane build, one wavefront per SIMD and no scratch
// memory, takes two rounds of 1 024 in 0.34 ms.  Measured per round on one box (profiles/r05_ab_curve_spl.log): 0.171 ms
// (4 steps per lane, sharp rounds: the workgroups of a round end together) and 0.277 ms (2 steps per lane; beyond two rounds
// the workgroups overlap and the time grows linearly).  The cheaper one by that model: 2 048 walkers 4.16 -> 5.97 M
// evaluations/s, 3 072 5.45 -> 6.01 M, 4 096 5.66 -> 6.03 M, unchanged elsewhere.
inline int kernel_spl_curves(const DevShared &sh, int n) {
    if (n <= sh.n_simd) return 4;
    const double cap2 = 7.0 * (sh.n_simd / 4);                            // resident workgroups of the 2-steps-per-lane build
    const double r4 = (double)((n + sh.n_simd - 1) / sh.n_simd);          // rounds of the 4-steps-per-lane build
    const double x2 = n / cap2;
    const double r2 = 1.62 * (x2 <= 2.0 ? (x2 <= 1.0 ? 1.0 : 2.0) : x2 + 0.2);   // the other one's, in units of the former's
    return r4 <= r2 ? 4 : 2;
}
// Wavefronts per walker: launches that would leave SIMDs idle (n <= n_simd / 2) put a team of 4 wavefronts on every walker
// (mp_eval.hpp TeamX; mode A; handles with light curves of more than 64 points: the LONG builds of the same kernels); see
// launch_lnprob.  (force_waves = 2: the two-wavefront team, experiments build only.)
inline int kernel_waves(const DevShared &sh, int n) {
#ifdef MP_EXPERIMENTS
    if (sh.force_waves) return sh.force_waves;
#endif
    if (sh.force_spl) return 1;
    return 2 * n <= sh.n_simd ? 4 : 1;
}

// The same for the kernels of the device-resident sampler, by the size of a WHOLE step of the ensembles (3 x slots of a half:
// what mp_sampler_run evaluates per launch when that fits) -- for one launch per step and one per half-step alike, so that the
// two forms run the same arithmetic and their chains stay equal bit for bit; likewise every rank of a walker-sharded run.
// Ensembles of up to n_simd / 6 walkers in all (170 on an MI355X: what emcee is usually run with; the reference's own example
// has 24, code/synthetic_datasets/synth_mcmc.py): a step 0.091 -> 0.075 ms.
inline int stretch_waves(const DevShared &sh, int whole_step_blocks) {
    if (sh.force_spl) return 1;
#ifdef MP_EXPERIMENTS
    if (sh.force_waves) return sh.force_waves == 4 ? 4 : 1;
#endif
    return 2 * whole_step_blocks <= sh.n_simd ? 4 : 1;
}

// mp_sampler_run evaluates a whole step per launch (3 x slots evaluations, a third of them discarded) while that beats two
// half-step launches: up to 19/8 x n_simd evaluations (2 432 on an MI355X: ensembles of up to 1 621 walkers).  Two wavefronts per
// SIMD hold 2 x n_simd of them at once; a little beyond, the second round is still shorter than a second launch (ms per step, whole
// step / two half-steps: 1 364 walkers 0.140 / 0.191, 1 536 walkers 0.173 / 0.193, 1 700 walkers 0.201 / 0.193, 2 048 walkers
// 0.216 / 0.195; until round 5 the limit was 2 x n_simd).
inline bool stretch_whole_step_fits(const DevShared &sh, long long whole_step_blocks) { return 8 * whole_step_blocks <= 19 * (long long)sh.n_simd; }

// Arguments of the batched right-hand-side evaluation (mp_kernels.hip: rhs_kernel), device pointers.
struct RhsArgs {
    const double *pars;     // [n][ndim], physical units
    const double *t;        // [n]
    const double *y;        // [n][2] = (Mdisc, omega)
    double *dydt;           // [n][2]
    double *lam;            // [n] or nullptr
    int32_t n;
    int32_t ndim;
};

// implemented in mp_kernels.hip; returns hipError_t as int
int launch_rhs(const DevShared &sh, const RhsArgs &r, void *stream);
int launch_lnprob(const DevShared &sh, const LaunchArgs &a, void *stream);
int launch_order(const DevShared &sh, const int32_t *ds_id, int n, int32_t *order, void *stream);   // fills order[n] (see order_kernel)
int launch_stretch(const DevShared &sh, const StretchArgs &g, int n_blocks, void *stream);
int launch_stretch_apply(const StretchArgs &g, void *stream);
int launch_stretch_step(const DevShared &sh, const StretchArgs &g, int n_blocks, void *stream);   // blocks [g.slot_lo, + n_blocks) of 3 * n_half * n_ensembles
int launch_stretch_step_commit(const StretchArgs &g, void *stream);
// columns behind the proposal in an outcome row of a whole-step launch
constexpr int kSpecExtra = 6;   // lnprob, status, (ndim - 1) ln z, ln u, lnprob of the walker before the move, partner's slot

}  // namespace mp
