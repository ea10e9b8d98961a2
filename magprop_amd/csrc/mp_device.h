// mp_device.h — structures shared by the host C-ABI (mp_capi.cpp) and the gfx950 kernels (mp_kernels.hip).
#pragma once
#include <stdint.h>

#include "../../include/magprop_amd.h"

namespace mp {

constexpr int kTile = 64;  // observation bucket size in time steps (the kernels' tiles are 64*SPL steps)

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
    int32_t pad;
};

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
    // light curves longer than 64 points: per walker [4][scratch_stride] doubles, (Mdisc, omega) at the two grid points
    // bracketing observation j >= 64 in column j - 64 (written tile by tile, read by the luminosity stage)
    double *obs_scratch;
    int32_t scratch_stride;   // max over datasets of (n_obs - 64), rounded up to 64; 0 = no long light curve
    int32_t force_pc;         // experiments: 1 = producer/consumer two-wavefront kernel, -1 = never, 0 = automatic
    // prior
    double lower[MP_MAX_NDIM];
    double upper[MP_MAX_NDIM];
    int32_t n_prior;
    uint32_t log_mask;
    // derived star constants (host-computed once from cfg)
    double GM, inertia, inv_inertia, crot /* 0.5*I/|W| */, sqrtGM, inv_sqrtGM, sqrtR;
    double crm_unit;      // (1e15 R^3)^(4/7) GM^(-1/7) f_Rm^(-2/7): Alfven-radius constant of a 1e15 G field
    // geometric grid: ratio q = t_{j+1}/t_j and the exponential Adams-Moulton quadrature matrix for it
    double q, inv_q;
    double sweep_tol;     // relative change of the step-end values that ends the Newton sweeps of a tile
    double ultra_tol;     // corrections below this let the next sweep linearise omega_dot instead of evaluating it
    int32_t n_simd;       // SIMDs of the device (multiProcessorCount x 4): batch sizes up to this get one wave per SIMD
    int32_t force_spl;    // experiments: 0 = automatic, else steps per lane of the one-wavefront kernels (2, 4)
    double eamW[4][4];
    mp_model_cfg cfg;
};

// Per-launch arguments.
struct LaunchArgs {
    const double *pars;     // [n][ndim], device
    const int32_t *ds_id;   // [n] or nullptr
    int32_t n;
    int32_t ndim;
    int32_t physical;       // 1: pars are physical, skip prior + un-logging (model_lc)
    int32_t want_chi2;      // 0: skip observations (model_lc only)
    double *lnprob;         // [n]
    int32_t *status;        // [n] or nullptr
    int32_t *sweeps;        // [n] or nullptr: total Newton sweeps over all tiles
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
};

// Steps per lane of the kernel variant used for a batch of n walkers (tiles are 64*spl steps): see launch_lnprob.
// Up to one wave per SIMD (256 CUs x 4 on MI355X) four steps per lane; beyond, two resident waves win (tools/spl_scan.sh).
inline int kernel_spl(const DevShared &sh, int n) { return n <= sh.n_simd ? 4 : 2; }

// Producer/consumer pair of wavefronts per walker: pays while every wavefront still gets a SIMD of its own.
inline bool two_wave_pair(const DevShared &sh, int n) { return 2 * n <= sh.n_simd; }

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
int launch_stretch(const DevShared &sh, const StretchArgs &g, int n_blocks, void *stream);
int launch_stretch_apply(const StretchArgs &g, void *stream);

}  // namespace mp
