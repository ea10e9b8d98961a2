// mp_kernels.hip — gfx950 (CDNA4) kernels of the magnetar log-posterior hot path and their launchers.
//
// Layout: ONE WALKER PER WAVEFRONT, SPL CONSECUTIVE TIME STEPS PER LANE.  The 10 000 grid intervals are processed in
// tiles of 64*SPL steps; inside a tile all steps advance at once (parallel in time).  Scheme = exponential
// Adams-Moulton of order 5 on geometric grids (serial restatement: oracle/mp_oracle.c
// mpo_trajectory_mode): a tile's step spans 1/8 (the first 32 intervals), 1, 2, 4 or 8 grid intervals, chosen tile by tile
// from the solution's smoothness; the states at skipped grid points come from the step's Hermite interpolant.
// Per tile (mp_eval.hpp, walker_eval):
//
//   1. Mdisc obeys dM/dt = Mdotfb(t) - M/tvisc (linear, omega-independent; reference RHS
//      code/synthetic_datasets/funcs.py:122-129, magnetar/funcs.py:86-92).  Every lane evaluates Mdotfb at its step
//      ends, fetches the four previous values from its neighbour (DPP), builds and composes the affine maps
//      M_{j+1} = e^{-h/tvisc} M_j + b_j of its steps; a wavefront scan of affine maps yields Mdisc at all step ends.
//   2. omega obeys a scalar nonlinear ODE fed by Mdisc(t).  Every lane evaluates omega_dot and its Jacobian lambda
//      ONCE per sweep at its current guess of omega at its step ends, fetches (omega_dot, omega) of the four
//      previous points, and forms the step maps omega_{j+1} = e^{h lambda} omega_j + h sum_m phi_{m+1}(h lambda) g_m.
//      A second affine scan propagates the tile's start value through all linearised maps.  This Newton-type sweep
//      contracts by 1e-2..1e-3 per pass (about 2 sweeps from an extrapolated guess, the last one being the convergence
//      check), terminates in any case after at most tile-length sweeps, and reproduces the serial recurrence to ~1e-11.
//   3. Observations.  np.interp needs the model only at the two grid points bracketing each observed time: when an
//      observation falls in the tile, the tile's (Mdisc, omega) image goes to LDS and the lane that owns the
//      observation keeps the two bracketing states (the first 64 observations, register-resident; their luminosity --
//      reference luminosity stage, code/synthetic_datasets/funcs.py:175-229, magnetar/funcs.py:157-210 -- is evaluated
//      once, after the last tile) or scores it at once from the image (observations 64.. of a longer light curve).
//      When curve outputs are requested (CURVES) the luminosity is evaluated at every step end instead, the
//      tile's light curve is staged in LDS for the interpolation and written to HBM with coalesced stores.
//   4. A wavefront reduction of the per-lane chi^2 terms gives -0.5*chi^2 (code/synthetic_datasets/mcmc_eqns.py:25).
//
// Kernels: lnprob_kernel<CURVES, SPL, LONG> (one wavefront per walker), stretch_kernel<SPL, LONG> (emcee's stretch
// move fused around it) and stretch_apply_kernel (the state update of a half-step whose proposals were evaluated
// on several GPUs); LONG = built with the path for light curves of more than 64 points.
// No MFMA (no dense contraction anywhere on this path), fp64 throughout; bound by the VALU issue rate of one wave per
// SIMD (profiles/, tools/ubench).  The arithmetic is algebraically simplified with respect to the reference formulas
// (e.g. fastness w = (Rm/Rc)^1.5 = omega*Rm^1.5/sqrt(GM), eta1-eta2 = -tanh); oracle/mp_oracle.c keeps the literal
// formulas and the tests compare the two.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mp_eval.hpp"

namespace mp {

// ---------------------------------------------------------------- batched log-posterior kernel
// LOG: the build with the per-tile diagnostics (mp_tile_log(h, 1); developer builds: always), launched only on request
#if defined(MP_PHASE_PROFILE) || defined(MP_SWEEP_TRACE) || defined(MP_CORR_TRACE) || defined(MP_ABORT_STUDY)
constexpr bool kAlwaysLog = true;
#else
constexpr bool kAlwaysLog = false;
#endif
template <bool CURVES, int SPL, bool LONG, bool LOG = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SPL >= 4 ? 1 : 2, SPL >= 4 ? 1 : 2))) void lnprob_kernel(const DevShared sh, const LaunchArgs a) {
    __shared__ TileImage<SPL> im;
    __shared__ TimeTable<SPL> tt;
    __shared__ double Lbuf[CURVES ? 8 * 64 * SPL + 1 : 1];   // up to 8 grid points per step
    tables_init<SPL, 64>(sh, tt);
    const int walker = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    double par[MP_MAX_NDIM];
    const double *pw = a.pars + (size_t)walker * a.ndim;
#pragma unroll
    for (int i = 0; i < MP_MAX_NDIM; ++i) par[i] = i < a.ndim ? pw[i] : 0.0;
    double lnp;
    int status, sweeps, tiles;
    if constexpr (CURVES) {
        walker_eval<CURVES, SPL, LONG, LOG>(sh, a, walker, par, im, tt, Lbuf, lnp, status, sweeps, tiles);
    } else {
        // (only the curve kernels serve mp_model_lc: here the parameters are the sampler's and chi^2 is wanted, at compile time)
        LaunchArgs aa = a;
        aa.physical = 0;
        aa.want_chi2 = 1;
        walker_eval<CURVES, SPL, LONG, LOG>(sh, aa, walker, par, im, tt, Lbuf, lnp, status, sweeps, tiles);
    }
    if (threadIdx.x == 0) {
        a.lnprob[walker] = lnp;
        if (a.status) a.status[walker] = status;
        if (a.sweeps) a.sweeps[walker] = sweeps;
        if (a.tiles) a.tiles[walker] = tiles;
    }
}

// The same on a team of W wavefronts per walker (mp_eval.hpp TeamX): launches that would leave SIMDs idle.  One wavefront per
// SIMD (each takes what it needs of the register file), so a workgroup is spread over W SIMDs of its CU.
// OCC = wavefronts resident per SIMD the build is made for: 1 while the launch has a SIMD for every wavefront (n <= n_simd / W),
// 2 for twice as many walkers (256 registers; the walker's constants in scalar registers).
// LONG: built with the path for light curves of more than 64 points (their observations 64.. are scored chunk by chunk, the
// chunks dealt to the team's wavefronts in turn: a walker on a 1 944-point light curve spends a quarter of its time there).
template <int SPL, int W, int OCC, bool LOG = false, bool LONG = false>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void lnprob_team_kernel(const DevShared sh, const LaunchArgs a) {
    __shared__ TileImage<SPL * W> im;
    __shared__ TimeTable<SPL * W> tt;
    __shared__ TeamX<SPL * W> tx;
    __shared__ double Lbuf[1];
    tables_init<SPL * W, 64 * W>(sh, tt);
    const int walker = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    double par[MP_MAX_NDIM];
    const double *pw = a.pars + (size_t)walker * a.ndim;
#pragma unroll
    for (int i = 0; i < MP_MAX_NDIM; ++i) par[i] = i < a.ndim ? pw[i] : 0.0;
    double lnp;
    int status, sweeps, tiles;
    LaunchArgs aa = a;
    aa.physical = 0;
    aa.want_chi2 = 1;
    walker_eval<false, SPL, LONG, LOG, W, OCC >= 2>(sh, aa, walker, par, im, tt, Lbuf, lnp, status, sweeps, tiles, &tx);
    if (threadIdx.x == 0) {
        a.lnprob[walker] = lnp;
        if (a.status) a.status[walker] = status;
        if (a.sweeps) a.sweeps[walker] = sweeps;
        if (a.tiles) a.tiles[walker] = tiles;
    }
}

// ---------------------------------------------------------------- launch order of a mixed-length batch
// The hardware starts workgroups in index order, a new one whenever a wave slot frees up.  In a batch whose walkers refer to
// light curves of very different lengths (BASELINE config 5: 8 ... 1 944 points; a walker on the longest costs twice a walker
// on a 50-point set) and that needs more than one round of the device's wave slots, a long walker that starts in the last
// round decides the launch time.  This kernel (one workgroup) sorts the walker indices by the length class of their light
// curve, longest first (counting sort; the order inside a class is immaterial: every walker writes its own outputs only), and
// lnprob_kernel evaluates walker order[blockIdx.x].  4 096-walker launch of config 5: 0.391 -> 0.313 ms on one box.
constexpr int kOrderClasses = 6;
MP_DEV int order_class(const DevShared &sh, const int32_t *ds_id, int i) {
    const int d = ds_id[i];
    const int n_obs = (sh.ds != nullptr && d >= 0 && d < sh.n_ds) ? sh.ds[d].n_obs : 0;
    return n_obs > 1024 ? 0 : (n_obs > 512 ? 1 : (n_obs > 256 ? 2 : (n_obs > 128 ? 3 : (n_obs > 64 ? 4 : 5))));
}
__global__ __launch_bounds__(1024) void order_kernel(const DevShared sh, const int32_t *ds_id, int n, int32_t *order) {
    __shared__ int cnt[kOrderClasses], fill[kOrderClasses];
    const int t = threadIdx.x;
    if (t < kOrderClasses) cnt[t] = 0;
    __syncthreads();
    for (int i = t; i < n; i += 1024) atomicAdd(&cnt[order_class(sh, ds_id, i)], 1);
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int c = 0; c < kOrderClasses; ++c) { fill[c] = acc; acc += cnt[c]; }
    }
    __syncthreads();
    for (int i = t; i < n; i += 1024) order[atomicAdd(&fill[order_class(sh, ds_id, i)], 1)] = i;
}

int launch_order(const DevShared &sh, const int32_t *ds_id, int n, int32_t *order, void *stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sh, ds_id, n, order);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- fused stretch-move half-step kernel
// Counter-based RNG (Philox4x32-10, Salmon et al. 2011): one independent stream per (seed, step, walker).
MP_DEV void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Unfused double arithmetic (hipcc contracts a*b+c into an FMA by default, also through __dmul_rn/__dadd_rn):
// the proposal and the test target are computed with separately rounded operations so that a numpy
// restatement of the move reproduces the chain bit for bit.
MP_DEV double mul_rn(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
MP_DEV double add_rn(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
MP_DEV double sub_rn(double a, double b) {
#pragma clang fp contract(off)
    return a - b;
}

MP_DEV double u01(uint32_t hi, uint32_t lo) {   // 53-bit uniform in [0, 1)
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

// One wavefront = one walker of the active half: draw the partner and the stretch factor, form the
// proposal (emcee's StretchMove.get_proposal), evaluate its log-posterior with walker_eval, accept or
// reject against the walker's current value, update position / lnprob / counters in place and write the
// step's row of the chain.  Walkers of the complementary half are only read, so the update is race-free.
// A launch covers the slots [slot_lo, slot_lo + gridDim.x) of the active half (all ensembles flattened).  With g.upd set
// (walker-sharded ensembles, one process per GPU) nothing is updated in place: the outcome of slot s goes to row
// s - slot_lo of g.upd as (proposal[ndim], its lnprob, accepted 0/1) and stretch_apply_kernel commits the rows of all
// ranks after the all-gather.  The random numbers are keyed by (seed; step, half, walker), so every rank draws what the
// single-GPU launch would have drawn for the same walker.
// Half-step launches run the ensembles in the order of StretchArgs::ens_order (mp_capi.cpp: longest light curve first, four
// bits per position; 0 = the ensembles as they are numbered): the waves that take longest start first, as in order_kernel.
__device__ __forceinline__ int ens_of_slot(const StretchArgs &g, int e_pos) {
    return g.ens_order ? (int)((g.ens_order >> (4 * e_pos)) & 15u) : e_pos;
}

// (the team's exchange area in LDS, or nothing for the one-wavefront builds)
template <int G, bool ON>
struct TeamLds {
    TeamX<G> x;
    __device__ TeamX<G> *ptr() { return &x; }
};
template <int G>
struct TeamLds<G, false> {
    __device__ TeamX<G> *ptr() { return nullptr; }
};

// W, OCC: small ensembles evaluate every proposal on a team of W = 4 wavefronts (lnprob_team_kernel; OCC = wavefronts resident
// per SIMD the build is made for), chosen by the size of a WHOLE step of the sampler (stretch_waves, mp_device.h) so that one
// launch per step and one per half-step run the same arithmetic: the chains stay equal bit for bit.
template <int SPL, bool LONG, int W = 1, int OCC = 0>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(W > 1 ? OCC : (SPL >= 4 ? 1 : 2), W > 1 ? OCC : (SPL >= 4 ? 1 : 2))))
void stretch_kernel(const DevShared sh, const StretchArgs g) {
    __shared__ TileImage<SPL * W> im;
    __shared__ TimeTable<SPL * W> tt;
    __shared__ double lds[1];
    __shared__ double park[MP_MAX_NDIM + 3];
    __shared__ TeamLds<SPL * W, (W > 1)> tl;
    TeamX<SPL * W> *const tx = tl.ptr();
    tables_init<SPL * W, 64 * W>(sh, tt);
    const int gs = g.slot_lo + (int)blockIdx.x;                    // slot of the active half, all ensembles flattened
    const int w_ens = ens_of_slot(g, gs / g.n_half);               // which ensemble (longest light curve first)
    const int slot = gs % g.n_half;                                // which walker of the active half
    const int32_t *perm = g.perm + (size_t)w_ens * g.n_walkers;    // this step's random split of the ensemble
    const int base = w_ens * g.n_walkers;
    const int k = base + perm[g.half * g.n_half + slot];           // active walker (global index)
    uint32_t r[4], r2[4];
    philox4x32_10((uint32_t)g.seed, (uint32_t)(g.seed >> 32), (uint32_t)g.step, (uint32_t)g.half, (uint32_t)k, 0u, r);
    philox4x32_10((uint32_t)g.seed, (uint32_t)(g.seed >> 32), (uint32_t)g.step, (uint32_t)g.half, (uint32_t)k, 1u, r2);
    const int n_comp = g.n_walkers - g.n_half;
    const int jc = (int)(u01(r[0], r[1]) * n_comp);                // partner from the complementary half
    const int j = base + perm[(1 - g.half) * g.n_half + min(jc, n_comp - 1)];
    // (unfused arithmetic below, so that a numpy restatement of the move reproduces the chain bit for bit)
    const double zr = add_rn(mul_rn(g.a - 1.0, u01(r[2], r[3])), 1.0);
    const double zz = mul_rn(zr, zr) / g.a;                      // g(z) ~ 1/sqrt(z) on [1/a, a]
    double par[MP_MAX_NDIM];
#pragma unroll
    for (int i = 0; i < MP_MAX_NDIM; ++i) {
        const double xk = i < g.ndim ? g.pos[(size_t)k * g.ndim + i] : 0.0;
        const double xj = i < g.ndim ? g.pos[(size_t)j * g.ndim + i] : 0.0;
        par[i] = sub_rn(xj, mul_rn(sub_rn(xj, xk), zz));
    }
    // Nothing of the draw stays live across walker_eval (which needs every register): lane 0 parks the proposal, the
    // acceptance threshold and the walker's current value in LDS and reads them back behind the evaluation.
    //   park[0 .. ndim-1] proposal, [ndim] (ndim - 1) ln z, [ndim + 1] ln u, [ndim + 2] lnprob of the walker now
    const bool lane0 = W > 1 ? threadIdx.x == 0 : (threadIdx.x & 63) == 0;   // (of the team's first wavefront)
    double lnp = 0.0;
    if (g.target == 1) {   // isotropic unit Gaussian: exercises the move itself (tests)
#pragma unroll
        for (int i = 0; i < MP_MAX_NDIM; ++i) lnp = i < g.ndim ? sub_rn(lnp, mul_rn(mul_rn(0.5, par[i]), par[i])) : lnp;
    }
    if (lane0) {
#pragma unroll
        for (int i = 0; i < MP_MAX_NDIM; ++i) park[i] = par[i];
        park[MP_MAX_NDIM] = mul_rn(g.ndim - 1.0, log(zz));
        park[MP_MAX_NDIM + 1] = log(u01(r2[0], r2[1]));
        park[MP_MAX_NDIM + 2] = g.lnprob[k];
    }
    int status = MP_STATUS_OK, sweeps, tiles;
    if (g.target != 1) {
        LaunchArgs a{};
        a.ds_id = g.ds_id;
        a.ndim = g.ndim;
        a.physical = 0;
        a.want_chi2 = 1;
        if constexpr (W > 1) walker_eval<false, SPL, LONG, false, W, OCC >= 2>(sh, a, k, par, im, tt, lds, lnp, status, sweeps, tiles, tx);
        else walker_eval<false, SPL, LONG>(sh, a, k, par, im, tt, lds, lnp, status, sweeps, tiles);
    }
    if (lane0) {   // lane 0 of the evaluating wavefront (team: of its first one)
        const double lnp_old = park[MP_MAX_NDIM + 2];
        const double lnpdiff = sub_rn(add_rn(park[MP_MAX_NDIM], lnp), lnp_old);
        const bool accept = lnpdiff > park[MP_MAX_NDIM + 1];      // false for NaN / -inf proposals
        if (g.upd) {
            double *u = g.upd + (size_t)blockIdx.x * (g.ndim + 3);
            for (int i = 0; i < g.ndim; ++i) u[i] = park[i];
            u[g.ndim] = lnp;
            u[g.ndim + 1] = accept ? 1.0 : 0.0;
            u[g.ndim + 2] = (double)status;
        } else {
            if (accept) {
                for (int i = 0; i < g.ndim; ++i) g.pos[(size_t)k * g.ndim + i] = park[i];
                g.lnprob[k] = lnp;
                g.n_accepted[k] += 1;
            }
            if (g.chain) {
                double *c = g.chain + ((size_t)g.chain_row * g.n_total + k) * g.ndim;
                // (the walker's position AFTER the decision: written just above when the proposal was accepted.  A select between
                // the LDS slot and the global row made the compiler select the ADDRESS and load through a generic pointer:
                // nine flat loads and a private segment reserved for their sake)
                for (int i = 0; i < g.ndim; ++i) c[i] = g.pos[(size_t)k * g.ndim + i];
                g.chain_lnp[(size_t)g.chain_row * g.n_total + k] = accept ? lnp : lnp_old;
            }
            if (g.bad_log && (status == MP_STATUS_FLAG || status == MP_STATUS_NONFINITE)) {
                // the reference appends such parameter sets to its `fbad` file (code/synthetic_datasets/mcmc_eqns.py:72-79)
                const unsigned slot_b = atomicAdd(g.bad_count, 1u);
                if (slot_b < g.bad_cap)
                    for (int i = 0; i < g.ndim; ++i) g.bad_log[(size_t)slot_b * g.ndim + i] = park[i];
            }
        }
    }
}

// Commit one half-step from the gathered outcome rows (see stretch_kernel): one thread per slot of the active half.
__global__ __launch_bounds__(256) void stretch_apply_kernel(const StretchArgs g) {
    const int gs = blockIdx.x * 256 + threadIdx.x;
    if (gs >= g.n_half * g.n_ensembles) return;
    const int w_ens = ens_of_slot(g, gs / g.n_half), slot = gs % g.n_half;
    const int k = w_ens * g.n_walkers + g.perm[(size_t)w_ens * g.n_walkers + g.half * g.n_half + slot];
    const double *u = g.upd + (size_t)gs * (g.ndim + 3);
    const bool accept = u[g.ndim + 1] != 0.0;
    const double lnp_old = g.lnprob[k];
    if (accept) {
        for (int i = 0; i < g.ndim; ++i) g.pos[(size_t)k * g.ndim + i] = u[i];
        g.lnprob[k] = u[g.ndim];
        g.n_accepted[k] += 1;
    }
    if (g.chain) {
        double *c = g.chain + ((size_t)g.chain_row * g.n_total + k) * g.ndim;
        for (int i = 0; i < g.ndim; ++i) c[i] = accept ? u[i] : g.pos[(size_t)k * g.ndim + i];
        g.chain_lnp[(size_t)g.chain_row * g.n_total + k] = accept ? u[g.ndim] : lnp_old;
    }
    const int status = (int)u[g.ndim + 2];
    if (g.bad_log && (status == MP_STATUS_FLAG || status == MP_STATUS_NONFINITE)) {
        const unsigned slot_b = atomicAdd(g.bad_count, 1u);
        if (slot_b < g.bad_cap)
            for (int i = 0; i < g.ndim; ++i) g.bad_log[(size_t)slot_b * g.ndim + i] = u[i];
    }
}

// ---------------------------------------------------------------- a whole stretch-move step in one launch
// A half-step of a small ensemble leaves most of the chip idle: 512 proposals on the 1 024 SIMDs of an MI355X, and the second
// half-step cannot start before the first has decided.  But everything the second half needs is known up front except those
// decisions: walker k of the second half moves along the line to its partner j of the first half, who will stand either
// where it stands now or at its own proposal Y_j, and Y_j is known before it is evaluated.  One launch therefore evaluates
// the n/2 proposals of the first half AND, for every walker of the second half, BOTH candidate proposals (3 n/2 evaluations,
// n/2 of them discarded); stretch_step_commit_kernel then takes the first half's decisions, picks the candidate that matches
// the partner's outcome and decides on it.  The chain is the one the two half-step launches produce, bit for bit (same
// random numbers, same arithmetic, same order of decisions).  Used when 3 n/2 waves fit the device two per SIMD.
//
// Block b of 3 * slots: type = b / slots (0: first half; 1: second half, partner where it stands; 2: second half, partner at
// its proposal), slot = b % slots.  Outcome row of block b = proposal[ndim], lnprob, status, (ndim - 1) ln z, ln u,
// lnprob of the walker before the move, partner's slot; written at row b - slot_lo of g.spec (walker-sharded ensembles: every
// rank evaluates its share of the blocks, the rows are all-gathered, stretch_step_commit_kernel runs on every rank).
MP_DEV void stretch_draw(const StretchArgs &g, int half, int k, int n_comp, int &jc, double &zz, double &logu) {
    uint32_t r[4], r2[4];
    philox4x32_10((uint32_t)g.seed, (uint32_t)(g.seed >> 32), (uint32_t)g.step, (uint32_t)half, (uint32_t)k, 0u, r);
    philox4x32_10((uint32_t)g.seed, (uint32_t)(g.seed >> 32), (uint32_t)g.step, (uint32_t)half, (uint32_t)k, 1u, r2);
    jc = min((int)(u01(r[0], r[1]) * n_comp), n_comp - 1);
    const double zr = add_rn(mul_rn(g.a - 1.0, u01(r[2], r[3])), 1.0);
    zz = mul_rn(zr, zr) / g.a;
    logu = log(u01(r2[0], r2[1]));
}

template <int SPL, bool LONG, int W = 1, int OCC = 0>   // (W, OCC: as in stretch_kernel)
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(W > 1 ? OCC : (SPL >= 4 ? 1 : 2), W > 1 ? OCC : (SPL >= 4 ? 1 : 2))))
void stretch_step_kernel(const DevShared sh, const StretchArgs g) {
    __shared__ TileImage<SPL * W> im;
    __shared__ TimeTable<SPL * W> tt;
    __shared__ double lds[1];
    __shared__ TeamLds<SPL * W, (W > 1)> tl;
    TeamX<SPL * W> *const tx = tl.ptr();
    tables_init<SPL * W, 64 * W>(sh, tt);
    const int n_slots = g.n_half * g.n_ensembles;
    const int blk = g.slot_lo + (int)blockIdx.x;                    // a launch covers blocks [slot_lo, slot_lo + gridDim.x) (sharded: a rank's share)
    const int type = blk / n_slots, gs = blk - type * n_slots;
    const int half = type == 0 ? 0 : 1;
    const int w_ens = gs / g.n_half, slot = gs - w_ens * g.n_half;
    const int32_t *perm = g.perm + (size_t)w_ens * g.n_walkers;
    const int base = w_ens * g.n_walkers;
    const int n_comp = g.n_walkers - g.n_half;
    const int k = base + perm[half * g.n_half + slot];
    int jc;
    double zz, logu;
    stretch_draw(g, half, k, n_comp, jc, zz, logu);
    const int j = base + perm[(1 - half) * g.n_half + jc];
    double xj[MP_MAX_NDIM];
#pragma unroll
    for (int i = 0; i < MP_MAX_NDIM; ++i) xj[i] = i < g.ndim ? g.pos[(size_t)j * g.ndim + i] : 0.0;
    bool skip = false;
    if (type == 2) {
        // the partner's own proposal of the first half-step (the arithmetic of its type-0 block, bit for bit)
        int jcj;
        double zzj, loguj;
        stretch_draw(g, 0, j, n_comp, jcj, zzj, loguj);
        const int jj = base + perm[g.n_half + jcj];
        bool outside = false;
#pragma unroll
        for (int i = 0; i < MP_MAX_NDIM; ++i) {
            const double xjj = i < g.ndim ? g.pos[(size_t)jj * g.ndim + i] : 0.0;
            xj[i] = sub_rn(xjj, mul_rn(sub_rn(xjj, xj[i]), zzj));
            if (g.target == 0 && i < sh.n_prior && (!(xj[i] >= sh.lower[i]) || !(xj[i] <= sh.upper[i]))) outside = true;
        }
        skip = outside;   // a proposal outside the prior box is never accepted: this candidate cannot be the one
    }
    double par[MP_MAX_NDIM];
#pragma unroll
    for (int i = 0; i < MP_MAX_NDIM; ++i) {
        const double xk = i < g.ndim ? g.pos[(size_t)k * g.ndim + i] : 0.0;
        par[i] = sub_rn(xj[i], mul_rn(sub_rn(xj[i], xk), zz));
    }
    double lnp = -INFINITY;
    int status = MP_STATUS_PRIOR, sweeps, tiles;
    if (!skip && g.target == 1) {   // isotropic unit Gaussian: exercises the move itself (tests)
        lnp = 0.0;
        status = MP_STATUS_OK;
#pragma unroll
        for (int i = 0; i < MP_MAX_NDIM; ++i) lnp = i < g.ndim ? sub_rn(lnp, mul_rn(mul_rn(0.5, par[i]), par[i])) : lnp;
    }
    // Everything of the outcome row that does not depend on the evaluation is written NOW: nothing of the draw stays live
    // across walker_eval, which needs every register (round 3 held the proposal, the partner and the thresholds in registers
    // there: 180 B of scratch per lane, 16 MB of spill traffic per launch).
    const bool lane0 = W > 1 ? threadIdx.x == 0 : (threadIdx.x & 63) == 0;   // (of the team's first wavefront)
    if (lane0) {
        double *u = g.spec + (size_t)blockIdx.x * (g.ndim + kSpecExtra);
        for (int i = 0; i < g.ndim; ++i) u[i] = par[i];
        u[g.ndim] = lnp;
        u[g.ndim + 1] = (double)status;
        u[g.ndim + 2] = mul_rn(g.ndim - 1.0, log(zz));
        u[g.ndim + 3] = logu;
        u[g.ndim + 4] = g.lnprob[k];
        u[g.ndim + 5] = (double)jc;
    }
    if (!skip && g.target != 1) {
        LaunchArgs a{};
        a.ds_id = g.ds_id;
        a.ndim = g.ndim;
        a.physical = 0;
        a.want_chi2 = 1;
        if constexpr (W > 1) walker_eval<false, SPL, LONG, false, W, OCC >= 2>(sh, a, k, par, im, tt, lds, lnp, status, sweeps, tiles, tx);
        else walker_eval<false, SPL, LONG>(sh, a, k, par, im, tt, lds, lnp, status, sweeps, tiles);
        if (lane0) {
            double *u = g.spec + (size_t)blockIdx.x * (g.ndim + kSpecExtra);
            u[g.ndim] = lnp;
            u[g.ndim + 1] = (double)status;
        }
    }
}

// The decisions of a whole step from the outcome rows of stretch_step_kernel: one thread per walker.
__global__ __launch_bounds__(256) void stretch_step_commit_kernel(const StretchArgs g) {
    const int n_slots = g.n_half * g.n_ensembles, R = g.ndim + kSpecExtra;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 2 * n_slots) return;
    const int half = idx / n_slots, gs = idx - half * n_slots;
    const int w_ens = gs / g.n_half, slot = gs - w_ens * g.n_half;
    const int k = w_ens * g.n_walkers + g.perm[(size_t)w_ens * g.n_walkers + half * g.n_half + slot];
    auto accepted = [&](const double *u) {   // emcee: lnpdiff = (ndim - 1) ln z + lnprob(proposal) - lnprob(walker) > ln u
        return sub_rn(add_rn(u[g.ndim + 2], u[g.ndim]), u[g.ndim + 4]) > u[g.ndim + 3];
    };
    const double *u = g.spec + (size_t)gs * R;
    if (half == 1) {
        const double *u1 = g.spec + (size_t)(n_slots + gs) * R;
        const int gs_j = w_ens * g.n_half + (int)u1[g.ndim + 5];                  // the partner's slot in the first half
        const bool moved = accepted(g.spec + (size_t)gs_j * R);
        u = moved ? g.spec + (size_t)(2 * n_slots + gs) * R : u1;
    }
    const bool accept = accepted(u);
    const double lnp_old = u[g.ndim + 4];
    if (accept) {
        for (int i = 0; i < g.ndim; ++i) g.pos[(size_t)k * g.ndim + i] = u[i];
        g.lnprob[k] = u[g.ndim];
        g.n_accepted[k] += 1;
    }
    if (g.chain) {
        double *c = g.chain + ((size_t)g.chain_row * g.n_total + k) * g.ndim;
        for (int i = 0; i < g.ndim; ++i) c[i] = accept ? u[i] : g.pos[(size_t)k * g.ndim + i];
        g.chain_lnp[(size_t)g.chain_row * g.n_total + k] = accept ? u[g.ndim] : lnp_old;
    }
    const int status = (int)u[g.ndim + 1];
    if (g.bad_log && (status == MP_STATUS_FLAG || status == MP_STATUS_NONFINITE)) {
        const unsigned slot_b = atomicAdd(g.bad_count, 1u);
        if (slot_b < g.bad_cap)
            for (int i = 0; i < g.ndim; ++i) g.bad_log[(size_t)slot_b * g.ndim + i] = u[i];
    }
}

// ---------------------------------------------------------------- the right-hand side at arbitrary states
// One state per lane, evaluated by the device functions the solver kernels use (mdot_fb, disc_point, omega_rhs):
// dMdisc/dt = Mdotfb - Mdisc/tvisc (eta1 + eta2 = 1, code/synthetic_datasets/funcs.py:122-129) and domega/dt
// (funcs.py:119,131-140).  Serves `odes`/`ODEs` of the Python front end and pins the simplified algebra of the kernels
// against the reference's literal formulas point by point.
__global__ __launch_bounds__(64) void rhs_kernel(const DevShared sh, const RhsArgs r) {
    ktab_init();
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int ii = min(i, r.n - 1);                       // idle lanes repeat the last point (wave-wide votes inside)
    double par[MP_MAX_NDIM];
#pragma unroll
    for (int k = 0; k < MP_MAX_NDIM; ++k) par[k] = k < r.ndim ? r.pars[(size_t)ii * r.ndim + k] : 0.0;
    LaunchArgs a{};
    a.ndim = r.ndim;
    a.physical = 1;
    Walker w;
    (void)walker_setup(sh, a, par, w);
    const Vd<1> tv{{r.t[ii]}}, Mv{{r.y[2 * (size_t)ii]}}, ov{{r.y[2 * (size_t)ii + 1]}};
    const Vd<1> S = mdot_fb(w, tv);
    const DiscPt<1> d = disc_point(sh, w, Mv);
    Vd<1> rot, lam;
    const Vd<1> f = omega_rhs<true, true>(sh, w, d, ov, rot, lam);
    if (i < r.n) {
        r.dydt[2 * (size_t)i] = S[0] - Mv[0] * w.inv_tau;
        r.dydt[2 * (size_t)i + 1] = f[0];
        if (r.lam) r.lam[i] = lam[0];
    }
}

int launch_rhs(const DevShared &sh, const RhsArgs &r, void *stream) {
    if (r.n <= 0) return 0;
    hipLaunchKernelGGL(rhs_kernel, dim3((unsigned)((r.n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, sh, r);
    return (int)hipGetLastError();
}

int launch_lnprob(const DevShared &sh, const LaunchArgs &a, void *stream) {
    if (a.n <= 0) return 0;
    // (the alternative dipole torque, cfg.dipole_torque = 1, lives in the curve kernels only: such a handle runs them for
    // every batch, with or without curve outputs)
    const bool curves = a.ltot || a.lprop || a.ldip || a.mdisc || a.omega || sh.cfg.dipole_torque != 0;
    dim3 grid((unsigned)a.n), block(64);
    // Variants (results agree to rounding); sh.n_simd = SIMDs of the device:
    //  - up to n_simd walkers (one wave per SIMD): four steps per lane (256-step tiles amortise the wavefront scans best;
    //    needs the whole register file of a SIMD);
    //  - beyond: two steps per lane, which keeps two waves resident per SIMD (they fill each other's issue gaps);
    //  - a handle that holds a light curve of more than 64 points runs the LONG builds of the same kernels.
    //  - curve outputs (mode B): by rounds of resident workgroups, kernel_spl_curves (mp_device.h);
    const bool wide = (sh.force_spl ? sh.force_spl : (curves ? kernel_spl_curves(sh, a.n) : kernel_spl(sh, a.n))) == 4;
    const bool lng = sh.has_long != 0;
    hipStream_t st = (hipStream_t)stream;
    //  - launches that would leave SIMDs idle (n <= n_simd / 2): a team of four wavefronts per walker, one step per lane each
    //    (the same 256-step tiles and policy; mode A; LONG builds for handles with longer light curves).  Up to n_simd / 4 walkers every
    //    wavefront has a SIMD of its own; up to n_simd / 2 two share one and fill each other's stalls.  Measured on one box
    //    (profiles/r05_team_*.log): 256 walkers 0.0755 -> 0.0577 ms, 512 walkers 0.0767 -> 0.0699 ms near the truth
    //    (0.197 -> 0.165 ms prior-wide); a team of two (2 steps per lane) at 512 walkers 0.0701 / 0.197 ms.
    const int team = (curves || !a.want_chi2 || a.physical) ? 1 : kernel_waves(sh, a.n);
    if (team > 1) {
        const bool log = a.tile_log != nullptr || kAlwaysLog;
        if (team == 4 && 4 * a.n <= sh.n_simd) {
            if (lng) {
                if (log) hipLaunchKernelGGL((lnprob_team_kernel<1, 4, 1, true, true>), grid, dim3(256), 0, st, sh, a);
                else hipLaunchKernelGGL((lnprob_team_kernel<1, 4, 1, false, true>), grid, dim3(256), 0, st, sh, a);
            } else if (log) hipLaunchKernelGGL((lnprob_team_kernel<1, 4, 1, true>), grid, dim3(256), 0, st, sh, a);
            else hipLaunchKernelGGL((lnprob_team_kernel<1, 4, 1, false>), grid, dim3(256), 0, st, sh, a);
        } else if (team == 4) {
            if (lng) {
                if (log) hipLaunchKernelGGL((lnprob_team_kernel<1, 4, 2, true, true>), grid, dim3(256), 0, st, sh, a);
                else hipLaunchKernelGGL((lnprob_team_kernel<1, 4, 2, false, true>), grid, dim3(256), 0, st, sh, a);
            } else if (log) hipLaunchKernelGGL((lnprob_team_kernel<1, 4, 2, true>), grid, dim3(256), 0, st, sh, a);
            else hipLaunchKernelGGL((lnprob_team_kernel<1, 4, 2, false>), grid, dim3(256), 0, st, sh, a);
        } else {
#ifdef MP_EXPERIMENTS
            if (2 * a.n <= sh.n_simd) {
                if (log) hipLaunchKernelGGL((lnprob_team_kernel<2, 2, 1, true>), grid, dim3(128), 0, st, sh, a);
                else hipLaunchKernelGGL((lnprob_team_kernel<2, 2, 1, false>), grid, dim3(128), 0, st, sh, a);
            } else {
                if (log) hipLaunchKernelGGL((lnprob_team_kernel<2, 2, 2, true>), grid, dim3(128), 0, st, sh, a);
                else hipLaunchKernelGGL((lnprob_team_kernel<2, 2, 2, false>), grid, dim3(128), 0, st, sh, a);
            }
#endif
        }
        return (int)hipGetLastError();
    }
    if (curves) {
        if (wide) hipLaunchKernelGGL((lnprob_kernel<true, 4, false>), grid, block, 0, st, sh, a);
        else hipLaunchKernelGGL((lnprob_kernel<true, 2, false>), grid, block, 0, st, sh, a);
    } else if (a.tile_log != nullptr || kAlwaysLog) {   // diagnostics requested: the builds that record the tile words
        if (wide) {
            if (lng) hipLaunchKernelGGL((lnprob_kernel<false, 4, true, true>), grid, block, 0, st, sh, a);
            else hipLaunchKernelGGL((lnprob_kernel<false, 4, false, true>), grid, block, 0, st, sh, a);
        } else {
            if (lng) hipLaunchKernelGGL((lnprob_kernel<false, 2, true, true>), grid, block, 0, st, sh, a);
            else hipLaunchKernelGGL((lnprob_kernel<false, 2, false, true>), grid, block, 0, st, sh, a);
        }
    } else if (wide) {
        if (lng) hipLaunchKernelGGL((lnprob_kernel<false, 4, true>), grid, block, 0, st, sh, a);
        else hipLaunchKernelGGL((lnprob_kernel<false, 4, false>), grid, block, 0, st, sh, a);
    } else {
        if (lng) hipLaunchKernelGGL((lnprob_kernel<false, 2, true>), grid, block, 0, st, sh, a);
        else hipLaunchKernelGGL((lnprob_kernel<false, 2, false>), grid, block, 0, st, sh, a);
    }
    return (int)hipGetLastError();
}

// n_blocks slots of the active half starting at g.slot_lo
int launch_stretch(const DevShared &sh, const StretchArgs &g, int n_blocks, void *stream) {
    if (n_blocks <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)n_blocks);
    const bool lng = sh.has_long != 0;
    if (stretch_waves(sh, 3 * g.n_half * g.n_ensembles) == 4) {   // small ensembles: a team of four wavefronts per proposal
        if (4 * n_blocks <= sh.n_simd) {
            if (lng) hipLaunchKernelGGL((stretch_kernel<1, true, 4, 1>), grid, dim3(256), 0, st, sh, g);
            else hipLaunchKernelGGL((stretch_kernel<1, false, 4, 1>), grid, dim3(256), 0, st, sh, g);
        } else {
            if (lng) hipLaunchKernelGGL((stretch_kernel<1, true, 4, 2>), grid, dim3(256), 0, st, sh, g);
            else hipLaunchKernelGGL((stretch_kernel<1, false, 4, 2>), grid, dim3(256), 0, st, sh, g);
        }
        return (int)hipGetLastError();
    }
    if ((sh.force_spl ? sh.force_spl : kernel_spl(sh, n_blocks)) == 4) {
        if (lng) hipLaunchKernelGGL((stretch_kernel<4, true>), grid, dim3(64), 0, st, sh, g);
        else hipLaunchKernelGGL((stretch_kernel<4, false>), grid, dim3(64), 0, st, sh, g);
    } else {
        if (lng) hipLaunchKernelGGL((stretch_kernel<2, true>), grid, dim3(64), 0, st, sh, g);
        else hipLaunchKernelGGL((stretch_kernel<2, false>), grid, dim3(64), 0, st, sh, g);
    }
    return (int)hipGetLastError();
}

int launch_stretch_step(const DevShared &sh, const StretchArgs &g, int n_blocks, void *stream) {
    if (n_blocks <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)n_blocks);
    const bool lng = sh.has_long != 0;
    if (stretch_waves(sh, 3 * g.n_half * g.n_ensembles) == 4) {
        if (4 * n_blocks <= sh.n_simd) {
            if (lng) hipLaunchKernelGGL((stretch_step_kernel<1, true, 4, 1>), grid, dim3(256), 0, st, sh, g);
            else hipLaunchKernelGGL((stretch_step_kernel<1, false, 4, 1>), grid, dim3(256), 0, st, sh, g);
        } else {
            if (lng) hipLaunchKernelGGL((stretch_step_kernel<1, true, 4, 2>), grid, dim3(256), 0, st, sh, g);
            else hipLaunchKernelGGL((stretch_step_kernel<1, false, 4, 2>), grid, dim3(256), 0, st, sh, g);
        }
        return (int)hipGetLastError();
    }
    if ((sh.force_spl ? sh.force_spl : kernel_spl(sh, n_blocks)) == 4) {
        if (lng) hipLaunchKernelGGL((stretch_step_kernel<4, true>), grid, dim3(64), 0, st, sh, g);
        else hipLaunchKernelGGL((stretch_step_kernel<4, false>), grid, dim3(64), 0, st, sh, g);
    } else {
        if (lng) hipLaunchKernelGGL((stretch_step_kernel<2, true>), grid, dim3(64), 0, st, sh, g);
        else hipLaunchKernelGGL((stretch_step_kernel<2, false>), grid, dim3(64), 0, st, sh, g);
    }
    return (int)hipGetLastError();
}

int launch_stretch_step_commit(const StretchArgs &g, void *stream) {
    const int n = 2 * g.n_half * g.n_ensembles;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(stretch_step_commit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g);
    return (int)hipGetLastError();
}

int launch_stretch_apply(const StretchArgs &g, void *stream) {
    const int n = g.n_half * g.n_ensembles;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(stretch_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g);
    return (int)hipGetLastError();
}

}  // namespace mp
