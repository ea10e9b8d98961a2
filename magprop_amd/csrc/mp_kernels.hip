// mp_kernels.hip — gfx950 (CDNA4) kernels of the magnetar log-posterior hot path.
//
// Layout: ONE WALKER PER WAVEFRONT, ONE TIME STEP PER LANE.  The 10 000 grid intervals are
// processed in tiles of 64 consecutive steps; inside a tile the 64 lanes advance all 64 steps
// at once (parallel in time).  Scheme = exponential Adams-Moulton of order 4 on the geometric
// output grid (serial restatement: oracle/mp_oracle.c mpo_trajectory):
//
//   1. Mdisc obeys dM/dt = Mdotfb(t) - M/tvisc (linear, omega-independent; reference RHS
//      code/synthetic_datasets/funcs.py:122-129, magnetar/funcs.py:86-92).  Every lane evaluates
//      Mdotfb at its step end, fetches the three previous values from its neighbours (DPP), and builds
//      the affine map M_{j+1} = e^{-h/tvisc} M_j + b_j of its own step; a wavefront scan of affine
//      maps yields Mdisc at all 64 step ends.
//   2. omega obeys a scalar nonlinear ODE fed by Mdisc(t).  Every lane evaluates omega_dot and its
//      Jacobian lambda ONCE per sweep, at its current guess of omega at its step end, fetches
//      (omega_dot, omega) of the three previous grid points from its neighbours, and forms the step
//      map omega_{j+1} = e^{h lambda} omega_j + h sum_m phi_{m+1}(h lambda) g_m.  A second affine
//      scan propagates the tile's start value through all 64 linearised maps.  This Newton-type sweep
//      converges quadratically (about 2 sweeps from an extrapolated guess), terminates in at most
//      ~64 sweeps in any case, and reproduces the serial recurrence to rounding.
//   3. Each lane evaluates the luminosity at its step end (reference luminosity stage,
//      code/synthetic_datasets/funcs.py:175-229, magnetar/funcs.py:157-210), the tile's light curve
//      is staged in LDS, the observations that fall in the tile are interpolated from LDS
//      (np.interp semantics) and accumulated into per-lane chi^2 partial sums; optional coalesced
//      512-B-per-wave stores write the model light curve to HBM.
//   4. A wavefront reduction gives -0.5*chi^2 (code/synthetic_datasets/mcmc_eqns.py:25).
//
// No MFMA (no dense contraction anywhere on this path), fp64 throughout; the kernel is bound by the
// fp64 VALU issue rate of one wave per SIMD (profiles/).  The arithmetic is algebraically simplified
// with respect to the reference formulas (e.g. fastness w = (Rm/Rc)^1.5 = omega*Rm^1.5/sqrt(GM),
// eta1-eta2 = -tanh); oracle/mp_oracle.c keeps the literal formulas and the tests compare the two.
#include <hip/hip_runtime.h>

#include <cmath>

#include "mp_device.h"

namespace mp {

#define MP_DEV __device__ __forceinline__

// ---------------------------------------------------------------- wavefront helpers (DPP, no LDS)
// DPP controls (GFX9 encoding): row_shr:n = 0x110+n, wave_shr:1 = 0x138, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK>
MP_DEV double dpp_move(double keep, double src) {
    // lanes with a valid DPP source (and enabled by ROW_MASK) receive src from that lane, all others `keep`
    const int klo = __double2loint(keep), khi = __double2hiint(keep);
    const int slo = __double2loint(src), shi = __double2hiint(src);
    const int lo = __builtin_amdgcn_update_dpp(klo, slo, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(khi, shi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}

// value of lane-1 (lane 0 receives `first`)
MP_DEV double lane_prev(double v, double first) { return dpp_move<0x138, 0xF>(first, v); }

// broadcast lane `src` (wave-uniform index) to all lanes
MP_DEV double lane_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// Inclusive scan of affine maps x -> a*x + b over the 64 lanes: afterwards lane l holds
// m_l o m_{l-1} o ... o m_0.  Lanes without a DPP source combine with the identity (1, 0).
template <int CTRL, int ROW_MASK>
MP_DEV void scan_step(double &a, double &b) {
    const double pa = dpp_move<CTRL, ROW_MASK>(1.0, a);
    const double pb = dpp_move<CTRL, ROW_MASK>(0.0, b);
    b = fma(a, pb, b);
    a = a * pa;
}

MP_DEV void scan_affine(double &a, double &b) {
    scan_step<0x111, 0xF>(a, b);  // row_shr:1
    scan_step<0x112, 0xF>(a, b);  // row_shr:2
    scan_step<0x114, 0xF>(a, b);  // row_shr:4
    scan_step<0x118, 0xF>(a, b);  // row_shr:8   -> every 16-lane row scanned
    scan_step<0x142, 0xA>(a, b);  // row_bcast:15 into rows 1 and 3
    scan_step<0x143, 0xC>(a, b);  // row_bcast:31 into rows 2 and 3
}

MP_DEV double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ---------------------------------------------------------------- fp64 elementary functions
// Three-address FMA for Horner chains.  hipcc selects the two-address v_fmac_f64 there and then has to
// copy every polynomial coefficient into the accumulator first (one v_mov_b64 per term); the explicit
// v_fma_f64 reads the coefficient in place.  Only plain VALU results feed it (no transcendental-op hazard).
MP_DEV double fma3(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// Hand-rolled for this kernel's argument ranges (positive, normal, far from overflow): hardware
// seed (v_rcp_f64 / v_rsq_f64, ~2^-23) + two Newton steps, without the scaling / fix-up code the
// general-purpose library versions carry.  All are accurate to ~1-2 ulp.
MP_DEV double rcp_fast(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

MP_DEV double rsqrt_fast(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    return y;
}

// e^x for x in [-750, 700]; underflows cleanly to 0 below
MP_DEV double exp_fast(double x) {
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 1.0 / 479001600.0;                 // Taylor degree 12 on |r| <= ln2/2: 1.7e-16
    p = fma3(p, r, 1.0 / 39916800.0);
    p = fma3(p, r, 1.0 / 3628800.0);
    p = fma3(p, r, 1.0 / 362880.0);
    p = fma3(p, r, 1.0 / 40320.0);
    p = fma3(p, r, 1.0 / 5040.0);
    p = fma3(p, r, 1.0 / 720.0);
    p = fma3(p, r, 1.0 / 120.0);
    p = fma3(p, r, 1.0 / 24.0);
    p = fma3(p, r, 1.0 / 6.0);
    p = fma3(p, r, 0.5);
    p = fma3(p, r, 1.0);
    p = fma3(p, r, 1.0);
    return ldexp(p, (int)k);
}

// x^(-1/3) for positive normal x within float range: v_log_f32/v_exp_f32 seed (~1e-6) + two Newton steps
// y <- y (4 - x y^3)/3 (error -> 2 e^2): ~1 ulp.
MP_DEV double rcbrt_fast(double x) {
    double y = (double)__builtin_amdgcn_exp2f(-0.33333333f * __builtin_amdgcn_logf((float)x));
    const double x3 = x * (1.0 / 3.0);
    double y3 = y * y * y;
    y = y * fma(-x3, y3, 4.0 / 3.0);
    y3 = y * y * y;
    y = y * fma(-x3, y3, 4.0 / 3.0);
    return y;
}

// x^(-2/7) for positive normal x within float range: y = (x^2)^(-1/7), Newton y <- y (8 - x^2 y^7)/7 (error -> 4 e^2)
MP_DEV double pow_m2_7_fast(double x) {
    double y = (double)__builtin_amdgcn_exp2f(-0.28571429f * __builtin_amdgcn_logf((float)x));
    const double z7 = (x * x) * (1.0 / 7.0);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double y2 = y * y, y4 = y2 * y2;
        const double y7 = (y4 * y2) * y;
        y = y * fma(-z7, y7, 8.0 / 7.0);
    }
    return y;
}

// natural log of a positive normal number
MP_DEV double log_fast(double x) {
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);    // [0.5, 1)
    const bool lt = m < 0.70710678118654752440;
    m = lt ? 2.0 * m : m;                         // [sqrt(1/2), sqrt(2))
    e = lt ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * rcp_fast(2.0 + f);       // |s| <= 0.1716
    const double z = s * s;
    double p = 1.0 / 21.0;                        // atanh series: log(m) = 2s(1 + z/3 + z^2/5 + ...)
    p = fma3(p, z, 1.0 / 19.0);
    p = fma3(p, z, 1.0 / 17.0);
    p = fma3(p, z, 1.0 / 15.0);
    p = fma3(p, z, 1.0 / 13.0);
    p = fma3(p, z, 1.0 / 11.0);
    p = fma3(p, z, 1.0 / 9.0);
    p = fma3(p, z, 1.0 / 7.0);
    p = fma3(p, z, 1.0 / 5.0);
    p = fma3(p, z, 1.0 / 3.0);
    p = fma3(p, z, 1.0);
    return fma((double)e, 6.93147180559945309417e-01, 2.0 * s * p);
}

// ---------------------------------------------------------------- phi functions
// phi_j(z) = sum_k z^k/(k+j)!  : phi_1 = (e^z-1)/z, phi_{j+1} = (phi_j - 1/j!)/z
struct Phi {
    double e, p1, p2, p3, p4;
};

MP_DEV Phi phi1234(double z) {
    // Taylor series of phi_4 for |z| < 1/2 (13 terms: < 2e-17 relative), closed forms elsewhere
    double s = 1.0 / 20922789888000.0;            // 1/16!
    s = fma3(s, z, 1.0 / 1307674368000.0);         // 1/15!
    s = fma3(s, z, 1.0 / 87178291200.0);           // 1/14!
    s = fma3(s, z, 1.0 / 6227020800.0);            // 1/13!
    s = fma3(s, z, 1.0 / 479001600.0);             // 1/12!
    s = fma3(s, z, 1.0 / 39916800.0);              // 1/11!
    s = fma3(s, z, 1.0 / 3628800.0);               // 1/10!
    s = fma3(s, z, 1.0 / 362880.0);                // 1/9!
    s = fma3(s, z, 1.0 / 40320.0);                 // 1/8!
    s = fma3(s, z, 1.0 / 5040.0);                  // 1/7!
    s = fma3(s, z, 1.0 / 720.0);                   // 1/6!
    s = fma3(s, z, 1.0 / 120.0);                   // 1/5!
    s = fma3(s, z, 1.0 / 24.0);                    // 1/4!
    Phi r;
    r.p4 = s;
    r.p3 = fma(z, s, 1.0 / 6.0);
    r.p2 = fma(z, r.p3, 0.5);
    r.p1 = fma(z, r.p2, 1.0);
    r.e = fma(z, r.p1, 1.0);
    const bool big = !(fabs(z) < 0.5);
    if (__any(big)) {                              // wave-uniform: only stiff / late-time tiles pay for this
        const double ce = exp_fast(fmax(z, -750.0));
        const double rz = rcp_fast(big ? z : 1.0);
        const double c1 = (ce - 1.0) * rz;
        const double c2 = (c1 - 1.0) * rz;
        const double c3 = (c2 - 0.5) * rz;
        const double c4 = (c3 - 1.0 / 6.0) * rz;
        r.e = big ? ce : r.e;
        r.p1 = big ? c1 : r.p1;
        r.p2 = big ? c2 : r.p2;
        r.p3 = big ? c3 : r.p3;
        r.p4 = big ? c4 : r.p4;
    }
    return r;
}

// h * int_0^1 e^{z(1-theta)} P(theta) dtheta for the cubic P through the node values v0..v3 at
// t_{j+1}, t_j, t_{j-1}, t_{j-2} (quadrature matrix W of the geometric grid, DevShared::eamW)
MP_DEV double eam4_increment(const DevShared &sh, const Phi &p, double h, double v0, double v1, double v2, double v3) {
    double acc = 0.0;
    const double ph[4] = {p.p1, p.p2, p.p3, p.p4};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const double g = fma(sh.eamW[0][m], v0, fma(sh.eamW[1][m], v1, fma(sh.eamW[2][m], v2, sh.eamW[3][m] * v3)));
        acc = fma(ph[m], g, acc);
    }
    return h * acc;
}

// ---------------------------------------------------------------- per-walker constants
struct Walker {
    double inv_tau;   // 1/tvisc
    double S_amp;     // M0/tfb
    double inv_tfb;   // 1/tfb
    double Crm;       // mu^(4/7) GM^(-1/7) f_Rm^(-2/7)
    double DI;        // mu^2/(6 c^3 I)          dipole torque / I = -DI*omega^3
    double D;         // mu^2/(6 c^3)
    double armI;      // sqrt(GM)/I
    double kc;        // k*c
    double sqrt_kc;   // sqrt(k*c)
    double Kc;        // (k*c)^1.5 / sqrt(GM): fastness of a capped Alfven radius = Kc/sqrt(omega)
    double dipeff, propeff, f_beam;
};

// What the omega equation needs to know about the disc at one time point (all omega-independent,
// computed once per tile in the time-parallel Mdisc phase).
struct DiscPt {
    double mdot;  // Mdisc/tvisc
    double rmu;   // uncapped Alfven radius, code/synthetic_datasets/funcs.py:105-106
    double squ;   // sqrt(rmu)
    double qu;    // rmu^1.5/sqrt(GM): uncapped fastness = omega*qu
};

MP_DEV DiscPt disc_point(const DevShared &sh, const Walker &w, double Mdisc) {
    DiscPt p;
    p.mdot = Mdisc * w.inv_tau;
    p.rmu = w.Crm * pow_m2_7_fast(p.mdot);
    p.squ = p.rmu * rsqrt_fast(p.rmu);
    p.qu = p.rmu * p.squ * sh.inv_sqrtGM;
    return p;
}

MP_DEV DiscPt disc_prev(const DiscPt &p, const DiscPt &first) {  // the previous lane's point (lane 0: `first`)
    DiscPt r;
    r.mdot = lane_prev(p.mdot, first.mdot);
    r.rmu = lane_prev(p.rmu, first.rmu);
    r.squ = lane_prev(p.squ, first.squ);
    r.qu = lane_prev(p.qu, first.qu);
    return r;
}

MP_DEV DiscPt disc_bcast(const DiscPt &p, int src) {
    DiscPt r;
    r.mdot = lane_bcast(p.mdot, src);
    r.rmu = lane_bcast(p.rmu, src);
    r.squ = lane_bcast(p.squ, src);
    r.qu = lane_bcast(p.qu, src);
    return r;
}

// fallback accretion rate Mdotfb(t), code/synthetic_datasets/funcs.py:128
MP_DEV double mdot_fb(const Walker &w, double t) {
    const double u = fma(t, w.inv_tfb, 1.0);                        // (t + tfb)/tfb >= 1
    const double r = rcbrt_fast(u), r2 = r * r;
    return w.S_amp * (r2 * r2 * r);                                 // u^(-5/3)
}

// Radii / fastness / switch shared by the ODE right-hand side and the luminosity stage
// (code/synthetic_datasets/funcs.py:105-123 / magnetar/funcs.py:64-84 in simplified algebra):
//   Rm = min(rmu, k c/omega);  fastness = (Rm/Rc)^1.5 = omega Rm^1.5/sqrt(GM);  tanh(n (fastness-1)).
struct Flow {
    double inv_om, Rm, sq, fast, e, r, th;
    bool capped, big;
};

MP_DEV Flow flow_state(const Walker &w, double n, const DiscPt &p, double om) {
    Flow f;
    const double y = rsqrt_fast(om);
    f.inv_om = y * y;
    const double rlc = w.kc * f.inv_om;
    f.capped = p.rmu >= rlc;                                        // Rm >= k*Rlc -> Rm = k*Rlc
    f.Rm = f.capped ? rlc : p.rmu;
    f.sq = f.capped ? w.sqrt_kc * y : p.squ;                        // sqrt(Rm)
    f.fast = f.capped ? w.Kc * y : om * p.qu;
    const double x = fma(n, f.fast, -n);
    f.e = exp_fast(fmax(-2.0 * fabs(x), -750.0));
    f.r = rcp_fast(1.0 + f.e);
    f.th = copysign((1.0 - f.e) * f.r, x);                          // tanh(x) = eta2 - eta1
    f.big = f.Rm >= kR;
    return f;
}

// d(omega)/dt, code/synthetic_datasets/funcs.py:119,131-140; lam = d(omega_dot)/d(omega)
template <bool WANT_LAM>
MP_DEV double omega_rhs(const DevShared &sh, const Walker &w, const DiscPt &p, double om, double &rot, double &lam) {
    const Flow f = flow_state(w, sh.cfg.n_ode, p, om);
    const double om2 = om * om;
    rot = sh.crot * om2;
    const bool brk = rot > 0.27;                                    // break-up: Nacc = 0
    const double arm = w.armI * (f.big ? f.sq : sh.sqrtR);          // sqrt(GM*max(Rm,R))/I
    const double nacc = brk ? 0.0 : -arm * p.mdot * f.th;           // Nacc/I ; Macc - Mprop = -tanh * mdot
    if (WANT_LAM) {
        const double dfast = (f.capped ? -0.5 : 1.0) * f.fast * f.inv_om;
        const double dth = sh.cfg.n_ode * (4.0 * f.e * f.r * f.r) * dfast;   // n sech^2 dfast
        const double darm = (f.capped && f.big) ? -0.5 * arm * f.inv_om : 0.0;
        const double dn = brk ? 0.0 : -p.mdot * fma(darm, f.th, arm * dth);
        lam = fma(-3.0 * w.DI, om2, dn);
    }
    return fma(-w.DI * om2, om, nacc);
}

// luminosities (erg/s) at one grid point, reference luminosity stage
// (code/synthetic_datasets/funcs.py:204-229, magnetar/funcs.py:191-210)
MP_DEV void luminosity(const DevShared &sh, const Walker &w, const DiscPt &p, double om, double &Ltot,
                       double &Lprop, double &Ldip) {
    const Flow f = flow_state(w, sh.cfg.n_lum, p, om);
    const double eta2 = f.th >= 0.0 ? f.r : f.e * f.r;              // 0.5*(1 + tanh x)
    const double om2 = om * om;
    const double rot = sh.crot * om2;
    const double arm = sh.sqrtGM * (f.big ? f.sq : sh.sqrtR);
    const double Nacc = rot > sh.cfg.nacc_lum_threshold ? 0.0 : -arm * p.mdot * f.th;
    double ld = w.dipeff * (w.D * om2 * om2);
    if (ld <= 0.0) ld = 0.0;
    if (!isfinite(ld)) ld = 0.0;
    double lp = -Nacc * om;
    if (sh.cfg.lprop_gm_term) lp -= sh.GM * rcp_fast(f.Rm) * eta2 * p.mdot;
    lp *= w.propeff;
    if (lp <= 0.0) lp = 0.0;
    if (!isfinite(lp)) lp = 0.0;
    Ltot = w.f_beam * (ld + lp);
    Lprop = lp;
    Ldip = ld;
}

constexpr int kMaxSweepsMargin = 16;  // sweeps allowed beyond the tile length (after which every step is exact)
constexpr double kSweepTol = 1e-9;  // relative change of the step-end values that ends the sweeps

// ---------------------------------------------------------------- the kernel
// SPL = consecutive steps owned by one lane; a tile is 64*SPL steps.
template <bool CURVES, int SPL>
__global__ __launch_bounds__(64, 1) void lnprob_kernel(const DevShared sh, const LaunchArgs a) {
    constexpr int kSPL = SPL, kTile = 64 * SPL, kMaxSweeps = kTile + kMaxSweepsMargin;
    const int n_tiles = (sh.n_grid - 1 + kTile - 1) / kTile;
    const int walker = blockIdx.x;
    const int lane = threadIdx.x;
    __shared__ double Lbuf[kTile + 1];

    const int n_grid = sh.n_grid;
    const int nsteps = n_grid - 1;
    const size_t row = (size_t)walker * (size_t)n_grid;

    // ---- parameters, prior, un-logging (code/synthetic_datasets/mcmc_eqns.py:16-17,28-49)
    double par[MP_MAX_NDIM];
    const double *pw = a.pars + (size_t)walker * a.ndim;
#pragma unroll
    for (int i = 0; i < MP_MAX_NDIM; ++i) par[i] = i < a.ndim ? pw[i] : 0.0;

    int status = MP_STATUS_OK;
    if (!a.physical) {
        bool outside = false;
#pragma unroll
        for (int i = 0; i < MP_MAX_NDIM; ++i)
            if (i < sh.n_prior && (!(par[i] >= sh.lower[i]) || !(par[i] <= sh.upper[i]))) outside = true;
        if (outside) status = MP_STATUS_PRIOR;
#pragma unroll
        for (int i = 0; i < MP_MAX_NDIM; ++i)
            if (i < a.ndim && ((sh.log_mask >> i) & 1u)) par[i] = pow(10.0, par[i]);
    }

    // ---- walker constants (code/synthetic_datasets/funcs.py:98-102)
    Walker w;
    {
        const double B = par[0], MdiscI = par[2], RdiscI = par[3], epsilon = par[4], delta = par[5];
        const double tau = (RdiscI * 1.0e5) / (sh.cfg.alpha * sh.cfg.cs7 * 1.0e7);
        const double mu = 1.0e15 * B * (kR * kR * kR);
        const double M0 = delta * MdiscI * kMsol;
        const double tfb = epsilon * tau;
        w.inv_tau = 1.0 / tau;
        w.S_amp = M0 / tfb;
        w.inv_tfb = 1.0 / tfb;
        w.Crm = pow(mu, 4.0 / 7.0) * pow(sh.GM, -1.0 / 7.0) * pow(sh.cfg.rm_massflow_factor, -2.0 / 7.0);
        w.D = (mu * mu) / (6.0 * kC * kC * kC);
        w.DI = w.D * sh.inv_inertia;
        w.armI = sh.sqrtGM * sh.inv_inertia;
        w.kc = sh.cfg.k * kC;
        w.sqrt_kc = sqrt(w.kc);
        w.Kc = w.kc * w.sqrt_kc * sh.inv_sqrtGM;
        w.dipeff = sh.cfg.dipeff;
        w.propeff = sh.cfg.propeff;
        w.f_beam = sh.cfg.f_beam;
        // 7/8/9-parameter likelihoods, magnetar/mcmc_eqns.py:22-34
        if (a.ndim == 7) w.f_beam = par[6];
        if (a.ndim == 8) { w.dipeff = par[6]; w.propeff = par[7]; }
        if (a.ndim == 9) { w.dipeff = par[6]; w.propeff = par[7]; w.f_beam = par[8]; }
    }

    // ---- state carried from tile to tile (all wave-uniform).  Index 0 = the tile's start point P0,
    // 1 = P0-1, 2 = P0-2: the history the multistep formulas reach back to.
    const double t0 = sh.tgrid[0];
    double t_s = t0;
    double M_s = par[2] * kMsol;                         // initial conditions, code/synthetic_datasets/funcs.py:66-69
    double om_s = (2.0 * M_PI) / (1.0e-3 * par[1]);
    double cS0 = mdot_fb(w, t0), cS1 = mdot_fb(w, t0 * sh.inv_q), cS2 = mdot_fb(w, t0 * sh.inv_q * sh.inv_q);
    double cf0, cf1, cf2, cw1 = om_s, cw2 = om_s, cw3 = om_s, cw4 = om_s;   // (omega_dot, omega) history; cw0 == om_s (cw3, cw4: predictor only)
    double L_s, Lp_s, Ld_s;
    {
        const DiscPt d_s = disc_point(sh, w, M_s);
        double rot0, dummy;
        cf0 = omega_rhs<false>(sh, w, d_s, om_s, rot0, dummy);
        cf1 = cf2 = cf0;
        if (status == MP_STATUS_OK) {
            if (!(isfinite(M_s) && isfinite(om_s)) || M_s <= 0.0 || om_s <= 0.0) status = MP_STATUS_NONFINITE;
            else if (rot0 > 0.27) status = MP_STATUS_FLAG;
        }
        luminosity(sh, w, d_s, om_s, L_s, Lp_s, Ld_s);
    }

    const int dsid = a.ds_id ? a.ds_id[walker] : 0;
    const DsDesc dsd = sh.ds ? sh.ds[(dsid >= 0 && dsid < sh.n_ds) ? dsid : 0] : DsDesc{0, 0, 0, 0};
    const int32_t *tptr = sh.tile_ptr + dsd.tile_off;
    // The first 64 observations of the walker's light curve live in registers, one per lane (time-sorted;
    // every synthetic set has 50).  Longer light curves take the tile-bucketed global-memory path for the rest.
    int ob_g = -1;
    double ob_dx = 0.0, ob_idt = 0.0, ob_y = 0.0, ob_ye = 1.0;
    if (a.want_chi2 && lane < dsd.n_obs) {
        const int jj = dsd.obs_off + lane;
        ob_g = sh.obs_g[jj];
        ob_dx = sh.obs_dx[jj];
        ob_idt = sh.obs_idt[jj];
        ob_y = sh.obs_y[jj];
        ob_ye = sh.obs_yerr[jj];
    }
    const int ob_tile = ob_g >= 0 ? ob_g / kTile : -1;
    const bool long_lc = a.want_chi2 && dsd.n_obs > 64;
    double chi = 0.0;
    int sweeps_total = 0;

    if (status == MP_STATUS_OK) {
        if (CURVES && lane == 0) {
            if (a.ltot) a.ltot[row] = L_s / 1.0e50;
            if (a.lprop) a.lprop[row] = Lp_s / 1.0e50;
            if (a.ldip) a.ldip[row] = Ld_s / 1.0e50;
            if (a.mdisc) a.mdisc[row] = M_s;
            if (a.omega) a.omega[row] = om_s;
        }
        // Each lane owns kSPL consecutive steps of the tile: steps tile*kTile + lane*kSPL + s, s = 0..kSPL-1.
        // Step end times are fetched one tile ahead of their use.
        double tb_next[kSPL];
#pragma unroll
        for (int s = 0; s < kSPL; ++s) tb_next[s] = sh.tgrid[min(lane * kSPL + s + 1, nsteps)];

        for (int tile = 0; tile < n_tiles; ++tile) {
            const int i0 = tile * kTile + lane * kSPL;   // this lane's first step: tgrid[i0] -> tgrid[i0+1]
            double tb[kSPL], h[kSPL];
            bool active[kSPL];
#pragma unroll
            for (int s = 0; s < kSPL; ++s) {
                tb[s] = tb_next[s];
                tb_next[s] = sh.tgrid[min(i0 + kTile + s + 1, nsteps)];
                active[s] = i0 + s < nsteps;
            }
            {
                const double ta0 = lane_prev(tb[kSPL - 1], t_s);
#pragma unroll
                for (int s = 0; s < kSPL; ++s) h[s] = active[s] ? tb[s] - (s == 0 ? ta0 : tb[s - 1]) : 1.0;
            }

            // ---------------- Mdisc: exponential Adams-Moulton step (explicit: the source is known) + affine scan.
            // E*[k]: values at the three grid points before this lane's first step (k = 0,1,2) and at its step ends (k = 3+s).
            double M1[kSPL];
            double ES[kSPL + 3];
            {
#pragma unroll
                for (int s = 0; s < kSPL; ++s) ES[3 + s] = mdot_fb(w, tb[s]);
                ES[2] = lane_prev(ES[kSPL + 2], cS0);
                ES[1] = lane_prev(ES[kSPL + 1], cS1);
                ES[0] = lane_prev(ES[kSPL + 0], cS2);
                double am[kSPL], bm[kSPL];
                double A = 1.0, B = 0.0;                  // composition of this lane's step maps
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    const Phi pm = phi1234(-h[s] * w.inv_tau);
                    am[s] = active[s] ? pm.e : 1.0;
                    bm[s] = active[s] ? eam4_increment(sh, pm, h[s], ES[3 + s], ES[2 + s], ES[1 + s], ES[s]) : 0.0;
                    B = fma(am[s], B, bm[s]);
                    A = A * am[s];
                }
                scan_affine(A, B);
                const double Ax = lane_prev(A, 1.0), Bx = lane_prev(B, 0.0);   // exclusive prefix
                double Mc = fma(Ax, M_s, Bx);            // Mdisc at this lane's first step start
#pragma unroll
                for (int s = 0; s < kSPL; ++s) { Mc = fma(am[s], Mc, bm[s]); M1[s] = Mc; }
            }
            DiscPt d1[kSPL];
#pragma unroll
            for (int s = 0; s < kSPL; ++s) d1[s] = disc_point(sh, w, M1[s]);

            // ---------------- omega: predictor = quadratic extrapolation of the last three grid values in the
            // step index (the grid is logarithmic, so power laws are smooth in the index) ...
            double wg[kSPL];                              // current guess of omega at this lane's step ends
            {
                // Newton backward-difference extrapolation (cubic once four grid values exist)
                const double g1 = om_s - cw1, g2 = g1 - (cw1 - cw2);
                const double d2b = (cw1 - cw2) - (cw2 - cw3);
                const double g3 = tile == 0 ? 0.0 : g2 - d2b;
                const double g4 = tile == 0 ? 0.0 : g3 - (d2b - ((cw2 - cw3) - (cw3 - cw4)));
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    const double k = (double)(lane * kSPL + s + 1);
                    const double c2 = 0.5 * k * (k + 1.0);
                    const double c3 = c2 * (k + 2.0) * (1.0 / 3.0);
                    wg[s] = fma(k, g1, fma(c2, g2, fma(c3, g3, fma(c3 * (k + 3.0) * 0.25, g4, om_s))));
                }
            }
            // ... then Newton-type sweeps of the linearised step maps
            double Ef[kSPL + 3], Ew[kSPL + 3];
            unsigned long long flagged = 0ull, pending = ~0ull;
            bool settled = false;    // this lane's guesses moved by < 1e-3 in the previous sweep
            int sweep = 0;
            Ew[2] = om_s;
            while (true) {
                ++sweep;
                double lam[kSPL];
                bool flg = false;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    if (!(wg[s] > 0.0)) wg[s] = Ew[2] > 0.0 ? Ew[2] : om_s;   // keep the iteration alive after a wild or NaN guess
                    double rot;
                    Ef[3 + s] = omega_rhs<true>(sh, w, d1[s], wg[s], rot, lam[s]);
                    Ew[3 + s] = wg[s];
                    flg = flg || (active[s] && rot > 0.27);
                }
                // break-up reached by an iterate that is no longer a wild guess: the reference's 'flag'
                flagged |= __ballot(settled && flg);
                double h1 = cf1, h2 = cf2, u1 = cw1, u2 = cw2;
                if (tile == 0) {   // start-up: the two points before the grid continue points 0 and 1 linearly in the index
                    const double fp1 = lane_bcast(Ef[3], 0), wp1 = lane_bcast(Ew[3], 0);
                    h1 = 2.0 * cf0 - fp1; u1 = 2.0 * om_s - wp1;
                    h2 = 3.0 * cf0 - 2.0 * fp1; u2 = 3.0 * om_s - 2.0 * wp1;
                }
                Ef[2] = lane_prev(Ef[kSPL + 2], cf0);  Ew[2] = lane_prev(Ew[kSPL + 2], om_s);
                Ef[1] = lane_prev(Ef[kSPL + 1], h1);   Ew[1] = lane_prev(Ew[kSPL + 1], u1);
                Ef[0] = lane_prev(Ef[kSPL + 0], h2);   Ew[0] = lane_prev(Ew[kSPL + 0], u2);
                double aw[kSPL], bw[kSPL];
                double A = 1.0, B = 0.0;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    const Phi pw_ = phi1234(h[s] * lam[s]);
                    aw[s] = active[s] ? pw_.e : 1.0;
                    bw[s] = active[s] ? eam4_increment(sh, pw_, h[s], fma(-lam[s], Ew[3 + s], Ef[3 + s]),
                                                       fma(-lam[s], Ew[2 + s], Ef[2 + s]), fma(-lam[s], Ew[1 + s], Ef[1 + s]),
                                                       fma(-lam[s], Ew[s], Ef[s]))
                                      : 0.0;
                    B = fma(aw[s], B, bw[s]);
                    A = A * aw[s];
                }
                scan_affine(A, B);
                const double Ax = lane_prev(A, 1.0), Bx = lane_prev(B, 0.0);
                double wc = fma(Ax, om_s, Bx);           // omega at this lane's first step start
                bool all_ok = true, all_settled = true;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    wc = fma(aw[s], wc, bw[s]);
                    const double dw = fabs(wc - wg[s]), mag = fabs(wc);
                    all_settled = all_settled && (dw <= 1.0e-3 * mag);               // false for NaN
                    all_ok = all_ok && (!active[s] || dw <= kSweepTol * mag);
                    wg[s] = wc;
                }
                settled = all_settled;
                pending = __ballot(!all_ok);
                if (pending == 0ull || flagged != 0ull || sweep >= kMaxSweeps) break;
            }
            sweeps_total += sweep;

            // ---------------- failure detection in time order (oracle/mp_oracle.c)
            {
                bool bad = false, over = false;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    bad = bad || (active[s] && (!(isfinite(M1[s]) && isfinite(wg[s])) || M1[s] <= 0.0 || wg[s] <= 0.0));
                    over = over || (active[s] && sh.crot * wg[s] * wg[s] > 0.27);
                }
                const unsigned long long mb = __ballot(bad);
                // a step whose sweeps never settle is chattering on the Nacc discontinuity: same verdict as a flag
                const unsigned long long mf = flagged | __ballot(over) | (flagged ? 0ull : pending);
                if (mb | mf) {
                    const int first = __ffsll((unsigned long long)(mb | mf)) - 1;
                    status = ((mf >> first) & 1ull) ? MP_STATUS_FLAG : MP_STATUS_NONFINITE;
                    break;
                }
            }

            // ---------------- luminosity at the step ends, light curve through LDS, chi^2
            double Lt[kSPL];
#pragma unroll
            for (int s = 0; s < kSPL; ++s) {
                double Lp, Ld;
                luminosity(sh, w, d1[s], wg[s], Lt[s], Lp, Ld);
                if (CURVES && active[s]) {
                    const size_t o = row + (size_t)(i0 + s) + 1;
                    if (a.ltot) a.ltot[o] = Lt[s] / 1.0e50;
                    if (a.lprop) a.lprop[o] = Lp / 1.0e50;
                    if (a.ldip) a.ldip[o] = Ld / 1.0e50;
                    if (a.mdisc) a.mdisc[o] = M1[s];
                    if (a.omega) a.omega[o] = wg[s];
                }
            }
            {
                const bool mine = ob_tile == tile;
                int j0 = 0, j1 = 0;
                if (long_lc) { j0 = max(tptr[tile * kSPL], 64); j1 = tptr[min((tile + 1) * kSPL, sh.n_tiles)]; }   // 64-step buckets
                if (__any(mine) || j1 > j0) {
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) Lbuf[lane * kSPL + s + 1] = Lt[s];
                    if (lane == 0) Lbuf[0] = L_s;
                    __syncthreads();
                    if (mine) {
                        const int g = ob_g - tile * kTile;
                        const double La = Lbuf[g], Lb = Lbuf[g + 1];
                        const double mod = fma((Lb - La) * ob_idt, ob_dx, La) / 1.0e50;   // np.interp, then /1e50
                        const double res = (ob_y - mod) / ob_ye;
                        chi = fma(res, res, chi);
                    }
                    for (int j = j0 + lane; j < j1; j += 64) {
                        const int jj = dsd.obs_off + j;
                        const int g = sh.obs_g[jj] - tile * kTile;
                        const double La = Lbuf[g], Lb = Lbuf[g + 1];
                        const double mod = fma((Lb - La) * sh.obs_idt[jj], sh.obs_dx[jj], La) / 1.0e50;
                        const double res = (sh.obs_y[jj] - mod) / sh.obs_yerr[jj];
                        chi = fma(res, res, chi);
                    }
                    __syncthreads();
                }
            }

            // ---------------- carry the tile end (and the history behind it) to the next tile: only full tiles
            // have a successor, so the sources are the last three step ends of lane 63
            if (tile + 1 < n_tiles) {
                // step end number e of the tile (0-based) lives in lane e / kSPL, slot e % kSPL
                constexpr int e1 = kTile - 2, e2 = kTile - 3, e3 = kTile - 4, e4 = kTile - 5;
                cS0 = lane_bcast(ES[3 + kSPL - 1], 63);            cf0 = lane_bcast(Ef[3 + kSPL - 1], 63);
                cS1 = lane_bcast(ES[3 + e1 % kSPL], e1 / kSPL);    cf1 = lane_bcast(Ef[3 + e1 % kSPL], e1 / kSPL);
                cS2 = lane_bcast(ES[3 + e2 % kSPL], e2 / kSPL);    cf2 = lane_bcast(Ef[3 + e2 % kSPL], e2 / kSPL);
                cw1 = lane_bcast(wg[e1 % kSPL], e1 / kSPL);
                cw2 = lane_bcast(wg[e2 % kSPL], e2 / kSPL);
                cw3 = lane_bcast(wg[e3 % kSPL], e3 / kSPL);
                cw4 = lane_bcast(wg[e4 % kSPL], e4 / kSPL);
                t_s = lane_bcast(tb[kSPL - 1], 63);
                M_s = lane_bcast(M1[kSPL - 1], 63);
                om_s = lane_bcast(wg[kSPL - 1], 63);
                L_s = lane_bcast(Lt[kSPL - 1], 63);
            }
        }
    }

    double lnp = -INFINITY;
    if (status == MP_STATUS_OK) {
        lnp = -0.5 * wave_sum(chi);
        if (!isfinite(lnp)) { lnp = -INFINITY; status = MP_STATUS_NONFINITE; }
    }
    if (lane == 0) {
        a.lnprob[walker] = lnp;
        if (a.status) a.status[walker] = status;
        if (a.sweeps) a.sweeps[walker] = sweeps_total;
    }
}

int launch_lnprob(const DevShared &sh, const LaunchArgs &a, void *stream) {
    if (a.n <= 0) return 0;
    const bool curves = a.ltot || a.lprop || a.ldip || a.mdisc || a.omega;
    dim3 grid((unsigned)a.n), block(64);
    // Four steps per lane (256-step tiles) amortise the wavefront scans best and are the fastest variant while
    // every walker can have a SIMD to itself (256 CUs x 4 SIMDs); it needs > 256 VGPRs-worth of state per two
    // waves, so larger batches use two steps per lane, which keeps two waves resident per SIMD.
    const bool wide = kernel_spl(a.n) == 4;
    hipStream_t st = (hipStream_t)stream;
    if (curves) {
        if (wide) hipLaunchKernelGGL((lnprob_kernel<true, 4>), grid, block, 0, st, sh, a);
        else hipLaunchKernelGGL((lnprob_kernel<true, 2>), grid, block, 0, st, sh, a);
    } else {
        if (wide) hipLaunchKernelGGL((lnprob_kernel<false, 4>), grid, block, 0, st, sh, a);
        else hipLaunchKernelGGL((lnprob_kernel<false, 2>), grid, block, 0, st, sh, a);
    }
    return (int)hipGetLastError();
}

}  // namespace mp
