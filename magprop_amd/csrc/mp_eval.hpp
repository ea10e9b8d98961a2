// mp_eval.hpp — evaluation of ONE walker's log-posterior on a wavefront (walker_eval): physics of the reference's RHS /
// luminosity stage in simplified algebra and the time-parallel exponential Adams-Moulton solver with tile-level stride
// adaptivity (serial restatement: oracle/mp_oracle.c mpo_trajectory_mode).  Included by
// mp_kernels.hip only.
#pragma once
#include "mp_math.hpp"

namespace mp {

// ---------------------------------------------------------------- per-walker constants
struct Walker {
    double inv_tau;   // 1/tvisc
    double S_amp;     // M0/tfb
    double inv_tfb;   // 1/tfb
    double m53_inv_tfb;   // -(5/3)/tfb (fallback rate's time derivative)
    double Crm;       // mu^(4/7) GM^(-1/7) f_Rm^(-2/7)
    double sqrtCrm;   // sqrt(Crm)
    double Crm15;     // Crm^1.5 / sqrt(GM)
    double DI;        // mu^2/(6 c^3 I)          dipole torque / I = -DI*omega^3
    double D;         // mu^2/(6 c^3)
    double armI;      // sqrt(GM)/I
    double kc;        // k*c
    double sqrt_kc;   // sqrt(k*c)
    double Kc;        // (k*c)^1.5 / sqrt(GM): fastness of a capped Alfven radius = Kc/sqrt(omega)
    double dipeff, propeff, f_beam;
};

// What the omega equation needs to know about the disc at the N time points of a lane (all omega-independent,
// computed once per tile in the time-parallel Mdisc phase).
template <int N>
struct DiscPt {
    Vd<N> mdot;  // Mdisc/tvisc
    Vd<N> rmu;   // uncapped Alfven radius, code/synthetic_datasets/funcs.py:105-106
    Vd<N> squ;   // sqrt(rmu)
    Vd<N> qu;    // rmu^1.5/sqrt(GM): uncapped fastness = omega*qu
};

template <int N>
MP_DEV DiscPt<N> disc_point(const DevShared &sh, const Walker &w, const Vd<N> &Mdisc) {
    DiscPt<N> p;
    FORN p.mdot[i] = Mdisc[i] * w.inv_tau;
    const Vd<N> t = pow_m1_7_fast(p.mdot);                              // mdot^(-1/7)
    FORN {
        const double t2 = t[i] * t[i];
        p.rmu[i] = w.Crm * t2;                                          // Crm * mdot^(-2/7)
        p.squ[i] = w.sqrtCrm * t[i];                                    // sqrt(rmu)
        p.qu[i] = w.Crm15 * (t2 * t[i]);                                // rmu^1.5 / sqrt(GM)
    }
    return p;
}

// fallback accretion rate Mdotfb(t), code/synthetic_datasets/funcs.py:128
template <int N>
MP_DEV Vd<N> mdot_fb(const Walker &w, const Vd<N> &t) {
    Vd<N> u;
    FORN u[i] = fma(t[i], w.inv_tfb, 1.0);                          // (t + tfb)/tfb >= 1
    const Vd<N> r = rcbrt_fast(u);
    Vd<N> out;
    FORN { const double r2 = r[i] * r[i]; out[i] = w.S_amp * (r2 * r2 * r[i]); }   // u^(-5/3)
    return out;
}

// the same with its time derivative dMdotfb/dt = -(5/3) Mdotfb / (t + tfb) and iu = tfb / (t + tfb)
template <int N>
MP_DEV Vd<N> mdot_fb_d(const Walker &w, const Vd<N> &t, Vd<N> &dS, Vd<N> &iu) {
    Vd<N> u;
    FORN u[i] = fma(t[i], w.inv_tfb, 1.0);
    const Vd<N> r = rcbrt_fast(u);
    Vd<N> out;
    FORN {
        const double r2 = r[i] * r[i], r3 = r2 * r[i];
        out[i] = w.S_amp * (r2 * r3);
        iu[i] = r3;                                                  // 1/u = r^3
        dS[i] = w.m53_inv_tfb * out[i] * r3;
    }
    return out;
}

// Radii / fastness / switch shared by the ODE right-hand side and the luminosity stage
// (code/synthetic_datasets/funcs.py:105-123 / magnetar/funcs.py:64-84 in simplified algebra):
//   Rm = min(rmu, k c/omega);  fastness = (Rm/Rc)^1.5 = omega Rm^1.5/sqrt(GM);  tanh(n (fastness-1)).
template <int N>
struct Flow {
    Vd<N> inv_om, Rm, sq, fast, e, r, th;
    Vb<N> capped, big;
};

template <int N>
MP_DEV Flow<N> flow_state(const Walker &w, double n, const DiscPt<N> &p, const Vd<N> &om) {
    Flow<N> f;
    const Vd<N> y = rsqrt_fast(om);
    Vd<N> x, ea, den;
    FORN f.inv_om[i] = y[i] * y[i];
    FORN {
        const double rlc = w.kc * f.inv_om[i];
        f.capped[i] = p.rmu[i] >= rlc;                              // Rm >= k*Rlc -> Rm = k*Rlc
        f.Rm[i] = f.capped[i] ? rlc : p.rmu[i];
    }
    FORN f.sq[i] = f.capped[i] ? w.sqrt_kc * y[i] : p.squ[i];       // sqrt(Rm)
    FORN f.fast[i] = f.capped[i] ? w.Kc * y[i] : om[i] * p.qu[i];
    FORN x[i] = fma(n, f.fast[i], -n);
    const double xmin = lane_minabs(x.v);                           // |x| > 19.5: tanh(x) = +-1 to the last bit
    const bool saturated = xmin > 19.5;                             // (false for a NaN lane minimum)
    if (__all(saturated)) {                                         // wave-uniform: deep propeller / deep accretion tiles
        FORN { f.e[i] = 0.0; f.r[i] = 1.0; f.th[i] = copysign(1.0, x[i]); }
    } else {
        FORN ea[i] = fmax(-2.0 * fabs(x[i]), -750.0);
        f.e = exp_fast(ea);
        FORN den[i] = 1.0 + f.e[i];
        f.r = rcp_fast(den);
        FORN f.th[i] = copysign((1.0 - f.e[i]) * f.r[i], x[i]);    // tanh(x) = eta2 - eta1
    }
    FORN f.big[i] = f.Rm[i] >= kR;
    return f;
}

// d(omega)/dt, code/synthetic_datasets/funcs.py:119,131-140; lam = d(omega_dot)/d(omega)
// ALT: built with the alternative dipole torque of code/figure_3.py:105-165 (cfg.dipole_torque = 1; the curve kernels and the
// right-hand-side kernel only: the hot mode-A kernels do not carry the branch):
//   Ndip = -(2/3) (mu^2 omega^3 / c^3) (Rlc / Rm)^3 = -(2/3) mu^2 / Rm^3 with Rm after the cap, i.e. Ndip / I = -4 DI (c / Rm)^3
// (DI = mu^2 / (6 c^3 I)); it depends on omega through the capped radius Rm = k c / omega only: d/d omega = -12 DI omega^2 / k^3.
template <bool WANT_LAM, bool ALT = false, int N>
MP_DEV Vd<N> omega_rhs(const DevShared &sh, const Walker &w, const DiscPt<N> &p, const Vd<N> &om, Vd<N> &rot,
                       Vd<N> &lam) {
    const Flow<N> f = flow_state(w, sh.cfg.n_ode, p, om);
    Vd<N> out;
    if constexpr (ALT) {
        if (sh.cfg.dipole_torque == 1) {
            const Vd<N> irm = rcp_fast(f.Rm);
            FORN {
                const double om2 = om[i] * om[i];
                rot[i] = sh.crot * om2;
                const double live = __hiloint2double(rot[i] > 0.27 ? 0 : 0x3FF00000, 0);
                const double arm = live * (w.armI * fmax(f.sq[i], sh.sqrtR));
                const double nacc = -arm * p.mdot[i] * f.th[i];
                const double cr = kC * irm[i];                                   // c / Rm  (= omega / k where the radius is capped)
                const double dip = -4.0 * w.DI * (cr * cr * cr);
                if (WANT_LAM) {
                    const double cf = __hiloint2double(f.capped[i] ? (int)0xBFE00000 : 0x3FF00000, 0);
                    const double dfast = cf * f.fast[i] * f.inv_om[i];
                    const double dth = sh.cfg.n_ode * (4.0 * f.e[i] * f.r[i] * f.r[i]) * dfast;
                    const double cd = __hiloint2double((f.capped[i] && f.sq[i] >= sh.sqrtR) ? (int)0xBFE00000 : 0, 0);
                    const double darm = cd * arm * f.inv_om[i];
                    const double dn = -p.mdot[i] * fma(darm, f.th[i], arm * dth);
                    const double ddip = f.capped[i] ? 3.0 * dip * f.inv_om[i] : 0.0;   // d/d omega of -4 DI (omega / k)^3
                    lam[i] = ddip + dn;
                }
                out[i] = dip + nacc;
            }
            return out;
        }
    }
    FORN {
        const double om2 = om[i] * om[i];
        rot[i] = sh.crot * om2;
        // break-up (Nacc = 0) as a 0/1 factor: only the high dword of the double differs
        const double live = __hiloint2double(rot[i] > 0.27 ? 0 : 0x3FF00000, 0);
        const double arm = live * (w.armI * fmax(f.sq[i], sh.sqrtR));      // sqrt(GM*max(Rm,R))/I, or 0 beyond break-up
        const double nacc = -arm * p.mdot[i] * f.th[i];                    // Nacc/I ; Macc - Mprop = -tanh * mdot
        if (WANT_LAM) {
            const double cf = __hiloint2double(f.capped[i] ? (int)0xBFE00000 : 0x3FF00000, 0);   // -0.5 : 1.0
            const double dfast = cf * f.fast[i] * f.inv_om[i];
            const double dth = sh.cfg.n_ode * (4.0 * f.e[i] * f.r[i] * f.r[i]) * dfast;   // n sech^2 dfast
            const double cd = __hiloint2double((f.capped[i] && f.sq[i] >= sh.sqrtR) ? (int)0xBFE00000 : 0, 0);   // -0.5 : 0
            const double darm = cd * arm * f.inv_om[i];
            const double dn = -p.mdot[i] * fma(darm, f.th[i], arm * dth);
            lam[i] = fma(-3.0 * w.DI, om2, dn);
        }
        out[i] = fma(-w.DI * om2, om[i], nacc);
    }
    return out;
}

// Which smooth branch of the right-hand side a state is on: bit 0 = Alfven radius capped at k*Rlc
// (code/synthetic_datasets/funcs.py:109-110), bit 1 = Rm >= R (the torque-arm branch, :133-138).  Crossing either is a kink of
// omega_dot that no multistep formula crosses at a coarse step (oracle/mp_oracle.c branch_flags).
MP_DEV int branch_flags(const Walker &w, double rmu, double om) {
    const bool capped = rmu * om >= w.kc;                           // rmu >= k c / omega
    const bool big = capped ? (w.kc >= kR * om) : (rmu >= kR);
    return (capped ? 1 : 0) | (big ? 2 : 0);
}

// luminosities (erg/s) at the lane's N grid points, reference luminosity stage
// (code/synthetic_datasets/funcs.py:204-229, magnetar/funcs.py:191-210)
template <int N>
MP_DEV void luminosity(const DevShared &sh, const Walker &w, const DiscPt<N> &p, const Vd<N> &om, Vd<N> &Ltot,
                       Vd<N> &Lprop, Vd<N> &Ldip) {
    const Flow<N> f = flow_state(w, sh.cfg.n_lum, p, om);
    Vd<N> irm;
    if (sh.cfg.lprop_gm_term) irm = rcp_fast(f.Rm);
    FORN {
        const double eta2 = f.th[i] >= 0.0 ? f.r[i] : f.e[i] * f.r[i];     // 0.5*(1 + tanh x)
        const double om2 = om[i] * om[i];
        const double rot = sh.crot * om2;
        const double arm = sh.sqrtGM * (f.big[i] ? f.sq[i] : sh.sqrtR);
        const double Nacc = rot > sh.cfg.nacc_lum_threshold ? 0.0 : -arm * p.mdot[i] * f.th[i];
        double ld = w.dipeff * (w.D * om2 * om2);
        if (ld <= 0.0) ld = 0.0;
        if (!isfinite(ld)) ld = 0.0;
        double lp = -Nacc * om[i];
        if (sh.cfg.lprop_gm_term) lp -= sh.GM * irm[i] * eta2 * p.mdot[i];
        lp *= w.propeff;
        if (lp <= 0.0) lp = 0.0;
        if (!isfinite(lp)) lp = 0.0;
        Ltot[i] = w.f_beam * (ld + lp);
        Lprop[i] = lp;
        Ldip[i] = ld;
    }
}

// Prior box, un-logging of the log-masked coordinates and the per-walker constants
// (code/synthetic_datasets/mcmc_eqns.py:16-17,28-49; funcs.py:98-102).  Returns MP_STATUS_OK or MP_STATUS_PRIOR.
MP_DEV int walker_setup(const DevShared &sh, const LaunchArgs &a, double (&par)[MP_MAX_NDIM], Walker &w) {
    // ---- prior, un-logging (code/synthetic_datasets/mcmc_eqns.py:16-17,28-49)
    int status = MP_STATUS_OK;
    if (!a.physical) {
        bool outside = false;
#pragma unroll
        for (int i = 0; i < MP_MAX_NDIM; ++i)
            if (i < sh.n_prior && (!(par[i] >= sh.lower[i]) || !(par[i] <= sh.upper[i]))) outside = true;
        if (outside) status = MP_STATUS_PRIOR;
#pragma unroll
        for (int i = 0; i < MP_MAX_NDIM; ++i)
            if (i < a.ndim && ((sh.log_mask >> i) & 1u)) par[i] = exp10_fast(par[i]);
    }

    // ---- walker constants (code/synthetic_datasets/funcs.py:98-102)
    {
        const double B = par[0], MdiscI = par[2], RdiscI = par[3], epsilon = par[4], delta = par[5];
        const double tau = (RdiscI * 1.0e5) / (sh.cfg.alpha * sh.cfg.cs7 * 1.0e7);
        const double mu = 1.0e15 * B * (kR * kR * kR);
        const Vd<1> Bv{{B}};
        const double b17 = pow_m1_7_fast(Bv)[0];                       // B^(-1/7); B^(4/7) = B * (B^(-1/7))^3
        const double M0 = delta * MdiscI * kMsol;
        const double tfb = epsilon * tau;
        w.inv_tau = 1.0 / tau;
        w.S_amp = M0 / tfb;
        w.inv_tfb = 1.0 / tfb;
        w.m53_inv_tfb = (-5.0 / 3.0) * w.inv_tfb;
        w.Crm = sh.crm_unit * (B * (b17 * b17 * b17));                 // mu^(4/7) GM^(-1/7) f^(-2/7), mu = 1e15 B R^3
        w.sqrtCrm = sqrt(w.Crm);
        w.Crm15 = w.Crm * w.sqrtCrm * sh.inv_sqrtGM;
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

    return status;
}

// (the compile-time constants of the stride policy -- MP_LOGPRED_MIN_KIND, MP_ABORT_SKIP_RATIO, MP_CUT_BY_RATIO, MP_LIGHT_TOL,
// MP_PRE_EARLY_END_FACTOR -- are defined in mp_device.h, where mp_get_policy reports them too)
constexpr int kMaxSweepsMargin = 16;  // sweeps allowed beyond the tile length (after which every step is exact)
// A tile whose iterates keep crossing the break-up limit (rotation parameter 0.27: the accretion torque switches off,
// code/synthetic_datasets/funcs.py:131-132) is chattering on that discontinuity and ends as a 'flag' anyway; after this
// many sweeps with an iterate beyond the limit the verdict is taken at once instead of after tile-length sweeps.
// Not fewer: a 256-step tile that starts from a poor guess may overshoot the limit for a dozen sweeps and still converge
// below it (5 of the 32 768 soak walkers did with a threshold of 8).
constexpr int kChatterSweeps = 24;
// A tile at a coarse stride is kept if at least this many lanes precede the first offending one (oracle: MPO_MIN_KEEP).
constexpr int kMinKeepLanes = 8;
// Sweeps a tile at a coarse stride may take: of one that has not converged by then (a poor extrapolated guess across a fast
// feature) the converged leading lanes are kept and the rest is redone at the next finer stride.
constexpr int kCoarseMaxSweeps = 5;
constexpr int kFineMaxSweeps = 12;    // the same for tiles over single intervals or sub-steps, if enough lanes have converged (mp_capi.cpp: fine_max_sweeps)

// cubic Hermite on a step of length h (theta in [0, 1]): value, and its time derivative
MP_DEV double hermite(double th, double h, double y0, double d0, double y1, double d1) {
    const double D = y1 - y0;
    return y0 + th * (h * d0 + th * ((3.0 * D - h * (2.0 * d0 + d1)) + th * (h * (d0 + d1) - 2.0 * D)));
}
MP_DEV double hermite_d(double th, double h, double y0, double d0, double y1, double d1) {
    const double D = y1 - y0;
    return (h * d0 + th * (2.0 * (3.0 * D - h * (2.0 * d0 + d1)) + th * 3.0 * (h * (d0 + d1) - 2.0 * D))) / h;
}
// quintic Hermite (values, first and second derivatives at both ends): Mdisc, whose derivatives are analytic
// (dM/dt = Mdotfb - M/tvisc), passes through a transition a few tvisc after the start where the cubic is 1e-9 off at a
// stride of four grid intervals
MP_DEV double hermite5(double th, double h, double y0, double d0, double e0, double y1, double d1, double e1) {
    const double t2 = th * th, t3 = t2 * th;
    const double s = t3 * fma(th, fma(th, 6.0, -15.0), 10.0);          // 10 t^3 - 15 t^4 + 6 t^5
    const double b1 = fma(t3, fma(th, fma(th, -3.0, 8.0), -6.0), th);  // t - 6 t^3 + 8 t^4 - 3 t^5
    const double b4 = t3 * fma(th, fma(th, -3.0, 7.0), -4.0);          // -4 t^3 + 7 t^4 - 3 t^5
    const double b2 = 0.5 * t2 * fma(th, fma(th, fma(th, -1.0, 3.0), -3.0), 1.0);   // (t^2 - 3 t^3 + 3 t^4 - t^5)/2
    const double b5 = 0.5 * t3 * fma(th, fma(th, 1.0, -2.0), 1.0);     // (t^3 - 2 t^4 + t^5)/2
    return fma(s, y1 - y0, y0) + h * fma(b1, d0, b4 * d1) + (h * h) * fma(b2, e0, b5 * e1);
}
// Mdisc inside a step: the quintic while the step resolves the viscous time (z = h/tvisc < 1); beyond, Mdisc follows the
// fallback rate quasi-steadily, a power law the cubic represents to 2e-10, and its derivatives, formed as differences of
// nearly equal terms, carry the rounding of Mdisc amplified by z (z^2 for the second): the cubic then
MP_DEV double hermite_mdisc(double th, double h, double z, double y0, double d0, double e0, double y1, double d1, double e1) {
    return z < 1.0 ? hermite5(th, h, y0, d0, e0, y1, d1, e1) : hermite(th, h, y0, d0, y1, d1);
}

// Q^k of the tile kinds 1 .. 4, k = 0 .. 64*SPL: the step end times of a tile are t_s Q^k (the grid is geometric).  Filled
// once per kernel (every lane computes its own entries), read by every tile.  (The sub-stepped tiles at the start, kind 0,
// compute theirs: there are one or two of them.)
template <int SPL>
struct TimeTable {
    static constexpr int kN = 64 * SPL + 1;
    double E[kKinds - 1][kN];
};

template <int SPL>
MP_DEV void time_table_init(const DevShared &sh, TimeTable<SPL> &tt) {   // every thread of the (one-wavefront) workgroup calls this
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int kind = 1; kind < kKinds; ++kind) {
#pragma unroll
        for (int s = 0; s < SPL; ++s) tt.E[kind - 1][lane * SPL + s + 1] = sh.ttab[(kind - 1) * kTtabN + lane * SPL + s + 1];
        if (lane == 0) tt.E[kind - 1][0] = 1.0;
    }
    __syncthreads();
}
// All three LDS tables of a kernel (polynomial coefficients, constants of the tile kinds, Q^k) in ONE round trip to global
// memory: every load is issued before the first store waits for its data, one barrier at the end (the three separate copies
// cost a walker three global-memory latencies in a row at kernel entry).  NT = threads of the workgroup.
template <int G, int NT>
MP_DEV void tables_init(const DevShared &sh, TimeTable<G> &tt) {
    constexpr int kN = TimeTable<G>::kN, nT = (kKinds - 1) * kN, cW = (kWtabSize + NT - 1) / NT, cT = (nT + NT - 1) / NT;
    const int tid = threadIdx.x;
    double vw[cW], vt[cT];
    const double vk = kKtabInit[tid < kKtabN ? tid : 0];
#pragma unroll
    for (int c = 0; c < cW; ++c) { const int i = tid + c * NT; vw[c] = sh.wtab[i < kWtabSize ? i : 0]; }
#pragma unroll
    for (int c = 0; c < cT; ++c) {
        const int i = tid + c * NT, ii = i < nT ? i : 0, kind = ii / kN, k = ii - kind * kN;
        vt[c] = sh.ttab[kind * kTtabN + k];
    }
    if (tid < kKtabN) g_ktab[tid] = vk;
#pragma unroll
    for (int c = 0; c < cW; ++c) { const int i = tid + c * NT; if (i < kWtabSize) g_wtab[i] = vw[c]; }
#pragma unroll
    for (int c = 0; c < cT; ++c) {
        const int i = tid + c * NT, kind = i / kN, k = i - kind * kN;
        if (i < nT) tt.E[kind][k] = vt[c];
    }
    __syncthreads();
}

// ---------------------------------------------------------------- a walker on W wavefronts (team kernels)
// Launches that leave SIMDs idle (n <= n_simd / 2 walkers: what emcee hands over per half-step at the headline
// configuration) run one walker on a WORKGROUP of W = 2 or 4 wavefronts, each on a SIMD of its own.  The tile stays the
// 256-step tile of the 4-steps-per-lane kernel and so does its policy: lane g of EVERY wavefront belongs to the group of
// steps 4 g .. 4 g + 3, of which wavefront v owns steps 4 g + v SPL + s, s = 0 .. SPL - 1 (SPL W = 4).  Everything that is a
// function of a group -- the composed step maps and their scan, the convergence tests, the ballots of the stride policy -- is
// formed redundantly and identically in every wavefront from values exchanged through LDS, so the wavefronts take every
// tile-level decision alike without asking each other; what costs instructions -- the right-hand side, the phi
// functions, the quadrature -- is evaluated for the own steps only.  Per Newton sweep two exchanges (write, s_barrier, read):
// (omega_dot, omega) at the own step ends, whose neighbours are the history nodes of the others' steps, and the own step maps
// with their smoothness indicators; per tile one more for the Mdisc maps and one for the per-step flags of the acceptance.
// Layout: [j][lane] with j the step's place in its lane's group -- every access of a wavefront is lane-contiguous (the natural
// [4 lane + j] order puts the lanes 32 bytes apart: a 4-way bank conflict on every exchange, measured as 61 % of the LDS
// unit's busy cycles and ~2 000 cycles per sweep with the four wavefronts of a CU queueing behind each other).
template <int G>
struct TeamX {
    double F[G][66];     // omega_dot at the step ends: [j][1 + g] = end of step G g + j; [j][0] = the point G - j steps before the
                         // tile's first step end ([G - 1][0] the tile's start, [G - 2][0] ... the history in front of it)
    double Wv[G][66];    // omega, same layout
    double A[G][64], B[G][64];   // step maps x -> A x + B (Mdisc phase, then every sweep)
    double D4[G][64];    // smoothness indicators of the steps
    unsigned flags[4][64];
};
// entry v SPL + s of a group array: the value of this wavefront's own step s (v is wave-uniform)
template <int SPL, int W>
MP_DEV double own_of(const double (&g)[SPL * W], int wave, int s) {
    double v = g[s];
#pragma unroll
    for (int k = 1; k < W; ++k) {
        v = wave == k ? g[k * SPL + s] : v;
        asm("" : "+v"(v));   // (keeps the chain of selects: the optimiser otherwise turns it into g[wave], a dynamically indexed private array)
    }
    return v;
}

// OR of a per-lane flag word over the wavefronts of the team, lane by lane (one exchange: every wavefront calls this at
// the same point of the program)
template <int W, int G>
MP_DEV unsigned team_or(TeamX<G> *tx, int wave, int lane, unsigned fl) {
    tx->flags[wave][lane] = fl;
    __syncthreads();
    unsigned r = 0u;
#pragma unroll
    for (int k = 0; k < W; ++k) r |= tx->flags[k][lane];
    return r;
}

// Q^k of kind `kind` (wave-uniform or per-lane k)
template <int SPL>
MP_DEV double time_factor(const DevShared &sh, const TimeTable<SPL> &tt, int kind, int k) {
    if (kind == 0) { const Vd<1> e{{(double)k * sh.sk[0].lnQ}}; return exp_fast(e)[0]; }
    return tt.E[kind - 1][k];
}

// The image of the last kept tile in LDS: node 0 = the tile's start point, node e + 1 = step end e.  It serves the
// observations (mode A picks the states bracketing each observed time out of it) and, as the record of the most recent
// steps, the multistep history of a successor tile whose step differs.
template <int SPL>
struct TileImage {
    static constexpr int kN = 64 * SPL + 1;
    double W[kN];   // omega
    double F[kN];   // omega_dot
    double M[kN];   // Mdisc
    double D[kN];   // dMdisc/dt
    double D2[kN];  // d2Mdisc/dt2
    double R[kN];   // uncapped Alfven radius (the branch of the right-hand side at the start of the next tile)
    double C[4];    // the fallback rate, its time derivative and tfb / (t + tfb) at the last kept node: the next tile's start
};

// Mdisc at the i-th skipped grid point of step J (time t) of a tile of `keep` kept steps where the step is longer than the
// viscous time: cubic Lagrange interpolation of the ratio Mdisc / (tvisc Mdotfb) through the four nodes around the step
// (the first / last four of the kept part for its first / last step; weights: LDS table), times tvisc Mdotfb(t).
template <int SPL>
MP_DEV double dense_mdisc_qs(const Walker &w, const TileImage<SPL> &im, int kind, int i, int J, int keep, double t) {
    const int l0 = max(min(J - 1, keep - 3), 0), variant = J - l0;     // nodes l0 .. l0 + 3; the step is [l0 + variant, + 1]
    const int base = kWtabDense + ((kind - 2) * 7 + (i - 1)) * 12 + variant * 4;
    Vd<4> M, S;
#pragma unroll
    for (int k = 0; k < 4; ++k) { M[k] = im.M[l0 + k]; S[k] = fma(M[k], w.inv_tau, im.D[l0 + k]); }   // Mdotfb = dMdisc/dt + Mdisc/tvisc
    const Vd<4> iS = rcp_fast(S);
    double r = 0.0;                                                    // tvisc x the interpolated ratio
#pragma unroll
    for (int k = 0; k < 4; ++k) r = fma(g_wtab[base + k], M[k] * iS[k], r);
    const Vd<1> tv{{t}};
    return r * mdot_fb(w, tv)[0];
}

// (Mdisc, omega) at position p8 (in eighths of a grid interval) inside the kept part of the image of a tile of kind
// `kind` that started at pos8 / time t_s with steps of d8 eighths: the node itself when p8 is one, else the dense output of
// its step (strides 2, 4, 8 only; the remainder is then a whole number of grid intervals).
template <int SPL>
MP_DEV void image_state(const DevShared &sh, const Walker &w, const TileImage<SPL> &im, const TimeTable<SPL> &tt, int kind,
                        int pos8, int keep, double t_s, int p8, double &Mv, double &Wv) {
    const int sh8 = kind == 0 ? 0 : kind + 2;                        // d8 = 1 << sh8
    const int rel = p8 - pos8, J = rel >> sh8, rem = rel - (J << sh8);
    if constexpr (SPL < 4) {
        // (the 128-step kernels keep two wavefronts per SIMD, which hide each other's LDS latency, on 256 registers each: the two
        // dozen values of the batched form below cost them scratch memory.  They read as they go.)
        Mv = im.M[J];
        Wv = im.W[J];
        if (rem != 0) {
            const int i = (rem >> 3) & 7;
            const double th = wtab_theta(kind, i);
            const double t1 = t_s * tt.E[kind - 1][J + 1], h = t1 * sh.sk[kind].one_m_invQ;
            const double z = h * w.inv_tau;
            if (z < 1.0 || keep < 3) Mv = hermite_mdisc(th, h, z, im.M[J], im.D[J], im.D2[J], im.M[J + 1], im.D[J + 1], im.D2[J + 1]);
            else Mv = dense_mdisc_qs(w, im, kind, i, J, keep, fma(th - 1.0, h, t1));
            Wv = hermite(th, h, im.W[J], im.F[J], im.W[J + 1], im.F[J + 1]);
        }
        return;
    }
    if (kind < 2) {                                                  // steps of one grid interval or less: every grid point is a node
        Mv = im.M[J];
        Wv = im.W[J];
        return;
    }
    // Everything the dense output of a step can need is read from LDS TOGETHER, then used (round 5): read as it was
    // needed -- node, position of the skipped point, step time, the six Hermite inputs, the four nodes and weights of the
    // quasi-steady form -- this was a chain of a dozen LDS round trips in a row, ~65 exposed cycles each with one wavefront
    // per SIMD, twice per tile.  Same arithmetic on the same values: bit-identical.  (Indices of lanes that do not use a
    // value are clamped into the arrays.)
    const int J1 = min(J + 1, TileImage<SPL>::kN - 1), i = (rem >> 3) & 7;
    const int l0 = max(min(J - 1, keep - 3), 0), variant = min(J - l0, 2);
    const int base = kWtabDense + ((kind - 2) * 7 + max(i - 1, 0)) * 12 + variant * 4;
    const double M0 = im.M[J], M1 = im.M[J1], D0 = im.D[J], D1 = im.D[J1], E0 = im.D2[J], E1 = im.D2[J1];
    const double W0 = im.W[J], W1 = im.W[J1], F0 = im.F[J], F1 = im.F[J1];
    const double th = wtab_theta(kind, i), Et = tt.E[kind - 1][J1];
    Vd<4> Mq, Dq, wq;
#pragma unroll
    for (int k = 0; k < 4; ++k) { Mq[k] = im.M[l0 + k]; Dq[k] = im.D[l0 + k]; wq[k] = g_wtab[base + k]; }
    Mv = M0;
    Wv = W0;
    if (rem != 0) {
        const double t1 = t_s * Et, h = t1 * sh.sk[kind].one_m_invQ;
        const double z = h * w.inv_tau;
        if (z < 1.0 || keep < 3) {
            Mv = hermite_mdisc(th, h, z, M0, D0, E0, M1, D1, E1);
        } else {   // (dense_mdisc_qs on the values read above)
            Vd<4> S;
#pragma unroll
            for (int k = 0; k < 4; ++k) S[k] = fma(Mq[k], w.inv_tau, Dq[k]);
            const Vd<4> iS = rcp_fast(S);
            double r = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) r = fma(wq[k], Mq[k] * iS[k], r);
            const Vd<1> tv{{fma(th - 1.0, h, t1)}};
            Mv = r * mdot_fb(w, tv)[0];
        }
        Wv = hermite(th, h, W0, F0, W1, F1);
    }
}

// Phase profile (developer build, `make -C magprop_amd/csrc phase-profile`, tools/phase_profile.py): the shader clock is read
// between the sections of a tile and the per-section sums of one walker replace its tile log.  Compiles to nothing otherwise.
#ifdef MP_PHASE_PROFILE
#define MP_PHASE_N 18
#define MP_PHASE_DECL unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph_acc[MP_PHASE_N] = {0};
#define MP_PHASE(i) { const unsigned long long ph_n = __builtin_amdgcn_s_memtime(); ph_acc[i] += ph_n - ph_t; ph_t = ph_n; }
#define MP_PHASE_DUMP if (a.tile_log && lead) { for (int i = 0; i < MP_PHASE_N; ++i) a.tile_log[(size_t)walker * MP_TILE_LOG + i] = (int32_t)min(ph_acc[i], 0x7FFFFFFFull); }
#define MP_TILE_LOG_ON false   // (the per-tile words would be overwritten by the dump, and their stores would be waited for inside the timed sections)
#else
#define MP_PHASE_DECL
#define MP_PHASE(i)
#define MP_PHASE_DUMP
#if defined(MP_SWEEP_TRACE) || defined(MP_CORR_TRACE)
#define MP_TILE_LOG_ON false
#else
#define MP_TILE_LOG_ON true
#endif
#endif

// ---------------------------------------------------------------- the kernel
// Evaluate ONE walker on the calling wavefront (all 64 lanes enter with identical arguments).
// LOG: compiled with the per-tile diagnostics (mp_tile_log); the production builds are not: the words' bookkeeping (the `why`
// bits need four ballots nothing else uses) costs scalar registers in kernels that spill them.
// SPL = consecutive steps owned by one lane; a tile is 64*SPL steps.  par[] holds the sampler coordinates
// (prior checked and log-masked coordinates un-logged here unless a.physical); walker indexes ds_id and the
// optional curve outputs; im / Lbuf are the wave's LDS areas (Lbuf: [8*64*SPL + 1], staging of the curve outputs).
// W = wavefronts per walker (team kernels, see TeamX): the workgroup has 64 W threads, a lane's group of steps is SPL W
// long, im / tt are dimensioned for it and tx is the team's exchange area (nullptr for W = 1).
// UNI: the walker's constants live in scalar registers (the builds that keep two wavefronts resident per SIMD).
template <bool CURVES, int SPL, bool LONG, bool LOG = false, int W = 1, bool UNI = (SPL == 2 && W == 1)>
MP_DEV void walker_eval(const DevShared &sh, const LaunchArgs &a, int walker, double (&par)[MP_MAX_NDIM],
                        TileImage<SPL * W> &im, const TimeTable<SPL * W> &tt, double *Lbuf, double &lnp_out, int &status_out,
                        int &sweeps_out, int &tiles_out, TeamX<SPL * W> *tx = nullptr) {
    static_assert(W == 1 || (SPL * W == 4 && !CURVES), "team kernels: 256-step tiles, mode A");
    constexpr int kSPL = SPL, kG = SPL * W, kTile = 64 * kG, kMaxSweeps = kTile + kMaxSweepsMargin;
    const int lane = threadIdx.x & 63;
    const int wave = W > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;   // which wavefront of the walker's team
    const int j0 = wave * kSPL;              // its first step inside a lane's group of kG steps
    const int e0 = lane * kG + j0;           // this lane's first own step in the tile
    const bool lead = W > 1 ? threadIdx.x == 0 : lane == 0;
    MP_PHASE_DECL
    // every lane's correction in the previous sweep: the 2-steps-per-lane kernels have no register to spare (LDS); the
    // 4-steps-per-lane ones have no LDS to spare where they stage light curves (four workgroups of 40 KB per CU)
    __shared__ float s_dsum_prev[kG >= 4 ? 1 : 64];

    const int n_grid = sh.n_grid;
    const int nsteps = n_grid - 1;
    const size_t row = (size_t)walker * (size_t)n_grid;

    Walker w;
    int status = walker_setup(sh, a, par, w);
    if constexpr (UNI) {
        // two waves per SIMD: 256 vector registers per wave.  The walker's constants are wave-uniform; in scalar registers
        // (spilled, if at all, to lanes of a vector register, not to memory) they free 30 vector registers and the kernel
        // no longer reloads spilled values from scratch memory inside the tile loop.
        w.inv_tau = uniform(w.inv_tau); w.S_amp = uniform(w.S_amp); w.inv_tfb = uniform(w.inv_tfb); w.Crm = uniform(w.Crm);
        w.sqrtCrm = uniform(w.sqrtCrm); w.Crm15 = uniform(w.Crm15); w.DI = uniform(w.DI); w.D = uniform(w.D);
        w.armI = uniform(w.armI); w.kc = uniform(w.kc); w.sqrt_kc = uniform(w.sqrt_kc); w.Kc = uniform(w.Kc);
        w.dipeff = uniform(w.dipeff); w.propeff = uniform(w.propeff); w.f_beam = uniform(w.f_beam);
        // (kernels for long light curves: one more; elsewhere the constant is better off in a vector register pair: 4 096 Classic walkers +1 %)
        if constexpr (LONG) w.m53_inv_tfb = uniform(w.m53_inv_tfb);
    }

    // ---- state carried from tile to tile (all wave-uniform)
    double t_s = sh.t0;
    double M_s = par[2] * kMsol;                         // initial conditions, code/synthetic_datasets/funcs.py:66-69
    double om_s = (2.0 * M_PI) / (1.0e-3 * par[1]);
    double cf0;
    int flags_s;
    double L_s, Lp_s, Ld_s;
    {
        const Vd<1> Mv{{M_s}}, ov{{om_s}};
        const DiscPt<1> d_s = disc_point(sh, w, Mv);
        Vd<1> rot0, dummy, Lt0, Lp0, Ld0;
        cf0 = omega_rhs<false, CURVES>(sh, w, d_s, ov, rot0, dummy)[0];
        flags_s = branch_flags(w, d_s.rmu[0], om_s);
        if (status == MP_STATUS_OK) {
            if (!(isfinite(M_s) && isfinite(om_s)) || M_s <= 0.0 || om_s <= 0.0) status = MP_STATUS_NONFINITE;
            else if (rot0[0] > 0.27) status = MP_STATUS_FLAG;
        }
        if constexpr (CURVES) {
            luminosity(sh, w, d_s, ov, Lt0, Lp0, Ld0);
            L_s = Lt0[0]; Lp_s = Lp0[0]; Ld_s = Ld0[0];
        } else {
            L_s = Lp_s = Ld_s = 0.0;
        }
    }

    // A walker whose ds_id names no registered light curve (out of range, or a slot never set: n_obs = 0) must not
    // pass as a perfect fit (chi^2 = 0): lnprob = -inf with its own status, on every entry point.
    const int dsid = a.ds_id ? a.ds_id[walker] : 0;
    const bool ds_in_range = sh.ds != nullptr && dsid >= 0 && dsid < sh.n_ds;
    const DsDesc dsd = ds_in_range ? sh.ds[dsid] : DsDesc{0, 0, 0, 0};   // (n_obs, obs_off, tile_off, flags)
    if (a.want_chi2 && dsd.n_obs <= 0) status = MP_STATUS_BADDATASET;
    const int32_t *tptr = sh.tile_ptr + dsd.tile_off;
    // The first 64 observations of the walker's light curve live in registers, one per lane (time-sorted;
    // every synthetic set has 50).  The further observations of a longer light curve are scored tile by tile, when the
    // tile that brackets them is committed (below).
    // (mode A needs only the grid interval while the tiles go by; the other four values of the observation are fetched
    // behind the last tile, where they are used: eight registers less across the tile loop, which the 2-steps-per-lane
    // kernels, two waves per SIMD, were spilling to scratch memory)
    int ob_g = -1;
    double ob_dx = 0.0, ob_idt = 0.0, ob_y = 0.0, ob_ye = 1.0;
    if (a.want_chi2 && lane < dsd.n_obs) {
        const int jj = dsd.obs_off + lane;
        ob_g = sh.obs_g[jj];
        if constexpr (CURVES) {
            ob_dx = sh.obs_dx[jj];
            ob_idt = sh.obs_idt[jj];
            ob_y = sh.obs_y[jj];
            ob_ye = sh.obs_yerr[jj];
        }
    }
    // LONG: compiled with the path for light curves of more than 64 points (the launcher picks this variant when the
    // handle holds such a dataset; the short variant keeps that code out of the register budget)
    const bool long_lc = (CURVES || LONG) && a.want_chi2 && dsd.n_obs > 64;
    const bool deferred = !CURVES && a.want_chi2;               // see "observations" below
    // Every observed time of this light curve is a grid point (the reference's synthetic sets are built that way): np.interp's
    // weight of the upper bracket is exactly zero, fma((L1 - L0) idt, 0, L0) = L0 bit for bit (the luminosities are finite by
    // construction), so neither the state nor the luminosity at grid point g + 1 is formed (mode A, the first 64 observations).
    const bool on_knots = !LONG && (dsd.flags & kDsOnKnots) != 0;
    double obM[2] = {1.0e30, 1.0e30}, obW[2] = {1.0e3, 1.0e3};  // (Mdisc, omega) at the observation's bracketing grid points
    double chi = 0.0, chi_long = 0.0;
    int sweeps_total = 0, tiles_total = 0;
#ifdef MP_SWEEP_TRACE
    int tr_tile = -1;
#endif
#ifdef MP_CORR_TRACE
    int tr_word = 0;
#endif

    if (status == MP_STATUS_OK) {
        if (CURVES && lead) {
            if (a.ltot) a.ltot[row] = L_s / 1.0e50;
            if (a.lprop) a.lprop[row] = Lp_s / 1.0e50;
            if (a.ldip) a.ldip[row] = Ld_s / 1.0e50;
            if (a.mdisc) a.mdisc[row] = M_s;
            if (a.omega) a.omega[row] = om_s;
        }
        // ---- tile control (wave-uniform).  Positions in eighths of a grid interval: the first pre_fine intervals are
        // covered with 1/8-interval sub-steps (kind 0), the rest with steps of 1, 2, 4 or 8 intervals (kinds 1 .. 4).
        // (sub-stepped start: 32 grid intervals = one tile of 256 sub-steps, two of 128 in the 2-steps-per-lane kernels -- which
        // end it after the first when that tile was calm, see MP_PRE_EARLY_END_FACTOR below)
        const int max_kind = sh.max_kind;
        const int end8 = 8 * nsteps;
        int pre_end8 = 8 * sh.pre_fine;
        int pos8 = 0, kind = 1;
        bool rec_valid = false;                  // the image holds a kept tile (the history of the next one)
        int rec_kind = 0, rec_sh8 = 0, rec_J = 0; // its kind, log2 of its step in eighths and the number of steps kept
        int cool = 0;                            // tiles over single intervals for which the scaled indicator decides about coarsening
        int hold = 0, hold_kind = 0;             // tiles of kind hold_kind that stay there after the sweeps of the next coarser kind were slow
        int trouble = 0;                         // coarse tiles of this walker that were given up or kept nothing: after two of
                                                 // them stride 8 is no longer TRIED (heavy discs around fast, strongly magnetised
                                                 // stars: the sweeps of 2 048-interval tiles converge too slowly), only reached
                                                 // when the indicator of a full tile over 4 intervals promotes it
        int opt_kind = max_kind;                 // the kind that is tried after a calm tile over single intervals: lowered when
                                                 // such an attempt fails outright, raised when the indicator promotes a tile
        double cS0, cdS0, ciu0;                  // the fallback rate at the tile's start, its time derivative, tfb / (t_s + tfb)
        {
            const Vd<1> tsv{{t_s}};
            Vd<1> dS0, iu0;
            cS0 = mdot_fb_d(w, tsv, dS0, iu0)[0];
            cdS0 = dS0[0];
            ciu0 = iu0[0];
        }
        // Each lane owns kSPL consecutive steps of the tile: steps lane*kSPL + s, s = 0..kSPL-1.
        MP_PHASE(0)
        while (pos8 < end8) {
            const bool pre = pos8 < pre_end8;
            if (pre) kind = 0;
            int d8 = pre ? 1 : (4 << kind);                                 // 8, 16, 32, 64 eighths
            const int left8 = (pre ? pre_end8 : end8) - pos8;
            // (whole steps only; and at least three of them, for the dense output's four nodes)
            const int want_kind = kind;
            if (!pre) while (kind > 1 && ((left8 & (d8 - 1)) != 0 || left8 < 3 * d8)) { --kind; d8 >>= 1; }
            const int sh8 = pre ? 0 : kind + 2;                             // d8 = 1 << sh8
            int nc = min(kTile, left8 >> sh8);                              // steps of this tile that exist
            // What is left after a tile must be a whole number of the NEXT tile's steps.  A cut tile can leave a remainder that is
            // not a multiple of 8 grid intervals (kept lanes x steps per lane x stride), and full tiles of a finer stride
            // preserve it: until round 4 a walker so placed ran at 4 intervals to the end of the grid (burnt-in Classic walkers
            // of the 2-steps-per-lane kernels: ten tiles over 4 where five over 8 do).  Now the tile that had to step finer than
            // the policy asked ends a few steps early, where the asked-for stride fits.
            if (!pre && kind < want_kind && left8 >= 3 * (4 << want_kind)) {
                const int D = 4 << want_kind;                               // the asked-for step in eighths
                const int after = left8 - (nc << sh8);
                const int m = ((D - (after & (D - 1))) & (D - 1)) >> sh8;   // steps to leave for the next tile
                if (nc - m >= 3) nc -= m;
            }
            const StrideK &K = sh.sk[kind];
            const int wbase = kind * kWtabStride;                           // this kind's quadrature matrices in the LDS table
            ++tiles_total;
            MP_PHASE(8)

            // ---------------- step end times (the grid is geometric: t_k = t_s Q^k) and step lengths
            Vd<kSPL> h, S1, dS1, iu1;
            double Sp_team = 0.0, iup_team = 0.0;   // team kernels: the fallback rate and tfb / (t + tfb) at the start of this lane's first own step
            // (the kind's constants once, and ONE branch on the kind around the four table reads: read step by step, each behind its
            // own branch and its own scalar load, they were eight memory round trips in a row)
            const double K_omq = K.one_m_invQ, K_lnQ = K.lnQ, K_invQ = K.inv_Q;
            if constexpr (W == 1) {
                Vd<kSPL> tb;
                if constexpr (kSPL >= 4) {
                    if (kind == 0) {
#pragma unroll
                        for (int s = 0; s < kSPL; ++s) tb[s] = time_factor(sh, tt, 0, min(e0 + s + 1, nc));
                    } else {
#pragma unroll
                        for (int s = 0; s < kSPL; ++s) tb[s] = tt.E[kind - 1][min(e0 + s + 1, nc)];
                    }
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) {
                        tb[s] = t_s * tb[s];
                        h[s] = (e0 + s < nc) ? tb[s] * K_omq : 0.0;   // 0 for the padding steps of a short tile
                    }
                } else {   // (two wavefronts per SIMD hide these round trips; this form costs the 128-step kernels no scratch)
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) {
                        tb[s] = t_s * time_factor(sh, tt, kind, min(e0 + s + 1, nc));
                        h[s] = (e0 + s < nc) ? tb[s] * K.one_m_invQ : 0.0;
                    }
                }
                S1 = mdot_fb_d(w, tb, dS1, iu1);
            } else {
                // (the start of a lane's first own step is a step end of ANOTHER wavefront: its fallback rate is evaluated here,
                // next to the own ones -- the same function of the same time, so the same bits -- instead of being exchanged)
                Vd<kSPL + 1> tb, Sx, dSx, iux;
                if (kind == 0) {
#pragma unroll
                    for (int s = 0; s <= kSPL; ++s) tb[s] = time_factor(sh, tt, 0, min(e0 + s, nc));
                } else {
#pragma unroll
                    for (int s = 0; s <= kSPL; ++s) tb[s] = tt.E[kind - 1][min(e0 + s, nc)];
                }
#pragma unroll
                for (int s = 0; s <= kSPL; ++s) tb[s] = t_s * tb[s];
#pragma unroll
                for (int s = 0; s < kSPL; ++s) h[s] = (e0 + s < nc) ? tb[s + 1] * K_omq : 0.0;
                Sx = mdot_fb_d(w, tb, dSx, iux);
#pragma unroll
                for (int s = 0; s < kSPL; ++s) { S1[s] = Sx[s + 1]; dS1[s] = dSx[s + 1]; iu1[s] = iux[s + 1]; }
                Sp_team = e0 == 0 ? cS0 : Sx[0];
                iup_team = e0 == 0 ? ciu0 : iux[0];
            }

            MP_PHASE(10)
            // (the fallback rate at the tile's start, its derivative and tfb / (t + tfb) -- cS0, cdS0, ciu0 -- are the values the
            // previous tile computed at its last kept step end, carried through the image; the first tile's: before the loop)
            // ---------------- history at this tile's spacing: (omega_dot, omega) at the three (predictor: four) previous
            // points come from the record of the last kept tile
            double cf1 = cf0, cf2 = cf0, cf3 = cf0, cw1 = om_s, cw2 = om_s, cw3 = om_s, cw4 = om_s;
            bool have4 = false, interp_hist = false;
            const bool startup = !rec_valid;
            if (rec_valid) {
                // lane k-1 looks up the point k steps of this tile before its start (k = 1..4): node ja of the record,
                // or `rem` eighths before it inside the record's step [ja - 1, ja]
                const int o8 = (lane + 1) << sh8, idx = o8 >> rec_sh8, rem = o8 - (idx << rec_sh8);
                const int ja = rec_J - idx, jb = ja - (rem != 0 ? 1 : 0);
                double wv = om_s, fv = cf0;
                const bool avail = lane < 4 && jb >= 0;
                if (avail) {
                    wv = im.W[ja];
                    fv = im.F[ja];
                    if (rem != 0) {
                        const StrideK &R = sh.sk[rec_kind];
                        const Vd<1> e1{{-(double)rem * sh.lnq8}}, e2{{-(double)idx * R.lnQ}};
                        const double th = (exp_fast(e1)[0] - R.inv_Q) / R.one_m_invQ;    // time fraction inside [jb, ja]
                        const double hr = t_s * exp_fast(e2)[0] * R.one_m_invQ;
                        wv = hermite(th, hr, im.W[jb], im.F[jb], im.W[ja], im.F[ja]);
                        fv = hermite_d(th, hr, im.W[jb], im.F[jb], im.W[ja], im.F[ja]);
                    }
                }
                const unsigned long long av = __ballot(avail);
                interp_hist = (__ballot(avail && rem != 0) & 0xFull) != 0ull;
                // (the record always holds the three points the formula needs: it has kMinKeepLanes lanes at least)
                cw1 = lane_bcast(wv, 0); cf1 = lane_bcast(fv, 0);
                cw2 = lane_bcast(wv, 1); cf2 = lane_bcast(fv, 1);
                cw3 = lane_bcast(wv, 2); cf3 = lane_bcast(fv, 2);
                have4 = (av & 0xFull) == 0xFull;
                cw4 = have4 ? lane_bcast(wv, 3) : cw3;
            }

            MP_PHASE(1)
            // ---------------- Mdisc: exponential step with the analytic source + affine scan.  The source is a power law of
            // t + tfb: inside a step Mdotfb(t_j + theta h) = Mdotfb(t_j) (1 + theta x)^(-5/3), x = h/(t_j + tfb) <= 1 - 1/Q, and
            // its binomial series integrates term by term against the exponential kernel,
            //   Mdisc_{j+1} = e^{-z} Mdisc_j + h Mdotfb(t_j) sum_k c_k (-x)^k phi_{k+1}(-z),  c_k = Gamma(k + 5/3)/Gamma(5/3),
            // k = 0..5: where the step is much longer than tvisc the series is that of (1 + x)^(-5/3) itself and the first
            // neglected term is 4 x^6 = 7e-12 at a stride of 8 grid intervals (oracle/mp_oracle.c).
            Vd<kSPL> M1;
            DiscPt<kSPL> d1;
            {
                Vd<kSPL> zm, S0, x;
                double Sp, iup;                                                         // at this lane's first step start
                if constexpr (W == 1) { Sp = lane_prev(S1[kSPL - 1], cS0); iup = lane_prev(iu1[kSPL - 1], ciu0); }
                else { Sp = Sp_team; iup = iup_team; }
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    zm[s] = -h[s] * w.inv_tau;
                    S0[s] = s == 0 ? Sp : S1[s - 1];
                    x[s] = h[s] * w.inv_tfb * (s == 0 ? iup : iu1[s - 1]);
                }
                const Phi5<kSPL> pm = phi12345(zm);
                const Vd<kSPL> p6 = phi6(zm, pm);
                Vd<kSPL> inc;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    double P = fma(-x[s] * (104720.0 / 243.0), p6[s], (6160.0 / 81.0) * pm.p5[s]);
                    P = fma(x[s], P, (-440.0 / 27.0) * pm.p4[s]);
                    P = fma(x[s], P, (40.0 / 9.0) * pm.p3[s]);
                    P = fma(x[s], P, (-5.0 / 3.0) * pm.p2[s]);
                    P = fma(x[s], P, pm.p1[s]);
                    inc[s] = h[s] * S0[s] * P;                 // padding steps: h = 0
                }
                Vd<kSPL> am, bm;
                double A = 1.0, B = 0.0;                  // composition of this lane's step maps
                if constexpr (W == 1) {
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) {
                        am[s] = pm.e[s];      // padding steps: h = 0 -> e = 1, inc = 0
                        bm[s] = inc[s];
                        B = fma(am[s], B, bm[s]);
                        A = A * am[s];
                    }
                    scan_affine(A, B);
                    double Ax, Bx;
                    lane_prev_map(A, B, Ax, Bx);   // exclusive prefix
                    double Mc = fma(Ax, M_s, Bx);            // Mdisc at this lane's first step start
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) { Mc = fma(am[s], Mc, bm[s]); M1[s] = Mc; }
                } else {
                    // team: the own step maps go to LDS, every wavefront composes the maps of the lane's whole group in step order
                    // (the arithmetic of the 4-steps-per-lane kernel), scans, and takes its own steps' values
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) { tx->A[j0 + s][lane] = pm.e[s]; tx->B[j0 + s][lane] = inc[s]; }
                    __syncthreads();
                    double ga[kG], gb[kG], Mg[kG];
#pragma unroll
                    for (int j = 0; j < kG; ++j) { ga[j] = tx->A[j][lane]; gb[j] = tx->B[j][lane]; }
#pragma unroll
                    for (int j = 0; j < kG; ++j) { B = fma(ga[j], B, gb[j]); A = A * ga[j]; }
                    scan_affine(A, B);
                    double Ax, Bx;
                    lane_prev_map(A, B, Ax, Bx);
                    double Mc = fma(Ax, M_s, Bx);
#pragma unroll
                    for (int j = 0; j < kG; ++j) { Mc = fma(ga[j], Mc, gb[j]); Mg[j] = Mc; }
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) M1[s] = own_of<kSPL, W>(Mg, wave, s);
                }
                d1 = disc_point(sh, w, M1);
            }

            MP_PHASE(2)
            // ---------------- omega: predictor = extrapolation of the last five values in the step index
            // (the grid is logarithmic, so power laws are smooth in the index) ...
            Vd<kSPL> wg;                                  // current guess of omega at this lane's step ends
            {
                // Newton backward-difference extrapolation (quartic once five values exist).  History values that were
                // interpolated inside the record's steps (a finer successor) carry the interpolant's ~1e-11 ripple, which
                // the higher differences would amplify by the cube / fourth power of the tile length: quadratic then.
                // (compile-time: the 2-steps-per-lane kernels, two waves per SIMD on 256 registers each, do not carry this code)
                constexpr bool kLogPredEver = kG >= 4;
                bool guessed = false;
                if constexpr (kLogPredEver) {
                if (!startup && kind >= MP_LOGPRED_MIN_KIND) {
                    guessed = true;
                    // Coarse tiles (round 4): a tile over 8 grid intervals spans 1.2 decades of time, over which omega follows
                    // power laws, not polynomials: the quartic in the index was off by 15 % (median) to a factor of 6 (one tile in
                    // ten) at the tile's end, and its 4th difference amplifies the 1e-10 noise of the history by 2e8.  Instead
                    // log(omega) is extrapolated with the DERIVATIVE's history: d log(omega)/dk = omega_dot t ln(Q) / omega is known
                    // at the tile's start and the two points before it (exact right-hand sides, no differencing of values);
                    // its quadratic Newton polynomial in the index, integrated from 0 to k, gives
                    //   log(omega_k / omega_s) = a0 k + d1 k^2/2 + d2 (k^3/6 + k^2/4),   d1, d2 = backward differences of a.
                    // Median error at the end of such a tile 5 %, one tile in ten 60 % (tests/diagnostics/predictor_study.py); the Newton
                    // sweeps double the number of correct digits per pass, so this saves most coarse tiles a sweep.
                    const Vd<3> den{{om_s, cw1, cw2}};
                    const Vd<3> rd = rcp_fast(den);
                    const double tl = t_s * K_lnQ;
                    const double a0 = tl * cf0 * rd[0], a1 = (tl * K_invQ) * cf1 * rd[1], a2 = (tl * K_invQ * K_invQ) * cf2 * rd[2];
                    const double d1 = a0 - a1, d2 = d1 - (a1 - a2);
                    Vd<kSPL> ex;
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) {
                        const double k = (double)(e0 + s + 1);
                        const double e = k * fma(k, fma(fma(k, 1.0 / 6.0, 0.25), d2, 0.5 * d1), a0);
                        ex[s] = fmin(fmax(e, -4.0), 4.0);     // (a guess only: never further than a factor of 55 from the start value)
                    }
                    const Vd<kSPL> ee = exp_fast(ex);
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) wg[s] = om_s * ee[s];
                }
                }
                if (!guessed) {
                const double g1 = om_s - cw1, g2 = g1 - (cw1 - cw2);
                const double d2b = (cw1 - cw2) - (cw2 - cw3);
                const double g3 = (startup || interp_hist) ? 0.0 : g2 - d2b;
                const double g4 = (startup || interp_hist || !have4) ? 0.0 : g3 - (d2b - ((cw2 - cw3) - (cw3 - cw4)));
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    const double k = (double)(e0 + s + 1);
                    const double c2 = 0.5 * k * (k + 1.0);
                    const double c3 = c2 * (k + 2.0) * (1.0 / 3.0);
                    wg[s] = fma(k, g1, fma(c2, g2, fma(c3, g3, fma(c3 * (k + 3.0) * 0.25, g4, om_s))));
                }
                }
            }
            MP_PHASE(3)
            // ... then Newton-type sweeps of the linearised step maps.
            // E*[k]: values at the four points before this lane's first step (k = 0..3) and at its step ends (k = 4+s).
            double Ef[kSPL + 4], Ew[kSPL + 4];
            unsigned long long flagged = 0ull, pending = ~0ull, pending_tight = ~0ull;
            bool settled = false;    // this lane's guesses moved by < 1e-6 in the previous sweep: close enough to its solution
                                     // for an excursion beyond the break-up limit to be the solution's, not the iteration's
            int sweep = 0, over_sweeps = 0;
            float dsum_prev = 0.0f;   // this lane's correction in the previous sweep (4-steps-per-lane kernels)
            Ew[3] = om_s;
            // A sweep that follows a small correction (every lane moved by < MP_LIGHT_TOL) keeps the Jacobian lambda, e^{h lambda}
            // and the quadrature weights of the previous one and only re-evaluates omega_dot ("light" sweep): the scheme
            // may linearise about any nearby point, the result moves by ~1e-14, and the sweep costs a third less.
            // (Until round 4 a sweep behind a FULL sweep with a correction below 1e-5 did not evaluate omega_dot at all but
            // linearised it: the cheap verification pass of a tile whose first guess was good.  Those passes are no longer
            // run -- see the end of the sweep -- and without that path every kernel is 3 - 4 % faster on top.)
            Vd<kSPL> lam, ez, p5, n0, n1, n2, n3, n4;
            EamW5<kSPL> cw;
            bool light = false, early_stop = false;
            // (a step over 2, 4 or 8 grid intervals weighs an error of omega_dot that many times as much: tighter sweeps there)
            const double tol_k = kind >= 2 ? sh.coarse_tol_factor * sh.sweep_tol : sh.sweep_tol;
            // (the tile tried at a coarse stride right behind the sub-steps has no calm predecessor to vouch for it: a tenth;
            // likewise every coarse tile that starts before t = early_hold_t (4 s), and steps over 8 intervals:
            // oracle/mp_oracle.c)
            const double tile_tol = (kind >= 2 && (rec_kind == 0 || t_s < sh.early_hold_t)) ? 0.1 * sh.stride_tol : (kind >= 4 ? sh.k4_tol_factor * sh.stride_tol : sh.stride_tol);
            bool abort_tile = false, abort_skip = false;
            while (true) {
                ++sweep;
                if (!light) {   // (after a sweep that moved every lane by < MP_LIGHT_TOL the guesses are positive and finite)
                    bool wild = false;
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) wild = wild || !(wg[s] > 0.0);
                    if (__any(wild)) {   // keep the iteration alive after a wild or NaN guess (rare)
#pragma unroll
                        for (int s = 0; s < kSPL; ++s)
                            if (!(wg[s] > 0.0)) wg[s] = Ew[3] > 0.0 ? Ew[3] : om_s;
                    }
                }
                MP_PHASE(4)
                Vd<kSPL> rot, f1;
#ifdef MP_CORR_TRACE
                const bool full = !light;                 // lambda, e^{h lambda} and the weights are renewed in this sweep
#endif
                if (light) {
                    Vd<kSPL> unused;
                    f1 = omega_rhs<false, CURVES>(sh, w, d1, wg, rot, unused);
                } else {
                    f1 = omega_rhs<true, CURVES>(sh, w, d1, wg, rot, lam);
                }
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    Ef[4 + s] = f1[s];
                    Ew[4 + s] = wg[s];
                }
                MP_PHASE(13)
                // largest rotation parameter among this lane's step ends (the padding steps of a short tile repeat its
                // last point once the first sweep has run; before that nothing is decided on them: `settled` is false)
                double h1 = cf1, h2 = cf2, h3 = cf3, u1 = cw1, u2 = cw2, u3 = cw3;
                if (startup) {   // the three points before the grid continue points 0 and 1 linearly in the index
                    // (team kernels: point 1 belongs to the first wavefront, and only its first lane uses these values)
                    const double fp1 = lane_bcast(Ef[4], 0), wp1 = lane_bcast(Ew[4], 0);
                    h1 = 2.0 * cf0 - fp1; u1 = 2.0 * om_s - wp1;
                    h2 = 3.0 * cf0 - 2.0 * fp1; u2 = 3.0 * om_s - 2.0 * wp1;
                    h3 = 4.0 * cf0 - 3.0 * fp1; u3 = 4.0 * om_s - 3.0 * wp1;
                }
                double Wold[kG];                          // team kernels: omega at the step ends of the lane's whole group, as evaluated in this sweep
                double rot_max;
                if constexpr (W == 1) {
                    rot_max = lane_max(rot.v);
                } else {
                    // exchange 1: (omega_dot, omega) at the own step ends to LDS; back come the four points before this lane's
                    // first own step (other wavefronts' step ends, or the history in front of the tile) and omega of the group
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) { tx->F[j0 + s][1 + lane] = f1[s]; tx->Wv[j0 + s][1 + lane] = wg[s]; }
                    if (lead && (startup || sweep == 1)) {   // (the points in front of the tile: fixed, except in the walker's first tile)
                        tx->F[3][0] = cf0; tx->F[2][0] = h1; tx->F[1][0] = h2; tx->F[0][0] = h3;
                        tx->Wv[3][0] = om_s; tx->Wv[2][0] = u1; tx->Wv[1][0] = u2; tx->Wv[0][0] = u3;
                    }
                    // (the phi functions and the quadrature weights need nothing of the exchange: between the writes and the
                    // barrier they cover the LDS latency and whatever the wavefronts are apart)
                    if (!light) {
                        Vd<kSPL> zw;
#pragma unroll
                        for (int s = 0; s < kSPL; ++s) zw[s] = h[s] * lam[s];
                        const EamRows2 rows = eam5_rows_begin(wbase);
                        const Phi5<kSPL> pw_ = phi12345(zw);
                        ez = pw_.e;
                        p5 = pw_.p5;
                        cw = eam5_node_weights_pipelined(wbase, pw_, rows);
                    }
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {   // the point 4 - k steps before this lane's first own step: in the previous lane's group or in this one
                        const int q = j0 + k;
                        Ef[k] = tx->F[q & 3][lane + (q >> 2)];
                        Ew[k] = tx->Wv[q & 3][lane + (q >> 2)];
                    }
#pragma unroll
                    for (int j = 0; j < kG; ++j) Wold[j] = tx->Wv[j][1 + lane];
                    double rg[kG];
#pragma unroll
                    for (int j = 0; j < kG; ++j) rg[j] = sh.crot * (Wold[j] * Wold[j]);
                    rot_max = lane_max(rg);
                }
                const bool flg = rot_max > 0.27;
                const bool near_limit = rot_max > 0.26;   // close to the break-up switch of the torque: no linearisation
                // break-up reached by an iterate that is no longer a wild guess: the reference's 'flag'
                flagged |= __ballot(settled && flg);
                const unsigned long long over_now = __ballot(flg);
                over_sweeps += over_now != 0ull;
                if constexpr (W == 1) {
                    Ef[3] = lane_prev(Ef[kSPL + 3], cf0);  Ew[3] = lane_prev(Ew[kSPL + 3], om_s);
                    Ef[2] = lane_prev(Ef[kSPL + 2], h1);   Ew[2] = lane_prev(Ew[kSPL + 2], u1);
                    Ef[1] = lane_prev(Ef[kSPL + 1], h2);   Ew[1] = lane_prev(Ew[kSPL + 1], u2);
                    Ef[0] = lane_prev(Ef[kSPL + 0], h3);   Ew[0] = lane_prev(Ew[kSPL + 0], u3);
                }
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    n0[s] = fma(-lam[s], Ew[4 + s], Ef[4 + s]);
                    n1[s] = fma(-lam[s], Ew[3 + s], Ef[3 + s]);
                    n2[s] = fma(-lam[s], Ew[2 + s], Ef[2 + s]);
                    n3[s] = fma(-lam[s], Ew[1 + s], Ef[1 + s]);
                    n4[s] = fma(-lam[s], Ew[s], Ef[s]);
                }
                MP_PHASE(14)
                if (W == 1 && !light) {
                    Vd<kSPL> zw;
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) zw[s] = h[s] * lam[s];
                    if constexpr (kSPL >= 4) {   // (one wavefront per SIMD: the quadrature table's rows are read ahead of their use)
                        const EamRows2 rows = eam5_rows_begin(wbase);
                        const Phi5<kSPL> pw_ = phi12345(zw);
                        ez = pw_.e;
                        p5 = pw_.p5;
                        cw = eam5_node_weights_pipelined(wbase, pw_, rows);
                    } else {
                        const Phi5<kSPL> pw_ = phi12345(zw);
                        ez = pw_.e;
                        p5 = pw_.p5;
                        cw = eam5_node_weights(wbase, pw_);
                    }
                }
                MP_PHASE(15)
                const Vd<kSPL> inc = eam5_increment_nodes(cw, h, n0, n1, n2, n3, n4);
                Vd<kSPL> aw, bw;
                double A = 1.0, B = 0.0;
                double wc, dsum = 0.0;
                double Wnew[kG], Dg[kG];                  // team kernels: the new omega and the smoothness indicators of the lane's whole group
                if constexpr (W == 1) {
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) {
                        aw[s] = ez[s];
                        bw[s] = inc[s];
                        B = fma(aw[s], B, bw[s]);
                        A = A * aw[s];
                    }
                    scan_affine(A, B);
                    double Ax, Bx;
                    lane_prev_map(A, B, Ax, Bx);
                    wc = fma(Ax, om_s, Bx);           // omega at this lane's first step start
                    // The convergence tests look at the lane's MEAN correction (its steps are consecutive and move together):
                    // one set of comparisons per lane instead of one per step.  NaN propagates through the sum: "not converged".
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) {
                        wc = fma(aw[s], wc, bw[s]);
                        dsum += fabs(wc - wg[s]);
                        wg[s] = wc;
                    }
                } else {
                    // exchange 2: the own step maps (and, at coarse strides, the own steps' smoothness indicators: the early give-up
                    // below looks at them) to LDS; every wavefront composes the group's maps in step order, scans, propagates the
                    // tile's start value through ALL steps of the group and forms the group's correction: the convergence tests and
                    // every decision that follows see the same numbers in every wavefront
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) { tx->A[j0 + s][lane] = ez[s]; tx->B[j0 + s][lane] = inc[s]; }
                    if (kind >= 2) {
#pragma unroll
                        for (int s = 0; s < kSPL; ++s)
                            tx->D4[j0 + s][lane] = 120.0 * fabs(p5[s]) * h[s] * fabs((n0[s] + n4[s]) - 4.0 * (n1[s] + n3[s]) + 6.0 * n2[s]);
                    }
                    __syncthreads();
                    double ga[kG], gb[kG];
#pragma unroll
                    for (int j = 0; j < kG; ++j) { ga[j] = tx->A[j][lane]; gb[j] = tx->B[j][lane]; Dg[j] = 0.0; }
                    if (kind >= 2) {
#pragma unroll
                        for (int j = 0; j < kG; ++j) Dg[j] = tx->D4[j][lane];
                    }
#pragma unroll
                    for (int j = 0; j < kG; ++j) { B = fma(ga[j], B, gb[j]); A = A * ga[j]; }
                    scan_affine(A, B);
                    double Ax, Bx;
                    lane_prev_map(A, B, Ax, Bx);
                    wc = fma(Ax, om_s, Bx);
#pragma unroll
                    for (int j = 0; j < kG; ++j) {
                        wc = fma(ga[j], wc, gb[j]);
                        dsum += fabs(wc - Wold[j]);
                        Wnew[j] = wc;
                    }
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) wg[s] = own_of<kSPL, W>(Wnew, wave, s);
                }
                MP_PHASE(16)
                const double mag = (double)kG * fabs(wc);
                const bool all_settled = dsum <= 1.0e-6 * mag;                       // false for NaN
                const bool all_small = dsum <= MP_LIGHT_TOL * mag;
                const bool all_ok = dsum <= tol_k * mag;
                settled = all_settled;
                light = __all(all_small);
                pending = __ballot(!all_ok);
#ifdef MP_CORR_TRACE
                bool est_stop = false;
#endif
                // The last sweep of a tile used to be a verification pass: its correction is below the tolerance, it changes
                // nothing that matters, and near the truths it was 7 of a walker's 25 sweeps.  Where the corrections contract fast
                // it is not needed to know that: a lane whose correction fell by at least a factor of ten from the previous sweep,
                // and whose next correction -- no larger than this one times that factor (linear estimate; Newton sweeps
                // contract faster) -- would be below stop_factor (a hundredth, MP_STOP_FACTOR) of the tolerance, is converged.
                // The tile ends when that holds on every lane, away from the break-up switch of the accretion torque (lambda is
                // the derivative at the point this sweep evaluated: no linearisation sees that discontinuity); omega_dot at
                // the step ends follows the last correction through its linearisation, f + lambda (omega_new - omega_old),
                // exact to the second order in that correction.
                if (sweep >= 2 && pending != 0ull) {
                    const double dp = (double)(kG >= 4 ? dsum_prev : s_dsum_prev[lane]), lim_c = sh.stop_factor * tol_k * mag;
                    const bool pass = dsum <= lim_c || (all_small && dsum <= 0.1 * dp && dsum * dsum <= lim_c * dp);
                    if (__all(pass && !near_limit)) {
#pragma unroll
                        for (int s = 0; s < kSPL; ++s) Ef[4 + s] = fma(lam[s], wg[s] - Ew[4 + s], Ef[4 + s]);
                        pending = 0ull;
#ifdef MP_CORR_TRACE
                        est_stop = true;
#endif
                    }
                }
                if constexpr (kG >= 4) dsum_prev = (float)dsum;
                else s_dsum_prev[lane] = (float)dsum;
#ifdef MP_SWEEP_TRACE
                // developer build (make sweep-trace): how the converged region of a slowly converging tile over single intervals grows,
                // one word per sweep in the walker's tile-log row: first pending lane | pending lanes << 8 | lanes beyond the break-up limit << 16
                if (a.tile_log && lead && kind <= 1 && sweep <= MP_TILE_LOG && (tr_tile < 0 || tr_tile == tiles_total)) {
                    if (sweep > 12) tr_tile = tiles_total;
                    if (tr_tile < 0 || tr_tile == tiles_total)
                        a.tile_log[(size_t)walker * MP_TILE_LOG + sweep - 1] = (pending ? __ffsll(pending) - 1 : 64) | (__popcll(pending) << 8) | (__popcll(over_now) << 16) | (tiles_total << 24);
                }
#endif
#ifdef MP_CORR_TRACE
                // developer build (make corr-trace, tools/corr_trace.py): one word per sweep of every tile in the walker's tile-log row:
                // tile | kind << 8 | (0 full, 1 light) << 11 | (tile ended on the contraction estimate) << 12 | pending lanes << 13 | -10 log10(largest relative correction) << 20
                {
                    double rel = dsum / mag;
                    if (!(rel >= 1.0e-25)) rel = rel == rel ? 1.0e-25 : 1.0;
                    for (int d = 32; d >= 1; d >>= 1) rel = fmax(rel, __shfl_xor(rel, d, 64));
                    const int q = min(255, max(0, (int)(-10.0 * log10(rel) + 0.5)));
                    if (a.tile_log && lead && tr_word < MP_TILE_LOG)
                        a.tile_log[(size_t)walker * MP_TILE_LOG + tr_word++] = (tiles_total & 0xFF) | (kind << 8) | ((full ? 0 : 1) << 11) | ((est_stop ? 1 : 0) << 12) | (__popcll(pending) << 13) | (q << 20);
                }
#endif
                // (lanes of a tile that is stopped before all of it has converged are kept only if their own last
                // correction was a hundred times below the tolerance: slow sweeps contract by 0.3-0.5 per pass, so a
                // correction just below the tolerance leaves an error of the same size)
                pending_tight = __ballot(!(dsum <= 0.01 * tol_k * mag));
                if (pending == 0ull || flagged != 0ull || sweep >= kMaxSweeps) break;
                if (kind >= 2 && sweep >= sh.coarse_max_sweeps) break;   // not worth it at this stride (the rest is redone finer)
                // A coarse tile whose first lanes (the ones closest to the known start: they converge first) show a smoothness
                // indicator beyond the bound after the third sweep (well beyond it after the second) will keep nothing: it is
                // given up now instead of after coarse_max_sweeps sweeps.  (Measured on 8 000 coarse tiles of prior-wide walkers:
                // tiles that went on to keep lanes had <= 0.8 of the bound there after sweep 3; of those that kept nothing,
                // 90 % were beyond it after sweep 2 already.)
                if (kind >= 2 && sweep >= 2) {
                    const double margin = sweep == 2 ? 4.0 : 1.0;
                    bool hot = false, very_hot = false;
                    if constexpr (W == 1) {
#pragma unroll
                        for (int s = 0; s < kSPL; ++s) {
                            const double d4 = 120.0 * fabs(p5[s]) * h[s] * fabs((n0[s] + n4[s]) - 4.0 * (n1[s] + n3[s]) + 6.0 * n2[s]);
                            const double lim = margin * tile_tol * wg[s];
                            hot = hot || d4 > lim;
                            very_hot = very_hot || d4 > MP_ABORT_SKIP_RATIO * lim;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < kG; ++j) {   // (the group's indicators came with exchange 2)
                            const double lim = margin * tile_tol * Wnew[j];
                            hot = hot || Dg[j] > lim;
                            very_hot = very_hot || Dg[j] > MP_ABORT_SKIP_RATIO * lim;
                        }
                    }
                    if ((__ballot(hot) & ((1ull << kMinKeepLanes) - 1ull)) != 0ull) {
                        abort_tile = true;
                        abort_skip = (__ballot(very_hot) & ((1ull << kMinKeepLanes) - 1ull)) != 0ull;
                        break;
                    }
                }
                // Slow sweeps on single intervals (a poor extrapolated guess through a fast spin-up, far from the break-up
                // limit): the lanes that have converged are final (a step depends on earlier ones only); they are kept
                // and a new tile starts behind them with a fresh extrapolation, instead of sweeping on over all 64 lanes.
                if (kind <= 1 && sweep >= sh.fine_max_sweeps && over_sweeps == 0 && __ffsll(pending_tight) - 1 >= 2 * kMinKeepLanes) {
                    early_stop = true;
                    break;
                }
                if (over_sweeps >= kChatterSweeps) { flagged |= over_now ? over_now : pending; break; }
            }
            sweeps_total += sweep;
            MP_PHASE(17)
            if (abort_tile) {                                                   // redo at stride 1 (as after a tile that keeps nothing)
#ifdef MP_ABORT_STUDY
                // developer build (tools/abort_study.py): by how much the first lanes exceeded the bound they were held to, as
                // 8 log2 of the wave maximum of indicator / (margin x bound) over the first kMinKeepLanes lanes, in the tile log
                double hot_ratio = 0.0;
                {
                    const double margin = sweep == 2 ? 4.0 : 1.0;
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) {
                        const double d4 = 120.0 * fabs(p5[s]) * h[s] * fabs((n0[s] + n4[s]) - 4.0 * (n1[s] + n3[s]) + 6.0 * n2[s]);
                        const double r = d4 / (margin * tile_tol * wg[s]);
                        hot_ratio = (lane < kMinKeepLanes && r > hot_ratio) ? r : hot_ratio;     // (a NaN ratio is ignored)
                    }
#pragma unroll
                    for (int d = 4; d >= 1; d >>= 1) hot_ratio = fmax(hot_ratio, __shfl_xor(hot_ratio, d, 64));
                    hot_ratio = lane_bcast(hot_ratio, 0);
                }
                const int hot_q = min(255, max(0, (int)(8.0 * log2(fmax(hot_ratio, 1.0)))));
#else
                const int hot_q = abort_skip ? 255 : 0;
#endif
                if (MP_TILE_LOG_ON && LOG && a.tile_log && lead && tiles_total <= MP_TILE_LOG)
                    a.tile_log[(size_t)walker * MP_TILE_LOG + tiles_total - 1] = kind | (sweep << 4) | (hot_q << 16) | (64 << 24);
                cool = 3;
                ++trouble;
                // the next finer stride is tried at once -- or the one after it, when the first lanes were beyond the bound by
                // more than a halving of the step buys (order 5: a factor of 32).  Measured on 1 700 given-up tiles of 4 096
                // prior-wide walkers (tools/abort_study.py, profiles/r04_abort_study.log): with an excess >= 32 the next finer
                // stride kept nothing either in 50 % (from 8 intervals) / 95 % (from 4) of the cases and a dozen lanes otherwise.
                int drop = abort_skip ? 2 : 1;
                // ... or single intervals at once when the first lanes sit on a kink of the right-hand side (it makes the
                // indicator hot at every coarse stride, and a coarse tile would only be kept up to it: near the Classic and
                // Sloped truths the 128-step kernels spent a tile over 4 intervals on a dozen lanes there)
                {
                    bool brk = false;
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) brk = brk || (e0 + s < nc && branch_flags(w, d1.rmu[s], wg[s]) != flags_s);
                    if constexpr (W > 1) brk = team_or<W>(tx, wave, lane, brk ? 1u : 0u) != 0u;   // (any step of the lane's group)
                    if ((__ballot(brk) & ((1ull << kMinKeepLanes) - 1ull)) != 0ull) drop = 4;
                }
                opt_kind = max(2, kind - (drop > 2 ? 1 : drop));
                kind = kind - drop >= 2 ? kind - drop : 1;
                continue;
            }

            // ---------------- failure detection in time order (oracle/mp_oracle.c).  Per lane: a NaN or an
            // infinity anywhere shows in the sum, a non-positive value in the minimum, the break-up limit in the largest
            // omega (the padding steps of a short tile repeat its last point).
            unsigned long long mb, mf;
            unsigned team_fl = 0u;   // team kernels: the per-step flags of this tile, OR-ed over the steps of the lane's group (all wavefronts)
            {
                double vsum = M1[0] + wg[0];
#pragma unroll
                for (int s = 1; s < kSPL; ++s) vsum += M1[s] + wg[s];
                const double vmin = min_raw(lane_min(M1.v), lane_min(wg.v)), wmax = lane_max(wg.v);
                bool bad = !isfinite(vsum) || !(vmin > 0.0);
                bool over = sh.crot * wmax * wmax > 0.27;
                if constexpr (W > 1) {
                    // One exchange for everything the rest of the tile decides on: these two and the flags of the acceptance
                    // below, computed here for the own steps.  Its barrier is also the one behind which the image may be
                    // rewritten (every earlier reader of the image is in front of it).
                    const double prom8 = 64.0 / sh.k4_tol_factor;
                    unsigned fl = (bad ? 1u << 9 : 0u) | (over ? 1u << 10 : 0u);
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) {
                        const bool valid = e0 + s < nc;
                        const double d4 = 120.0 * fabs(p5[s]) * h[s] * fabs((n0[s] + n4[s]) - 4.0 * (n1[s] + n3[s]) + 6.0 * n2[s]);
                        const double lim = tile_tol * wg[s];
                        fl |= (valid && branch_flags(w, d1.rmu[s], wg[s]) != flags_s) ? 1u : 0u;
                        fl |= d4 > lim ? 2u : 0u;
                        fl |= 64.0 * d4 > lim ? 4u : 0u;
                        fl |= 2048.0 * d4 > lim ? 8u : 0u;
                        fl |= 65536.0 * d4 > lim ? 16u : 0u;
                        fl |= prom8 * d4 > lim ? 32u : 0u;
                        fl |= d4 > 16.0 * lim ? 128u : 0u;
                        fl |= d4 > 512.0 * lim ? 256u : 0u;
                    }
                    team_fl = team_or<W>(tx, wave, lane, fl);
                    bad = (team_fl & (1u << 9)) != 0u;
                    over = (team_fl & (1u << 10)) != 0u;
                }
                mb = __ballot(bad);
                // a step whose sweeps never settle is chattering on the Nacc discontinuity: same verdict as a flag
                mf = flagged | __ballot(over) | ((flagged || early_stop) ? 0ull : pending);
                if (early_stop) { mb &= ~pending_tight; mf &= ~pending_tight; }   // nothing is decided on lanes that are not kept
            }
            MP_PHASE(11)
            // At a coarse stride nothing of this is a verdict: the lanes before the first one that failed, or whose sweeps
            // had not converged when they were stopped, hold converged steps (a step depends on earlier ones only) and
            // are kept like the steps before a kink; the rest is redone finer.
            const unsigned long long unconv = kind >= 2 ? (mb | mf | (pending ? pending_tight : 0ull)) : (early_stop ? pending_tight : 0ull);
            if (kind <= 1 && (mb | mf)) {
                const int first = __ffsll((unsigned long long)(mb | mf)) - 1;
                status = ((mf >> first) & 1ull) ? MP_STATUS_FLAG : MP_STATUS_NONFINITE;
                break;
            }

            // ---------------- what is kept, and the stride of the next tile (oracle/mp_oracle.c mpo_trajectory_mode).
            // Per lane: does the solution leave the smooth branch of the right-hand side the tile started on; and the
            // smoothness indicator 120 |phi_5(h lambda)| h |4th difference of (f - lambda omega)| / omega against stride_tol, and against
            // stride_tol / 64 and / 2048 (what it would be at twice / four times the step: 5th-order scaling, margin 2).
            int keep_lanes = 64, next_kind = kind, why = 0;   // why: diagnostics (tile log)
            {
                bool brk = false, ind1 = false, ind64 = false, ind2048 = false, ind65536 = false, indp8 = false, indpre = false;
#if MP_CUT_BY_RATIO > 0
                bool ind16x = false, ind512x = false;
#endif
                const double prom8 = 64.0 / sh.k4_tol_factor;   // a tile over 8 intervals is held to k4_tol_factor x the bound: its
                                                                // promotion asks the same of the scaled indicator
                if constexpr (W == 1) {
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    const bool valid = e0 + s < nc;
                    brk = brk || (valid && branch_flags(w, d1.rmu[s], wg[s]) != flags_s);
                    // (the formula's error term: h phi_5(h lambda) x the 4th difference; 120 phi_5 = 1 at 0, -> 5/|h lambda| where
                    // the equation is stiff and the exponential integrator tracks the spin equilibrium)
                    const double d4 = 120.0 * fabs(p5[s]) * h[s] * fabs((n0[s] + n4[s]) - 4.0 * (n1[s] + n3[s]) + 6.0 * n2[s]);
                    const double lim = tile_tol * wg[s];
                    ind1 = ind1 || d4 > lim;
                    ind64 = ind64 || 64.0 * d4 > lim;
                    ind2048 = ind2048 || 2048.0 * d4 > lim;
                    ind65536 = ind65536 || 65536.0 * d4 > lim;
                    if constexpr (kG < 4) indpre = indpre || MP_PRE_EARLY_END_FACTOR * d4 > lim;
                    indp8 = indp8 || prom8 * d4 > lim;
#if MP_CUT_BY_RATIO > 0
                    ind16x = ind16x || d4 > 16.0 * lim;
                    ind512x = ind512x || d4 > 512.0 * lim;
#endif
                }
                } else {   // (formed for the own steps in front of the failure detection and exchanged there)
                    (void)prom8;
                    brk = (team_fl & 1u) != 0u; ind1 = (team_fl & 2u) != 0u; ind64 = (team_fl & 4u) != 0u; ind2048 = (team_fl & 8u) != 0u;
                    ind65536 = (team_fl & 16u) != 0u; indp8 = (team_fl & 32u) != 0u;
#if MP_CUT_BY_RATIO > 0
                    ind16x = (team_fl & 128u) != 0u; ind512x = (team_fl & 256u) != 0u;
#endif
                }
                const unsigned long long B = __ballot(brk), I1 = __ballot(ind1), I64 = __ballot(ind64), I2048 = __ballot(ind2048),
                                         I65536 = __ballot(ind65536), Ip8 = __ballot(indp8);
                const int full_lanes = (nc + kG - 1) / kG;                 // lanes that hold steps of this tile
                why = (B != 0ull ? 1 : 0) | (I1 != 0ull ? 2 : 0) | (I64 != 0ull ? 4 : 0) | (I2048 != 0ull ? 8 : 0) | (unconv != 0ull ? 16 : 0) |
                      (I65536 != 0ull ? 32 : 0);
                if (kind >= 2) {
                    const unsigned long long bad = B | I1 | unconv;
                    const int first = bad ? __ffsll(bad) - 1 : 64;
                    if (first < full_lanes) {
                        if (first < 2 * kMinKeepLanes) cool = 3;                // a coarse attempt that failed early
                        if (first < kMinKeepLanes) {                            // nothing worth keeping: redo at stride 1
                            opt_kind = max(2, kind - 1);
                            ++trouble;
                            if (MP_TILE_LOG_ON && LOG && a.tile_log && lead && tiles_total <= MP_TILE_LOG)
                                a.tile_log[(size_t)walker * MP_TILE_LOG + tiles_total - 1] = kind | (sweep << 4) | (why << 24);
                            // a kink in those lanes: single intervals; else the next finer stride is tried at once
                            kind = (kind > 2 && (B & ((1ull << kMinKeepLanes) - 1ull)) == 0ull) ? kind - 1 : 1;
                            continue;
                        }
                        keep_lanes = first;
                        // a kink or a fast feature gets single intervals; slow sweeps alone, the next finer stride, which is
                        // then held for a few tiles
                        next_kind = (((B | I1) >> first) & 1ull) ? 1 : kind - 1;
                        bool by_ratio = false;
#if MP_CUT_BY_RATIO > 0
                        // round 4: a fast feature (no kink) gets the stride its excess over the bound asks for (order 5: a halving of
                        // the step buys 32 x; margin 2 as in the promotions), judged over the MP_CUT_BY_RATIO lanes from the cut on:
                        // near the truths the propeller switch-on exceeds the stride-8 bound by < 16 x and a stride-4 tile takes it
                        // where single intervals needed a tile of their own (9 -> 8 tiles, 28 -> 25 sweeps per walker; burnt-in
                        // ensembles 10.2 -> 9.1 tiles; profiles/r04_ab_cut_by_ratio.log).  The tile that follows is cut where IT
                        // does not hold, like any other.
                        if (((B >> first) & 1ull) == 0ull && ((I1 >> first) & 1ull) != 0ull) {
                            const unsigned long long win = (first + MP_CUT_BY_RATIO < 64 ? (1ull << (first + MP_CUT_BY_RATIO)) - 1ull : ~0ull) & ~((1ull << first) - 1ull);
                            const unsigned long long X16 = __ballot(ind16x), X512 = __ballot(ind512x);
                            if ((B & win) == 0ull) {
                                next_kind = (X16 & win) == 0ull ? kind - 1 : ((X512 & win) == 0ull ? max(kind - 2, 1) : 1);
                                by_ratio = kG < 4;
                            }
                        }
#endif
                        // (a coarse successor is held for three tiles either way: releasing the one a feature's excess asked for at
                        // once, or promoting a calm stride-2 tile straight to 8, changed nothing measurable: profiles/r04_ab_cut_by_ratio.log)
                        if (next_kind >= 2 && !by_ratio) { hold_kind = next_kind; hold = 3; }
                    } else if (kind < max_kind && nc + 8 > kTile) {            // (a full tile, or one that ended a few steps early to realign)
                        // (256-step tiles: the hold ends early when the held tile shows that the slow zone is behind -- sweeps converged
                        // in three passes, indicator with room for four times the step: burnt-in ensembles 0.107 -> 0.103 ms; with
                        // 128-step tiles it cost the sampler 1 %: profiles/r04_ab_hold_release.log)
                        if (kG >= 4 && hold > 0 && kind == hold_kind && sweep <= 3 && I2048 == 0ull) hold = 0;
                        if (hold > 0 && kind == hold_kind) --hold;
                        else next_kind = (kind == 3 ? Ip8 : I64) == 0ull ? kind + 1 : kind;
                        opt_kind = max(opt_kind, min(next_kind, max_kind));
                    }
                } else if (unconv != 0ull) {                                    // kinds 0, 1 stopped early: the converged lanes
                    keep_lanes = __ffsll(unconv) - 1;
                    if (pre) keep_lanes &= ~(8 / kG - 1);                    // (whole grid intervals of sub-steps)
                } else if (kind == 1 && nc + 8 > kTile) {
                    // No kink in this tile: the scaled indicator decides while a recent coarse attempt has failed early
                    // (cool > 0); otherwise the coarsest stride is simply tried (a tile is cut where it does not hold): the
                    // indicator of a tile whose sweeps stopped at the tolerance carries their residual, amplified by the 4th
                    // difference, and kept stiff late-time stretches at single intervals for a dozen tiles.
                    if (B == 0ull) {
                        if (cool > 0) { --cool; next_kind = I65536 == 0ull ? 4 : (I2048 == 0ull ? 3 : (I64 == 0ull ? 2 : 1)); }
                        else next_kind = I1 == 0ull ? (trouble >= sh.trouble_limit ? min(opt_kind, 3) : opt_kind) : 1;
                    }
                    else {
                        // a kink inside this tile: the history of a coarse successor must lie behind it
                        const int first_clean = __ffsll(B) - 1 + 2;            // lanes from here on are past the kink
                        const int tail = nc - first_clean * kG;              // steps in them
                        const unsigned long long post = first_clean < 64 ? ~0ull << first_clean : 0ull;
                        if (tail >= 24 + kG && (I65536 & post) == 0ull) next_kind = 4;
                        else if (tail >= 12 + kG && (I2048 & post) == 0ull) next_kind = 3;
                        else if (tail >= 6 + kG && (I64 & post) == 0ull) next_kind = 2;
                    }
                }
                // after the sub-stepped tiles: optimistic (a tile that meets a fast feature is cut) -- as far as the record of the
                // last of them reaches: the successor's history points lie 1, 2, 3 of ITS steps before its start, and a tile
                // of 128 sub-steps (2 steps per lane) covers 16 grid intervals, not the 24 a step over 8 asks for.  (Until round 4
                // that attempt was made, read a default for the third point, failed its indicator test in the first lanes and
                // was redone over 4 intervals: a tile wasted per walker in every 2-steps-per-lane launch.)
                if constexpr (kG < 4) {
                    if (pre && max_kind > 1 && keep_lanes == 64 && nc == kTile && pos8 + nc < pre_end8 && B == 0ull && __ballot(indpre) == 0ull)
                        pre_end8 = pos8 + nc;                                  // the sub-stepped start ends here
                }
                if (kind == 0) next_kind = ((keep_lanes * kG) >> 3) >= 24 ? 4 : (((keep_lanes * kG) >> 3) >= 12 ? 3 : (((keep_lanes * kG) >> 3) >= 6 ? 2 : 1));
                next_kind = min(next_kind, max_kind);
            }
            MP_PHASE(5)
            const int keep = min(keep_lanes * kG, nc);                       // steps kept
            if (MP_TILE_LOG_ON && LOG && a.tile_log && lead && tiles_total <= MP_TILE_LOG)
                a.tile_log[(size_t)walker * MP_TILE_LOG + tiles_total - 1] = kind | (sweep << 4) | (keep_lanes << 16) | (why << 24);
            const int end_kept8 = pos8 + keep * d8;

            // ---------------- commit: the tile's image goes to LDS (record for the next tile's history, source of the
            // observations' states)
            // (curve kernels: in a tile whose kept steps are all longer than tvisc the dense output of Mdisc never uses the
            // second derivative; the D2 array then carries the node values of Mdisc / Mdotfb its quasi-steady form
            // interpolates: one reciprocal per step here instead of four per skipped grid point there)
            bool all_qs = false;
            Vd<kSPL> ratio;
            if constexpr (CURVES) {
                bool lane_qs = true;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) lane_qs = lane_qs && (e0 + s >= keep || h[s] * w.inv_tau >= 1.0);
                all_qs = kind >= 2 && keep >= 3 && __all(lane_qs);
                if (all_qs) {
                    const Vd<kSPL> iS = rcp_fast(S1);
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) ratio[s] = M1[s] * iS[s];
                }
            }
            if constexpr (W == 1) __syncthreads();                              // earlier readers of the image are done (team kernels: the flag exchange above)
#pragma unroll
            for (int s = 0; s < kSPL; ++s) {
                const int e = e0 + s + 1;
                const double dM = fma(-M1[s], w.inv_tau, S1[s]);               // dM/dt = Mdotfb - M/tvisc, and its derivative
                im.W[e] = wg[s]; im.F[e] = Ef[4 + s]; im.M[e] = M1[s]; im.D[e] = dM;
                im.D2[e] = (CURVES && all_qs) ? ratio[s] : fma(-dM, w.inv_tau, dS1[s]);
                im.R[e] = d1.rmu[s];
                if (e == keep) { im.C[0] = S1[s]; im.C[1] = dS1[s]; im.C[2] = iu1[s]; }
            }
            if (lead) {
                const double dM = fma(-M_s, w.inv_tau, cS0);
                im.W[0] = om_s; im.F[0] = cf0; im.M[0] = M_s; im.D[0] = dM;
                if (CURVES && all_qs) { const Vd<1> s0{{cS0}}; im.D2[0] = M_s * rcp_fast(s0)[0]; }
                else im.D2[0] = fma(-dM, w.inv_tau, cdS0);
            }
            __syncthreads();

            MP_PHASE(6)
            // ---------------- observations
            if constexpr (!CURVES) {
                // The model is only needed at the two grid points bracketing each observation.  The lane holding one of the
                // first 64 observations picks (Mdisc, omega) at those two points out of the tile's image as the tile goes by;
                // the luminosity stage runs after the last tile on those captured states (the 10 001-point light curve is
                // never formed).  Observations 64.. of a longer light curve are scored here and now, 64 at a time, while the
                // image of the tile that brackets them is in LDS: states -> luminosity -> np.interp -> chi^2 term.  (Until
                // round 3 their states were parked in per-walker rows in HBM and read back after the last tile: 95 MB of
                // traffic per 4 096-walker launch of BASELINE config 5, and an ordering constraint between launches.)
                if (deferred) {
                    const int p8 = 8 * ob_g;
                    if (ob_g >= 0 && p8 >= pos8 && p8 < end_kept8) {
                        image_state(sh, w, im, tt, kind, pos8, keep, t_s, p8, obM[0], obW[0]);
                        if (!on_knots) image_state(sh, w, im, tt, kind, pos8, keep, t_s, p8 + 8, obM[1], obW[1]);
                    }
                    if constexpr (LONG) {
                        if (long_lc) {
                            // observations 64.. whose interval starts inside the kept range (64-interval buckets of the dataset)
                            const int g_lo = pos8 >> 3, g_hi = end_kept8 >> 3;                 // grid intervals [g_lo, g_hi)
                            const int b_lo = g_lo / kTile64, b_hi = min((g_hi + kTile64 - 1) / kTile64, sh.n_tiles);
                            const int j0 = max(tptr[b_lo], 64), j1 = tptr[b_hi];
                            // (team kernels: the chunks of 64 observations are dealt to the team's wavefronts in turn -- the image is
                            // read-only until the next tile's flag exchange, behind which every wavefront has left this loop; the
                            // partial sums meet behind the last tile)
                            // The five values of an observation are asked for one chunk AHEAD of their use (round 5): read where they
                            // were needed, the grid interval was a global-memory round trip at the top of every chunk with nothing
                            // to cover it -- a walker on a 1 944-point light curve took twice the time of one on 50 points.
                            const int jstep = 64 * W;
                            int jb = j0 + 64 * wave;
                            int jj_n = dsd.obs_off + min(jb + lane, j1 - 1);
                            int g_n = 0;
                            double idt_n = 0.0, dx_n = 0.0, y_n = 0.0, ye_n = 1.0;
                            if (jb < j1) { g_n = sh.obs_g[jj_n]; idt_n = sh.obs_idt[jj_n]; dx_n = sh.obs_dx[jj_n]; y_n = sh.obs_y[jj_n]; ye_n = sh.obs_yerr[jj_n]; }
                            for (; jb < j1; jb += jstep) {                                     // (wave-uniform trip count)
                                const int j = jb + lane;
                                const int g = g_n;
                                const double o_idt = idt_n, o_dx = dx_n, o_y = y_n, o_ye = ye_n;
                                if (jb + jstep < j1) {                                         // (wave-uniform)
                                    jj_n = dsd.obs_off + min(jb + jstep + lane, j1 - 1);
                                    g_n = sh.obs_g[jj_n]; idt_n = sh.obs_idt[jj_n]; dx_n = sh.obs_dx[jj_n]; y_n = sh.obs_y[jj_n]; ye_n = sh.obs_yerr[jj_n];
                                }
                                const bool mine = j < j1 && g >= g_lo && g < g_hi;
                                const int q8 = mine ? 8 * g : pos8;                            // idle lanes: the tile's first interval
                                double Ma, Wa, Mb, Wb;
                                image_state(sh, w, im, tt, kind, pos8, keep, t_s, q8, Ma, Wa);
                                image_state(sh, w, im, tt, kind, pos8, keep, t_s, q8 + 8, Mb, Wb);
                                const Vd<2> Mx{{Ma, Mb}}, Wx{{Wa, Wb}};
                                const DiscPt<2> dx = disc_point(sh, w, Mx);
                                Vd<2> Lx, Lpx, Ldx;
                                luminosity(sh, w, dx, Wx, Lx, Lpx, Ldx);
                                const double mod = fma((Lx[1] - Lx[0]) * o_idt, o_dx, Lx[0]) / 1.0e50;
                                const double res = (o_y - mod) / o_ye;
                                if (mine) chi_long = fma(res, res, chi_long);
                            }
                        }
                    }
                }
            } else {
                // Curve outputs requested: the luminosity at every grid point of the kept steps.  A step over ns grid
                // intervals holds ns of them: the states at the skipped ones come from the step's Hermite interpolant
                // (the same states mode A would pick for an observation there); in the sub-stepped tiles every 8th step
                // end is a grid point.  The tile of the light curve is staged in LDS ([m] = the m-th grid point after the
                // tile's start point, [0] = the start point itself) for the interpolation at the observed times and leaves
                // for HBM from there with lane-contiguous addresses: every store instruction of the wavefront writes 512
                // consecutive bytes of the walker's row.
                const int ns = pre ? 1 : (d8 >> 3);                             // grid points per step (kinds 1 .. 4)
                const int g0 = pos8 >> 3;                                       // grid index of the tile's start
                const int n_here = pre ? keep >> 3 : keep * ns;                 // grid points this tile adds
                const size_t o0 = row + (size_t)g0 + 1;
                // Staging index: lanes write their steps' points ns * SPL doubles apart (a 32- or 64-way bank conflict as it
                // stands); XOR-ing the low five bits with the index of the 32-double block spreads them over the banks and
                // keeps the coalesced read-out (64 consecutive points) conflict-free.  Stays inside the block: no extra LDS.
                auto sw = [](int m) { return m ^ ((m >> 5) & 31); };
                // Dense output (round 4: the Hermite forms were evaluated from scratch at every skipped point before, ~220 VALU
                // instructions each; now the basis values, which depend on the wave-uniform position theta of the skipped point
                // only, are formed once per position):
                //   omega(theta)  = W0 + cD dW + c0 h F0 + c1 h F1                               (cubic Hermite)
                //   Mdisc(theta)  = M0 + s5 dM + b1 h D0 + b4 h D1 + b2 h^2 E0 + b5 h^2 E1       (quintic, steps shorter than tvisc)
                // which: 0 Ltot, 1 Lprop, 2 Ldip, 3 Mdisc, 4 omega -> stage[sw(m)], m = 1..n_here
                auto stage_curve = [&](int which, double *stage) {
                    for (int i = 1; i <= ns; ++i) {
                        Vd<kSPL> Mv = M1, Wv = wg;                              // i == ns: the step ends themselves
                        if (i < ns) {
                            const double th = wtab_theta(kind, i), t2 = th * th, t3 = t2 * th;
                            const double cD = fma(-2.0, t3, 3.0 * t2), c0 = fma(-2.0, t2, th) + t3, c1 = t3 - t2;
#pragma unroll
                            for (int s = 0; s < kSPL; ++s) {
                                const int J = e0 + s;                 // nodes J (step start) and J + 1 of the image
                                const double W0 = im.W[J];
                                Wv[s] = fma(cD, wg[s] - W0, fma(c0, h[s] * im.F[J], fma(c1, h[s] * Ef[4 + s], W0)));
                            }
                            if (all_qs) {
                                // every kept step of the tile is longer than tvisc: Mdisc = [cubic Lagrange of Mdisc/Mdotfb through the
                                // four nodes around the step; node values in the image's D2 array] x Mdotfb(t), the fallback rate
                                // from its binomial series about the step end: (1 + y)^(-5/3), |y| <= 1 - 1/Q <= 0.011, 6 terms: 1e-13
                                const double thm1 = th - 1.0;
#pragma unroll
                                for (int s = 0; s < kSPL; ++s) {
                                    const int J = e0 + s;
                                    const int l0 = max(min(J - 1, keep - 3), 0), variant = min(J - l0, 2);
                                    const int base = kWtabDense + ((kind - 2) * 7 + (i - 1)) * 12 + variant * 4;
                                    double r = 0.0;
#pragma unroll
                                    for (int k = 0; k < 4; ++k) r = fma(g_wtab[base + k], im.D2[l0 + k], r);
                                    const double y = thm1 * (h[s] * w.inv_tfb * iu1[s]);   // (theta - 1) h / (t_{J+1} + tfb)
                                    double P = fma(y, 26180.0 / 6561.0, -2618.0 / 729.0);
                                    P = fma(y, P, 770.0 / 243.0);
                                    P = fma(y, P, -220.0 / 81.0);
                                    P = fma(y, P, 20.0 / 9.0);
                                    P = fma(y, P, -5.0 / 3.0);
                                    P = fma(y, P, 1.0);
                                    Mv[s] = r * (S1[s] * P);
                                }
                            } else {
                                const double s5 = t3 * fma(th, fma(th, 6.0, -15.0), 10.0);          // the quintic's basis (hermite5)
                                const double b1 = fma(t3, fma(th, fma(th, -3.0, 8.0), -6.0), th);
                                const double b4 = t3 * fma(th, fma(th, -3.0, 7.0), -4.0);
                                const double b2 = 0.5 * t2 * fma(th, fma(th, fma(th, -1.0, 3.0), -3.0), 1.0);
                                const double b5 = 0.5 * t3 * fma(th, fma(th, 1.0, -2.0), 1.0);
                                bool any_qs = false;
#pragma unroll
                                for (int s = 0; s < kSPL; ++s) {
                                    const int J = e0 + s;
                                    const double hs = h[s], hh = hs * hs, M0 = im.M[J], dMs = M1[s] - M0;
                                    const double hD0 = hs * im.D[J], hD1 = hs * im.D[J + 1];
                                    const bool shortstep = hs * w.inv_tau < 1.0;
                                    const double quint = fma(s5, dMs, M0) + fma(b1, hD0, b4 * hD1) + hh * fma(b2, im.D2[J], b5 * im.D2[J + 1]);
                                    const double cub = fma(cD, dMs, fma(c0, hD0, fma(c1, hD1, M0)));
                                    Mv[s] = shortstep ? quint : cub;
                                    any_qs = any_qs || !(shortstep || keep < 3 || J >= keep);
                                }
                                if (__any(any_qs)) {   // the one tile in which the steps outgrow tvisc: node ratios formed on the spot
#pragma unroll
                                    for (int s = 0; s < kSPL; ++s) {
                                        const int J = e0 + s;
                                        if (!(h[s] * w.inv_tau < 1.0 || keep < 3 || J >= keep))
                                            Mv[s] = dense_mdisc_qs(w, im, kind, i, J, keep, fma(th - 1.0, h[s], t_s * tt.E[kind - 1][J + 1]));
                                    }
                                }
                            }
                        }
                        Vd<kSPL> val;
                        if (which <= 2) {
                            const DiscPt<kSPL> dv = i < ns ? disc_point(sh, w, Mv) : d1;
                            Vd<kSPL> Lt, Lp, Ld;
                            luminosity(sh, w, dv, Wv, Lt, Lp, Ld);
                            val = which == 0 ? Lt : (which == 1 ? Lp : Ld);
                        } else {
                            val = which == 3 ? Mv : Wv;
                        }
#pragma unroll
                        for (int s = 0; s < kSPL; ++s) {
                            const int e = e0 + s;                     // step index in the tile
                            if (pre) { if ((e & 7) == 7) stage[sw((e >> 3) + 1)] = val[s]; }
                            else stage[sw(e * ns + i)] = val[s];
                        }
                    }
                };
                // x / 1e50 correctly rounded without the division sequence (one per stored grid point): q = x r, then one
                // Newton correction with the exact remainder (r = fl(1e-50)); div == 1: the value itself
                auto store_curve = [&](double *dst, const double *stage, bool scale) {
                    for (int c = 0; c * 64 < n_here; ++c) {
                        const int m = c * 64 + lane;
                        if (m < n_here) {
                            const double v = stage[sw(m + 1)];
                            double q = v;
                            if (scale) { const double q0 = v * 1.0e-50; q = fma(fma(-q0, 1.0e50, v), 1.0e-50, q0); }
                            dst[o0 + m] = q;
                        }
                    }
                };
                stage_curve(0, Lbuf);
                if (lane == 0) Lbuf[sw(0)] = L_s;
                __syncthreads();
                if (a.ltot) store_curve(a.ltot, Lbuf, true);
                if (a.want_chi2) {
                    const int g_hi = end_kept8 >> 3;
                    if (ob_g >= g0 && ob_g < g_hi) {
                        const int m = ob_g - g0;
                        const double La = Lbuf[sw(m)], Lb = Lbuf[sw(m + 1)];
                        const double mod = fma((Lb - La) * ob_idt, ob_dx, La) / 1.0e50;   // np.interp, then /1e50
                        const double res = (ob_y - mod) / ob_ye;
                        chi = fma(res, res, chi);
                    }
                    if (long_lc) {
                        const int b_lo = g0 / kTile64, b_hi = min((g_hi + kTile64 - 1) / kTile64, sh.n_tiles);
                        const int j0 = max(tptr[b_lo], 64), j1 = tptr[b_hi];
                        for (int j = j0 + lane; j < j1; j += 64) {
                            const int jj = dsd.obs_off + j;
                            const int g = sh.obs_g[jj];
                            if (g < g0 || g >= g_hi) continue;
                            const int m = g - g0;
                            const double La = Lbuf[sw(m)], Lb = Lbuf[sw(m + 1)];
                            const double mod = fma((Lb - La) * sh.obs_idt[jj], sh.obs_dx[jj], La) / 1.0e50;
                            const double res = (sh.obs_y[jj] - mod) / sh.obs_yerr[jj];
                            chi = fma(res, res, chi);
                        }
                    }
                }
                L_s = Lbuf[sw(n_here)];
                __syncthreads();                                                // the staging area is reused below / by the next tile
                // the other curves (mp_model_lc only) go through the same staging area, one at a time
                auto put = [&](double *dst, int which, bool scale) {
                    if (!dst) return;                                           // wave-uniform
                    stage_curve(which, Lbuf);
                    __syncthreads();
                    store_curve(dst, Lbuf, scale);
                    __syncthreads();
                };
                put(a.lprop, 1, true);
                put(a.ldip, 2, true);
                put(a.mdisc, 3, false);
                put(a.omega, 4, false);
            }

            MP_PHASE(7)
            // ---------------- carry the end of the kept steps to the next tile
            if constexpr (kG >= 4) {
                // (all the reads of the hand-over issued together, the table value among them: one LDS round trip)
                const double tf_tab = tt.E[max(kind, 1) - 1][keep], Rk = im.R[keep];
                M_s = im.M[keep];
                om_s = im.W[keep];
                cf0 = im.F[keep];
                cS0 = im.C[0]; cdS0 = im.C[1]; ciu0 = im.C[2];
                t_s = t_s * (kind == 0 ? time_factor(sh, tt, 0, keep) : tf_tab);
                flags_s = branch_flags(w, Rk, om_s);
            } else {
                t_s = t_s * time_factor(sh, tt, kind, keep);
                M_s = im.M[keep];
                om_s = im.W[keep];
                cf0 = im.F[keep];
                flags_s = branch_flags(w, im.R[keep], om_s);
                cS0 = im.C[0]; cdS0 = im.C[1]; ciu0 = im.C[2];
            }
            rec_valid = true;
            rec_kind = kind;
            rec_sh8 = sh8;
            rec_J = keep;
            pos8 = end_kept8;
            kind = next_kind;
            MP_PHASE(12)
        }
        if (deferred && status == MP_STATUS_OK && on_knots) {   // the luminosity at the observations' own grid points
            const Vd<1> Mv{{obM[0]}}, Wv{{obW[0]}};
            const DiscPt<1> dp = disc_point(sh, w, Mv);
            Vd<1> Lt, Lp, Ld;
            luminosity(sh, w, dp, Wv, Lt, Lp, Ld);
            if (ob_g >= 0) {
                const int jj = dsd.obs_off + lane;
                const double res = (sh.obs_y[jj] - Lt[0] / 1.0e50) / sh.obs_yerr[jj];
                chi = res * res;
            }
        } else if (deferred && status == MP_STATUS_OK) {   // the luminosity evaluations of this walker: one per 64 observations
            const Vd<2> Mv{{obM[0], obM[1]}}, Wv{{obW[0], obW[1]}};
            const DiscPt<2> dp = disc_point(sh, w, Mv);
            Vd<2> Lt, Lp, Ld;
            luminosity(sh, w, dp, Wv, Lt, Lp, Ld);
            if (ob_g >= 0) {
                const int jj = dsd.obs_off + lane;
                ob_dx = sh.obs_dx[jj]; ob_idt = sh.obs_idt[jj]; ob_y = sh.obs_y[jj]; ob_ye = sh.obs_yerr[jj];
                const double mod = fma((Lt[1] - Lt[0]) * ob_idt, ob_dx, Lt[0]) / 1.0e50;   // np.interp, then /1e50
                const double res = (ob_y - mod) / ob_ye;
                chi = res * res;
            }
            if constexpr (W > 1 && LONG) {
                // the wavefronts' shares of the observations 64.., summed in a fixed order: every wavefront ends with the same bits
                tx->D4[wave][lane] = chi_long;
                __syncthreads();
                chi_long = tx->D4[0][lane];
#pragma unroll
                for (int k = 1; k < W; ++k) chi_long += tx->D4[k][lane];
            }
            chi += chi_long;                                         // observations 64.., scored tile by tile above
        }
    }

    if constexpr (CURVES) {
        // A walker that did not finish (prior, flag, non-finite) leaves NaN in every requested curve: the rows of a
        // device-pointer call are defined for every status, and the host entry needs no memset of the output.
        if (status != MP_STATUS_OK) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wavefront's earlier stores to the row have landed
            const double qnan = __longlong_as_double(0x7FF8000000000000ll);
            for (int i = lane; i < n_grid; i += 64) {
                if (a.ltot) a.ltot[row + i] = qnan;
                if (a.lprop) a.lprop[row + i] = qnan;
                if (a.ldip) a.ldip[row + i] = qnan;
                if (a.mdisc) a.mdisc[row + i] = qnan;
                if (a.omega) a.omega[row + i] = qnan;
            }
        }
    }

    double lnp = -INFINITY;
    if (status == MP_STATUS_OK) {
        lnp = -0.5 * wave_sum(chi);
        if (!isfinite(lnp)) { lnp = -INFINITY; status = MP_STATUS_NONFINITE; }
    }
    MP_PHASE(9)
    MP_PHASE_DUMP
    lnp_out = lnp;
    status_out = status;
    sweeps_out = sweeps_total;
    tiles_out = tiles_total;
}

}  // namespace mp
