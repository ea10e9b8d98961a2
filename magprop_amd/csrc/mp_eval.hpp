// mp_eval.hpp — evaluation of ONE walker's log-posterior on a wavefront (walker_eval; optionally fed by a producer
// wavefront, walker_produce): physics of the reference's RHS / luminosity stage in simplified algebra and the
// time-parallel exponential Adams-Moulton solver.  Included by mp_kernels.hip only.
#pragma once
#include "mp_math.hpp"

namespace mp {

// ---------------------------------------------------------------- per-walker constants
struct Walker {
    double inv_tau;   // 1/tvisc
    double S_amp;     // M0/tfb
    double inv_tfb;   // 1/tfb
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

// (Mdisc/tvisc)^(-1/7): the one expensive ingredient of disc_point (the producer wavefront of the two-wavefront
// kernel computes it and hands it over together with Mdisc)
template <int N>
MP_DEV Vd<N> disc_power(const Walker &w, const Vd<N> &Mdisc) {
    Vd<N> mdot;
    FORN mdot[i] = Mdisc[i] * w.inv_tau;
    return pow_m1_7_fast(mdot);                                     // mdot^(-1/7)
}

template <int N>
MP_DEV DiscPt<N> disc_from_power(const Walker &w, const Vd<N> &Mdisc, const Vd<N> &t) {
    DiscPt<N> p;
    FORN p.mdot[i] = Mdisc[i] * w.inv_tau;
    FORN {
        const double t2 = t[i] * t[i];
        p.rmu[i] = w.Crm * t2;                                      // Crm * mdot^(-2/7)
        p.squ[i] = w.sqrtCrm * t[i];                                // sqrt(rmu)
        p.qu[i] = w.Crm15 * (t2 * t[i]);                            // rmu^1.5 / sqrt(GM)
    }
    return p;
}

template <int N>
MP_DEV DiscPt<N> disc_point(const DevShared &sh, const Walker &w, const Vd<N> &Mdisc) {
    return disc_from_power(w, Mdisc, disc_power(w, Mdisc));
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
template <bool WANT_LAM, int N>
MP_DEV Vd<N> omega_rhs(const DevShared &sh, const Walker &w, const DiscPt<N> &p, const Vd<N> &om, Vd<N> &rot,
                       Vd<N> &lam) {
    const Flow<N> f = flow_state(w, sh.cfg.n_ode, p, om);
    Vd<N> out;
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

// ---------------------------------------------------------------- producer / consumer pair of wavefronts
// About half of a walker's work does not depend on omega: the Mdisc recurrence and the powers of Mdisc/tvisc the
// omega equation reads.  The two-wavefront kernel gives that half to a PRODUCER wavefront, which runs ahead tile by tile
// and hands (Mdisc, (Mdisc/tvisc)^(-1/7)) at every step end to the CONSUMER wavefront (predictor, Newton sweeps,
// observations) through a two-slot LDS ring.  No barriers: two counters in LDS (tiles produced / tiles consumed) with
// workgroup-scope release/acquire; the consumer raises `abort` when it stops early.  The arithmetic is that of the
// one-wavefront kernel, instruction for instruction, so the results are bit-identical.
template <int SPL>
struct PcRing {
    static constexpr int R = 2;
    double M[R][SPL][64];      // Mdisc at step end lane*SPL + s of the tile in slot r: [r][s][lane] (conflict-free)
    double T[R][SPL][64];      // (Mdisc/tvisc)^(-1/7) there
    int produced;              // tiles published by the producer
    int consumed;              // tiles taken over by the consumer
    int abort;                 // consumer -> producer: stop
};

constexpr int kSpinLimit = 1 << 20;   // x s_sleep(2): far beyond any legitimate wait; a hang becomes a failed walker

MP_DEV int ring_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
MP_DEV void ring_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// wait until *counter >= need; false if the partner raised abort or never came
MP_DEV bool ring_wait(const int *counter, int need, const int *abort) {
    for (int spin = 0; spin < kSpinLimit; ++spin) {
        if (ring_load(counter) >= need) return true;
        if (ring_load(abort)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// orders this wavefront's own LDS writes before its later LDS reads (all 64 lanes run in lockstep)
MP_DEV void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// The producer wavefront: the Mdisc phase of walker_eval (same statements), tile after tile.
template <int SPL>
MP_DEV void walker_produce(const DevShared &sh, const LaunchArgs &a, double (&par)[MP_MAX_NDIM], PcRing<SPL> &ring) {
    constexpr int kSPL = SPL, kTile = 64 * SPL, R = PcRing<SPL>::R;
    const int n_tiles = (sh.n_grid - 1 + kTile - 1) / kTile;
    const int lane = threadIdx.x & 63;
    const int nsteps = sh.n_grid - 1;
    Walker w;
    (void)walker_setup(sh, a, par, w);
    const double t0 = sh.tgrid[0];
    double t_s = t0;
    double M_s = par[2] * kMsol;
    double cS0, cS1, cS2;
    {
        const Vd<3> tg{{t0, t0 * sh.inv_q, t0 * sh.inv_q * sh.inv_q}};
        const Vd<3> Sg = mdot_fb(w, tg);
        cS0 = Sg[0]; cS1 = Sg[1]; cS2 = Sg[2];
    }
    double tb_next[kSPL];
#pragma unroll
    for (int s = 0; s < kSPL; ++s) tb_next[s] = sh.tgrid[min(lane * kSPL + s + 1, nsteps)];
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int i0 = tile * kTile + lane * kSPL;
        Vd<kSPL> tb, h;
#pragma unroll
        for (int s = 0; s < kSPL; ++s) {
            tb[s] = tb_next[s];
            tb_next[s] = sh.tgrid[min(i0 + kTile + s + 1, nsteps)];
        }
        {
            const double ta0 = lane_prev(tb[kSPL - 1], t_s);
#pragma unroll
            for (int s = 0; s < kSPL; ++s) h[s] = tb[s] - (s == 0 ? ta0 : tb[s - 1]);
        }
        Vd<kSPL> M1;
        double ES[kSPL + 3];
        {
            const Vd<kSPL> S1 = mdot_fb(w, tb);
#pragma unroll
            for (int s = 0; s < kSPL; ++s) ES[3 + s] = S1[s];
            ES[2] = lane_prev(ES[kSPL + 2], cS0);
            ES[1] = lane_prev(ES[kSPL + 1], cS1);
            ES[0] = lane_prev(ES[kSPL + 0], cS2);
            Vd<kSPL> zm, v0, v1, v2, v3;
#pragma unroll
            for (int s = 0; s < kSPL; ++s) {
                zm[s] = -h[s] * w.inv_tau;
                v0[s] = ES[3 + s]; v1[s] = ES[2 + s]; v2[s] = ES[1 + s]; v3[s] = ES[s];
            }
            const Phi<kSPL> pm = phi1234(zm);
            const Vd<kSPL> inc = eam4_increment(sh, pm, h, v0, v1, v2, v3);
            Vd<kSPL> am, bm;
            double A = 1.0, B = 0.0;
#pragma unroll
            for (int s = 0; s < kSPL; ++s) {
                am[s] = pm.e[s];
                bm[s] = inc[s];
                B = fma(am[s], B, bm[s]);
                A = A * am[s];
            }
            scan_affine(A, B);
            double Ax, Bx;
            lane_prev_map(A, B, Ax, Bx);
            double Mc = fma(Ax, M_s, Bx);
#pragma unroll
            for (int s = 0; s < kSPL; ++s) { Mc = fma(am[s], Mc, bm[s]); M1[s] = Mc; }
        }
        const Vd<kSPL> tp = disc_power(w, M1);
        // publish: the slot must have been taken over by the consumer first
        if (tile >= R && !ring_wait(&ring.consumed, tile - R + 1, &ring.abort)) return;
        if (ring_load(&ring.abort)) return;
        const int slot = tile % R;
#pragma unroll
        for (int s = 0; s < kSPL; ++s) { ring.M[slot][s][lane] = M1[s]; ring.T[slot][s][lane] = tp[s]; }
        ring_store(&ring.produced, tile + 1);            // release: the slot's contents are visible before the counter
        if (tile + 1 < n_tiles) {
            constexpr int e1 = kTile - 2, e2 = kTile - 3;
            cS0 = lane_bcast(ES[3 + kSPL - 1], 63);
            cS1 = lane_bcast(ES[3 + e1 % kSPL], e1 / kSPL);
            cS2 = lane_bcast(ES[3 + e2 % kSPL], e2 / kSPL);
            t_s = lane_bcast(tb[kSPL - 1], 63);
            M_s = lane_bcast(M1[kSPL - 1], 63);
        }
    }
}

constexpr int kMaxSweepsMargin = 16;  // sweeps allowed beyond the tile length (after which every step is exact)
// A tile whose iterates keep crossing the break-up limit (rotation parameter 0.27: the accretion torque switches off,
// code/synthetic_datasets/funcs.py:131-132) is chattering on that discontinuity and ends as a 'flag' anyway; after this
// many sweeps with an iterate beyond the limit the verdict is taken at once instead of after tile-length sweeps
// (a prior-wide launch used to last as long as its one chattering walker: 0.45 instead of 0.30 ms at 1 024 walkers).
// Not fewer: a 256-step tile that starts from a poor guess may overshoot the limit for a dozen sweeps and still converge
// below it (5 of the 32 768 soak walkers did with a threshold of 8).
constexpr int kChatterSweeps = 24;

// ---------------------------------------------------------------- the kernel
// Evaluate ONE walker on the calling wavefront (all 64 lanes enter with identical arguments).
// SPL = consecutive steps owned by one lane; a tile is 64*SPL steps.  par[] holds the sampler coordinates
// (prior checked and log-masked coordinates un-logged here unless a.physical); walker indexes ds_id and the
// optional curve outputs; Lbuf is the wave's LDS staging area [2*(64*SPL + 1)].
// ROLE 0: the whole evaluation on this wavefront.  ROLE 1: consumer of a PcRing (the Mdisc phase runs on the partner
// wavefront, walker_produce); the wavefront must then be the only user of Lbuf and must not meet workgroup barriers.
template <bool CURVES, int SPL, bool LONG, int ROLE = 0>
MP_DEV void walker_eval(const DevShared &sh, const LaunchArgs &a, int walker, double (&par)[MP_MAX_NDIM], double *Lbuf,
                        double &lnp_out, int &status_out, int &sweeps_out, PcRing<SPL> *ring = nullptr) {
    constexpr int kSPL = SPL, kTile = 64 * SPL, kMaxSweeps = kTile + kMaxSweepsMargin;
    const int n_tiles = (sh.n_grid - 1 + kTile - 1) / kTile;
    const int lane = threadIdx.x & 63;

    const int n_grid = sh.n_grid;
    const int nsteps = n_grid - 1;
    const size_t row = (size_t)walker * (size_t)n_grid;

    Walker w;
    int status = walker_setup(sh, a, par, w);

    // ---- state carried from tile to tile (all wave-uniform).  Index 0 = the tile's start point P0,
    // 1 = P0-1, 2 = P0-2: the history the multistep formulas reach back to.
    const double t0 = sh.tgrid[0];
    double t_s = t0;
    double M_s = par[2] * kMsol;                         // initial conditions, code/synthetic_datasets/funcs.py:66-69
    double om_s = (2.0 * M_PI) / (1.0e-3 * par[1]);
    double cS0, cS1, cS2;
    {
        const Vd<3> tg{{t0, t0 * sh.inv_q, t0 * sh.inv_q * sh.inv_q}};     // the grid continued backwards
        const Vd<3> Sg = mdot_fb(w, tg);
        cS0 = Sg[0]; cS1 = Sg[1]; cS2 = Sg[2];
    }
    double cf0, cf1, cf2, cw1 = om_s, cw2 = om_s, cw3 = om_s, cw4 = om_s;   // (omega_dot, omega) history; cw0 == om_s (cw3, cw4: predictor only)
    double L_s, Lp_s, Ld_s;
    {
        const Vd<1> Mv{{M_s}}, ov{{om_s}};
        const DiscPt<1> d_s = disc_point(sh, w, Mv);
        Vd<1> rot0, dummy, Lt0, Lp0, Ld0;
        cf0 = omega_rhs<false>(sh, w, d_s, ov, rot0, dummy)[0];
        cf1 = cf2 = cf0;
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
    const DsDesc dsd = ds_in_range ? sh.ds[dsid] : DsDesc{0, 0, 0, 0};
    if (a.want_chi2 && dsd.n_obs <= 0) status = MP_STATUS_BADDATASET;
    const int32_t *tptr = sh.tile_ptr + dsd.tile_off;
    // The first 64 observations of the walker's light curve live in registers, one per lane (time-sorted;
    // every synthetic set has 50).  Longer light curves park the states they need in the walker's scratch rows.
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
    // LONG: compiled with the scratch-row path for light curves of more than 64 points (the launcher picks this
    // variant when the handle holds such a dataset; the short variant keeps that code out of the register budget)
    const bool long_lc = (CURVES || LONG) && a.want_chi2 && dsd.n_obs > 64;
    const bool deferred = !CURVES && a.want_chi2;               // see "luminosity and chi^2" below
    const size_t sc_stride = (size_t)sh.scratch_stride;
    double *sc = sh.obs_scratch + (size_t)walker * 4 * sc_stride;   // [4][stride]: observations 64.. of this walker
    double obM[2] = {1.0e30, 1.0e30}, obW[2] = {1.0e3, 1.0e3};  // (Mdisc, omega) at the observation's bracketing grid points
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
            Vd<kSPL> tb, h;
#pragma unroll
            for (int s = 0; s < kSPL; ++s) {
                tb[s] = tb_next[s];
                tb_next[s] = sh.tgrid[min(i0 + kTile + s + 1, nsteps)];
            }
            {
                const double ta0 = lane_prev(tb[kSPL - 1], t_s);
#pragma unroll
                for (int s = 0; s < kSPL; ++s) h[s] = tb[s] - (s == 0 ? ta0 : tb[s - 1]);   // 0 for the padding steps of the last tile
            }

            // ---------------- Mdisc: exponential Adams-Moulton step (explicit: the source is known) + affine scan.
            // E*[k]: values at the three grid points before this lane's first step (k = 0,1,2) and at its step ends (k = 3+s).
            Vd<kSPL> M1;
            double ES[kSPL + 3];
            DiscPt<kSPL> d1;
            if constexpr (ROLE == 1) {
                // the partner wavefront has done this phase: take the tile over from the ring
                if (!ring_wait(&ring->produced, tile + 1, &ring->abort)) { status = MP_STATUS_NONFINITE; break; }
                const int slot = tile % PcRing<SPL>::R;
                Vd<kSPL> tp;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) { M1[s] = ring->M[slot][s][lane]; tp[s] = ring->T[slot][s][lane]; }
                wave_lds_fence();                                 // the values are in registers ...
                ring_store(&ring->consumed, tile + 1);            // ... before the slot is handed back
                d1 = disc_from_power(w, M1, tp);
#pragma unroll
                for (int s = 0; s < kSPL + 3; ++s) ES[s] = 0.0;
            } else {
                const Vd<kSPL> S1 = mdot_fb(w, tb);
#pragma unroll
                for (int s = 0; s < kSPL; ++s) ES[3 + s] = S1[s];
                ES[2] = lane_prev(ES[kSPL + 2], cS0);
                ES[1] = lane_prev(ES[kSPL + 1], cS1);
                ES[0] = lane_prev(ES[kSPL + 0], cS2);
                Vd<kSPL> zm, v0, v1, v2, v3;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    zm[s] = -h[s] * w.inv_tau;
                    v0[s] = ES[3 + s]; v1[s] = ES[2 + s]; v2[s] = ES[1 + s]; v3[s] = ES[s];
                }
                const Phi<kSPL> pm = phi1234(zm);
                const Vd<kSPL> inc = eam4_increment(sh, pm, h, v0, v1, v2, v3);
                Vd<kSPL> am, bm;
                double A = 1.0, B = 0.0;                  // composition of this lane's step maps
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
                d1 = disc_point(sh, w, M1);
            }

            // ---------------- omega: predictor = extrapolation of the last five grid values in the step index
            // (the grid is logarithmic, so power laws are smooth in the index) ...
            Vd<kSPL> wg;                                  // current guess of omega at this lane's step ends
            {
                // Newton backward-difference extrapolation (quartic once five grid values exist)
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
            int sweep = 0, over_sweeps = 0;
            Ew[2] = om_s;
            // A sweep that follows a small correction (every lane moved by < 1e-4) keeps the Jacobian lambda, e^{h lambda}
            // and the quadrature weights of the previous one and only re-evaluates omega_dot ("light" sweep): the scheme
            // may linearise about any nearby point, the result moves by ~1e-14, and the verification sweep costs a third less.
            // A sweep that follows a FULL sweep whose correction was below ultra_tol (1e-5 at the default sweep tolerance, 1e-7 at the strict one) does
            // not evaluate omega_dot at all: omega_dot at the new point is its linearisation about the previous one,
            // f + lambda*(omega_new - omega_old), exact to the second order in that correction (< 5e-9 relative in omega_dot even across
            // the propeller switch).  Such a sweep is the cheap verification pass of a tile whose first guess was good.
            Vd<kSPL> lam, ez;
            EamW<kSPL> cw;
            bool light = false, ultra = false;
            while (true) {
                ++sweep;
                if (!light) {   // (after a sweep that moved every lane by < 1e-4 the guesses are positive and finite)
                    bool wild = false;
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) wild = wild || !(wg[s] > 0.0);
                    if (__any(wild)) {   // keep the iteration alive after a wild or NaN guess (rare)
#pragma unroll
                        for (int s = 0; s < kSPL; ++s)
                            if (!(wg[s] > 0.0)) wg[s] = Ew[2] > 0.0 ? Ew[2] : om_s;
                    }
                }
                Vd<kSPL> rot, f1;
                const bool full = !light;                 // lambda, e^{h lambda} and the weights are renewed in this sweep
                if (ultra) {
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) {
                        f1[s] = fma(lam[s], wg[s] - Ew[3 + s], Ef[3 + s]);
                        rot[s] = sh.crot * (wg[s] * wg[s]);
                    }
                } else if (light) {
                    Vd<kSPL> unused;
                    f1 = omega_rhs<false>(sh, w, d1, wg, rot, unused);
                } else {
                    f1 = omega_rhs<true>(sh, w, d1, wg, rot, lam);
                }
                // largest rotation parameter among this lane's step ends (the padding steps of the last tile repeat the
                // last grid point once the first sweep has run; before that nothing is decided on them: `settled` is false)
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    Ef[3 + s] = f1[s];
                    Ew[3 + s] = wg[s];
                }
                const double rot_max = lane_max(rot.v);
                const bool flg = rot_max > 0.27;
                const bool near_limit = rot_max > 0.26;   // close to the break-up switch of the torque: no linearisation
                // break-up reached by an iterate that is no longer a wild guess: the reference's 'flag'
                flagged |= __ballot(settled && flg);
                const unsigned long long over_now = __ballot(flg);
                over_sweeps += over_now != 0ull;
                double h1 = cf1, h2 = cf2, u1 = cw1, u2 = cw2;
                if (tile == 0) {   // start-up: the two points before the grid continue points 0 and 1 linearly in the index
                    const double fp1 = lane_bcast(Ef[3], 0), wp1 = lane_bcast(Ew[3], 0);
                    h1 = 2.0 * cf0 - fp1; u1 = 2.0 * om_s - wp1;
                    h2 = 3.0 * cf0 - 2.0 * fp1; u2 = 3.0 * om_s - 2.0 * wp1;
                }
                Ef[2] = lane_prev(Ef[kSPL + 2], cf0);  Ew[2] = lane_prev(Ew[kSPL + 2], om_s);
                Ef[1] = lane_prev(Ef[kSPL + 1], h1);   Ew[1] = lane_prev(Ew[kSPL + 1], u1);
                Ef[0] = lane_prev(Ef[kSPL + 0], h2);   Ew[0] = lane_prev(Ew[kSPL + 0], u2);
                Vd<kSPL> n0, n1, n2, n3;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    n0[s] = fma(-lam[s], Ew[3 + s], Ef[3 + s]);
                    n1[s] = fma(-lam[s], Ew[2 + s], Ef[2 + s]);
                    n2[s] = fma(-lam[s], Ew[1 + s], Ef[1 + s]);
                    n3[s] = fma(-lam[s], Ew[s], Ef[s]);
                }
                if (!light) {
                    Vd<kSPL> zw;
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) zw[s] = h[s] * lam[s];
                    const Phi<kSPL> pw_ = phi1234(zw);
                    ez = pw_.e;
                    cw = eam4_node_weights(sh, pw_);
                }
                const Vd<kSPL> inc = eam4_increment_nodes(cw, h, n0, n1, n2, n3);
                Vd<kSPL> aw, bw;
                double A = 1.0, B = 0.0;
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
                double wc = fma(Ax, om_s, Bx);           // omega at this lane's first step start
                // The convergence tests look at the lane's MEAN correction (its steps are consecutive and move together):
                // one set of comparisons per lane instead of one per step.  NaN propagates through the sum: "not converged".
                double dsum = 0.0;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    wc = fma(aw[s], wc, bw[s]);
                    dsum += fabs(wc - wg[s]);
                    wg[s] = wc;
                }
                const double mag = (double)kSPL * fabs(wc);
                const bool all_settled = dsum <= 1.0e-3 * mag;                       // false for NaN
                const bool all_small = dsum <= 1.0e-4 * mag;
                const bool all_tiny = dsum <= sh.ultra_tol * mag;
                const bool all_ok = dsum <= sh.sweep_tol * mag;
                settled = all_settled;
                light = __all(all_small);
                // (lambda is the derivative at the point this sweep evaluated; the linearisation cannot see the break-up
                // discontinuity of the accretion torque, so tiles that come near it keep evaluating omega_dot)
                ultra = full && __all(all_tiny && !near_limit);
                pending = __ballot(!all_ok);
                if (pending == 0ull || flagged != 0ull || sweep >= kMaxSweeps) break;
                if (over_sweeps >= kChatterSweeps) { flagged |= over_now ? over_now : pending; break; }
            }
            sweeps_total += sweep;

            // ---------------- failure detection in time order (oracle/mp_oracle.c).  Per lane: a NaN or an
            // infinity anywhere shows in the sum, a non-positive value in the minimum, the break-up limit in the largest
            // omega (the padding steps of the last tile repeat the last grid point).
            {
                double vsum = M1[0] + wg[0];
#pragma unroll
                for (int s = 1; s < kSPL; ++s) vsum += M1[s] + wg[s];
                const double vmin = min_raw(lane_min(M1.v), lane_min(wg.v)), wmax = lane_max(wg.v);
                const bool bad = !isfinite(vsum) || !(vmin > 0.0);
                const bool over = sh.crot * wmax * wmax > 0.27;
                const unsigned long long mb = __ballot(bad);
                // a step whose sweeps never settle is chattering on the Nacc discontinuity: same verdict as a flag
                const unsigned long long mf = flagged | __ballot(over) | (flagged ? 0ull : pending);
                if (mb | mf) {
                    const int first = __ffsll((unsigned long long)(mb | mf)) - 1;
                    status = ((mf >> first) & 1ull) ? MP_STATUS_FLAG : MP_STATUS_NONFINITE;
                    break;
                }
            }

            // ---------------- luminosity and chi^2
            const bool mine = ob_tile == tile;
            if constexpr (!CURVES) {
                // The model is only needed at the two grid points bracketing each observation.  The lane holding an
                // observation picks (Mdisc, omega) at those two points out of the tile's LDS image as the tile goes by
                // (observations beyond the 64 register-resident ones: into the walker's scratch rows); the luminosity
                // stage runs after the last tile on those captured states, once per 64 observations (the 10 001-point
                // light curve is never formed).
                int j0 = 0, j1 = 0;
                if (long_lc) { j0 = max(tptr[tile * kSPL], 64); j1 = tptr[min((tile + 1) * kSPL, sh.n_tiles)]; }   // 64-step buckets
                if (deferred && (__any(mine) || j1 > j0)) {
                    double *Mbuf = Lbuf, *Wbuf = Lbuf + kTile + 1;
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) { Mbuf[lane * kSPL + s + 1] = M1[s]; Wbuf[lane * kSPL + s + 1] = wg[s]; }
                    if (lane == 0) { Mbuf[0] = M_s; Wbuf[0] = om_s; }
                    if constexpr (ROLE == 1) wave_lds_fence(); else __syncthreads();
                    if (mine) {
                        const int g = ob_g - tile * kTile;
                        obM[0] = Mbuf[g]; obM[1] = Mbuf[g + 1];
                        obW[0] = Wbuf[g]; obW[1] = Wbuf[g + 1];
                    }
                    for (int j = j0 + lane; j < j1; j += 64) {
                        const int g = sh.obs_g[dsd.obs_off + j] - tile * kTile;
                        double *p = sc + (j - 64);
                        p[0] = Mbuf[g]; p[sc_stride] = Mbuf[g + 1];
                        p[2 * sc_stride] = Wbuf[g]; p[3 * sc_stride] = Wbuf[g + 1];
                    }
                    if constexpr (ROLE == 1) wave_lds_fence(); else __syncthreads();
                }
            } else {   // curve outputs requested: the whole light curve is formed anyway
                int j0 = 0, j1 = 0;
                if (long_lc) { j0 = max(tptr[tile * kSPL], 64); j1 = tptr[min((tile + 1) * kSPL, sh.n_tiles)]; }   // 64-step buckets
                const bool tile_has_obs = __any(mine) || j1 > j0;
                Vd<kSPL> Lt, Lp, Ld;
                luminosity(sh, w, d1, wg, Lt, Lp, Ld);
                // The tile of the light curve is staged in LDS ([e + 1] = step end e, [0] = the tile's start point) for
                // the interpolation and leaves for HBM from there with lane-contiguous addresses: every store
                // instruction of the wavefront writes 512 consecutive bytes of the walker's row.
                const int n_here = min(kTile, nsteps - tile * kTile);          // step ends of this tile that exist
                const size_t o0 = row + (size_t)tile * kTile + 1;
                double *S2 = Lbuf + kTile + 1;                                  // second staging area
#pragma unroll
                for (int s = 0; s < kSPL; ++s) Lbuf[lane * kSPL + s + 1] = Lt[s];
                if (lane == 0) Lbuf[0] = L_s;
                __syncthreads();
                if (a.ltot) {
#pragma unroll
                    for (int c = 0; c < kSPL; ++c) {
                        const int e = c * 64 + lane;
                        if (e < n_here) a.ltot[o0 + e] = Lbuf[e + 1] / 1.0e50;
                    }
                }
                // the other curves (mp_model_lc only) go through the second staging area, one at a time
                auto put = [&](double *dst, const Vd<kSPL> &v, double div) {
                    if (!dst) return;                                           // wave-uniform
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) S2[lane * kSPL + s] = v[s];
                    __syncthreads();
#pragma unroll
                    for (int c = 0; c < kSPL; ++c) {
                        const int e = c * 64 + lane;
                        if (e < n_here) dst[o0 + e] = S2[e] / div;
                    }
                    __syncthreads();
                };
                put(a.lprop, Lp, 1.0e50);
                put(a.ldip, Ld, 1.0e50);
                put(a.mdisc, M1, 1.0);
                put(a.omega, wg, 1.0);
                if (tile_has_obs) {
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
                }
                __syncthreads();                                                // the next tile overwrites the staging area
                L_s = lane_bcast(Lt[kSPL - 1], 63);
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
            }
        }
        if (deferred && status == MP_STATUS_OK) {   // the luminosity evaluations of this walker: one per 64 observations
            const Vd<2> Mv{{obM[0], obM[1]}}, Wv{{obW[0], obW[1]}};
            const DiscPt<2> dp = disc_point(sh, w, Mv);
            Vd<2> Lt, Lp, Ld;
            luminosity(sh, w, dp, Wv, Lt, Lp, Ld);
            if (ob_g >= 0) {
                const double mod = fma((Lt[1] - Lt[0]) * ob_idt, ob_dx, Lt[0]) / 1.0e50;   // np.interp, then /1e50
                const double res = (ob_y - mod) / ob_ye;
                chi = res * res;
            }
            if (long_lc) {
                if constexpr (ROLE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else __syncthreads();   // scratch rows written by other lanes
                for (int jb = 64; jb < dsd.n_obs; jb += 64) {
                    const bool valid = jb + lane < dsd.n_obs;
                    const int j = valid ? jb + lane : dsd.n_obs - 1;
                    const double *p = sc + (j - 64);
                    const Vd<2> Mx{{p[0], p[sc_stride]}}, Wx{{p[2 * sc_stride], p[3 * sc_stride]}};
                    const DiscPt<2> dx = disc_point(sh, w, Mx);
                    Vd<2> Lx, Lpx, Ldx;
                    luminosity(sh, w, dx, Wx, Lx, Lpx, Ldx);
                    const int jj = dsd.obs_off + j;
                    const double mod = fma((Lx[1] - Lx[0]) * sh.obs_idt[jj], sh.obs_dx[jj], Lx[0]) / 1.0e50;
                    const double res = (sh.obs_y[jj] - mod) / sh.obs_yerr[jj];
                    if (valid) chi = fma(res, res, chi);
                }
            }
        }
    }

    if constexpr (ROLE == 1) ring_store(&ring->abort, 1);   // done (or failed): release the producer

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
    lnp_out = lnp;
    status_out = status;
    sweeps_out = sweeps_total;
}

}  // namespace mp
