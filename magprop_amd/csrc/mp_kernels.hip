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
#include <type_traits>

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

// ---------------------------------------------------------------- N independent values per lane
// Every lane owns N consecutive time steps.  All per-step mathematics below is written on N-vectors, one
// statement at a time across the N steps, so that the N dependent chains (Horner polynomials, Newton
// refinements) sit next to each other in the instruction stream and hide each other's latency: with one
// wave per SIMD there is no other wave to do it.
template <int N>
struct Vd {
    double v[N];
    MP_DEV double &operator[](int i) { return v[i]; }
    MP_DEV const double &operator[](int i) const { return v[i]; }
};
template <int N>
struct Vb {
    bool v[N];
    MP_DEV bool &operator[](int i) { return v[i]; }
    MP_DEV const bool &operator[](int i) const { return v[i]; }
};
#define FORN _Pragma("unroll") for (int i = 0; i < N; ++i)

// ---------------------------------------------------------------- fp64 elementary functions
// Three-address FMA for Horner chains.  hipcc selects the two-address v_fmac_f64 there and then has to
// copy every polynomial coefficient into the accumulator first (one v_mov_b64 per term); the explicit
// v_fma_f64 reads the coefficient in place.  Only plain VALU results feed it (no transcendental-op hazard).
MP_DEV double fma3(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// p <- p*x + c on all N chains
template <int N>
MP_DEV void horner(Vd<N> &p, const Vd<N> &x, double c) {
    FORN p[i] = fma3(p[i], x[i], c);
}

// Hand-rolled for this kernel's argument ranges (positive, normal, far from overflow): hardware
// seed (v_rcp_f64 / v_rsq_f64, ~2^-23) + two Newton steps, without the scaling / fix-up code the
// general-purpose library versions carry.  All are accurate to ~1-2 ulp.
template <int N>
MP_DEV Vd<N> rcp_fast(const Vd<N> &x) {
    Vd<N> r, e;
    FORN r[i] = __builtin_amdgcn_rcp(x[i]);
    FORN e[i] = fma(-x[i], r[i], 1.0);
    FORN r[i] = fma(e[i], r[i], r[i]);
    FORN e[i] = fma(-x[i], r[i], 1.0);
    FORN r[i] = fma(e[i], r[i], r[i]);
    return r;
}

template <int N>
MP_DEV Vd<N> rsqrt_fast(const Vd<N> &x) {
    Vd<N> y, hx, t;
    FORN y[i] = __builtin_amdgcn_rsq(x[i]);
    FORN hx[i] = 0.5 * x[i];
    FORN t[i] = hx[i] * y[i];
    FORN t[i] = fma(-t[i], y[i], 0.5);
    FORN y[i] = fma(y[i], t[i], y[i]);
    FORN t[i] = hx[i] * y[i];
    FORN t[i] = fma(-t[i], y[i], 0.5);
    FORN y[i] = fma(y[i], t[i], y[i]);
    return y;
}

// e^x for x in [-750, 700]; underflows cleanly to 0 below
template <int N>
MP_DEV Vd<N> exp_fast(const Vd<N> &x) {
    Vd<N> k, r, p;
    FORN k[i] = __builtin_rint(x[i] * 1.4426950408889634074);
    FORN r[i] = fma(k[i], -6.93147180369123816490e-01, x[i]);
    FORN r[i] = fma(k[i], -1.90821492927058770002e-10, r[i]);
    FORN p[i] = 1.0 / 479001600.0;               // Taylor degree 12 on |r| <= ln2/2: 1.7e-16
    horner(p, r, 1.0 / 39916800.0);
    horner(p, r, 1.0 / 3628800.0);
    horner(p, r, 1.0 / 362880.0);
    horner(p, r, 1.0 / 40320.0);
    horner(p, r, 1.0 / 5040.0);
    horner(p, r, 1.0 / 720.0);
    horner(p, r, 1.0 / 120.0);
    horner(p, r, 1.0 / 24.0);
    horner(p, r, 1.0 / 6.0);
    horner(p, r, 0.5);
    horner(p, r, 1.0);
    horner(p, r, 1.0);
    FORN p[i] = ldexp(p[i], (int)k[i]);
    return p;
}

// x^(-1/3) for positive normal x within float range: v_log_f32/v_exp_f32 seed (~1e-6) + two Newton steps
// y <- y (4 - x y^3)/3 (error -> 2 e^2): ~1 ulp.
template <int N>
MP_DEV Vd<N> rcbrt_fast(const Vd<N> &x) {
    Vd<N> y, x3, y3;
    FORN y[i] = (double)__builtin_amdgcn_exp2f(-0.33333333f * __builtin_amdgcn_logf((float)x[i]));
    FORN x3[i] = x[i] * (1.0 / 3.0);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        FORN y3[i] = y[i] * y[i];
        FORN y3[i] = y3[i] * y[i];
        FORN y3[i] = fma(-x3[i], y3[i], 4.0 / 3.0);
        FORN y[i] = y[i] * y3[i];
    }
    return y;
}

// x^(-2/7) for positive normal x within float range: y = (x^2)^(-1/7), Newton y <- y (8 - x^2 y^7)/7 (error -> 4 e^2)
template <int N>
MP_DEV Vd<N> pow_m2_7_fast(const Vd<N> &x) {
    Vd<N> y, z7, y2, y4, y7;
    FORN y[i] = (double)__builtin_amdgcn_exp2f(-0.28571429f * __builtin_amdgcn_logf((float)x[i]));
    FORN z7[i] = (x[i] * x[i]) * (1.0 / 7.0);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        FORN y2[i] = y[i] * y[i];
        FORN y4[i] = y2[i] * y2[i];
        FORN y7[i] = y4[i] * y2[i];
        FORN y7[i] = y7[i] * y[i];
        FORN y7[i] = fma(-z7[i], y7[i], 8.0 / 7.0);
        FORN y[i] = y[i] * y7[i];
    }
    return y;
}

// ---------------------------------------------------------------- phi functions
// phi_j(z) = sum_k z^k/(k+j)!  : phi_1 = (e^z-1)/z, phi_{j+1} = (phi_j - 1/j!)/z
template <int N>
struct Phi {
    Vd<N> e, p1, p2, p3, p4;
};

template <int N>
MP_DEV Phi<N> phi1234(const Vd<N> &z) {
    // Taylor series of phi_4: 7 terms when every |z| of the wavefront is below 1/32 (< 2e-17 relative), 13 terms
    // for |z| < 1/2, closed forms elsewhere
    Vd<N> s;
    bool all_tiny = true;
    FORN all_tiny = all_tiny && fabs(z[i]) < 0.03125;
    if (__all(all_tiny)) {
        FORN s[i] = 1.0 / 3628800.0;              // 1/10!
        horner(s, z, 1.0 / 362880.0);             // 1/9!
        horner(s, z, 1.0 / 40320.0);              // 1/8!
        horner(s, z, 1.0 / 5040.0);               // 1/7!
        horner(s, z, 1.0 / 720.0);                // 1/6!
        horner(s, z, 1.0 / 120.0);                // 1/5!
        horner(s, z, 1.0 / 24.0);                 // 1/4!
    } else {
        FORN s[i] = 1.0 / 20922789888000.0;       // 1/16!
        horner(s, z, 1.0 / 1307674368000.0);      // 1/15!
        horner(s, z, 1.0 / 87178291200.0);        // 1/14!
        horner(s, z, 1.0 / 6227020800.0);         // 1/13!
        horner(s, z, 1.0 / 479001600.0);          // 1/12!
        horner(s, z, 1.0 / 39916800.0);           // 1/11!
        horner(s, z, 1.0 / 3628800.0);            // 1/10!
        horner(s, z, 1.0 / 362880.0);             // 1/9!
        horner(s, z, 1.0 / 40320.0);              // 1/8!
        horner(s, z, 1.0 / 5040.0);               // 1/7!
        horner(s, z, 1.0 / 720.0);                // 1/6!
        horner(s, z, 1.0 / 120.0);                // 1/5!
        horner(s, z, 1.0 / 24.0);                 // 1/4!
    }
    Phi<N> r;
    r.p4 = s;
    FORN r.p3[i] = fma(z[i], s[i], 1.0 / 6.0);
    FORN r.p2[i] = fma(z[i], r.p3[i], 0.5);
    FORN r.p1[i] = fma(z[i], r.p2[i], 1.0);
    FORN r.e[i] = fma(z[i], r.p1[i], 1.0);
    Vb<N> big;
    bool any_big = false;
    FORN { big[i] = !(fabs(z[i]) < 0.5); any_big = any_big || big[i]; }
    if (__any(any_big)) {                          // wave-uniform: only stiff / late-time tiles pay for this
        Vd<N> zc, zs;
        FORN zc[i] = fmax(z[i], -750.0);
        FORN zs[i] = big[i] ? z[i] : 1.0;
        const Vd<N> ce = exp_fast(zc);
        const Vd<N> rz = rcp_fast(zs);
        FORN {
            const double c1 = (ce[i] - 1.0) * rz[i];
            const double c2 = (c1 - 1.0) * rz[i];
            const double c3 = (c2 - 0.5) * rz[i];
            const double c4 = (c3 - 1.0 / 6.0) * rz[i];
            r.e[i] = big[i] ? ce[i] : r.e[i];
            r.p1[i] = big[i] ? c1 : r.p1[i];
            r.p2[i] = big[i] ? c2 : r.p2[i];
            r.p3[i] = big[i] ? c3 : r.p3[i];
            r.p4[i] = big[i] ? c4 : r.p4[i];
        }
    }
    return r;
}

// h * int_0^1 e^{z(1-theta)} P(theta) dtheta for the cubic P through the node values v0..v3 at
// t_{j+1}, t_j, t_{j-1}, t_{j-2} (quadrature matrix W of the geometric grid, DevShared::eamW)
template <int N>
MP_DEV Vd<N> eam4_increment(const DevShared &sh, const Phi<N> &p, const Vd<N> &h, const Vd<N> &v0, const Vd<N> &v1,
                            const Vd<N> &v2, const Vd<N> &v3) {
    Vd<N> acc, g;
    FORN acc[i] = 0.0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const Vd<N> &ph = m == 0 ? p.p1 : m == 1 ? p.p2 : m == 2 ? p.p3 : p.p4;
        FORN g[i] = sh.eamW[3][m] * v3[i];
        FORN g[i] = fma(sh.eamW[2][m], v2[i], g[i]);
        FORN g[i] = fma(sh.eamW[1][m], v1[i], g[i]);
        FORN g[i] = fma(sh.eamW[0][m], v0[i], g[i]);
        FORN acc[i] = fma(ph[i], g[i], acc[i]);
    }
    FORN acc[i] = h[i] * acc[i];
    return acc;
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
    const Vd<N> pw_ = pow_m2_7_fast(p.mdot);
    FORN p.rmu[i] = w.Crm * pw_[i];                                 // Crm * mdot^(-2/7)
    const Vd<N> rs = rsqrt_fast(p.rmu);
    FORN p.squ[i] = p.rmu[i] * rs[i];
    FORN p.qu[i] = p.rmu[i] * p.squ[i] * sh.inv_sqrtGM;
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
    bool saturated = true;                                          // |x| > 19.5: tanh(x) = +-1 to the last bit
    FORN saturated = saturated && fabs(x[i]) > 19.5;
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
            if (i < a.ndim && ((sh.log_mask >> i) & 1u)) par[i] = pow(10.0, par[i]);
    }

    // ---- walker constants (code/synthetic_datasets/funcs.py:98-102)
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

    return status;
}

constexpr int kMaxSweepsMargin = 16;  // sweeps allowed beyond the tile length (after which every step is exact)

// ---------------------------------------------------------------- the kernel
// Evaluate ONE walker on the calling wavefront (all 64 lanes enter with identical arguments).
// SPL = consecutive steps owned by one lane; a tile is 64*SPL steps.  par[] holds the sampler coordinates
// (prior checked and log-masked coordinates un-logged here unless a.physical); walker indexes ds_id and the
// optional curve outputs; Lbuf is the wave's LDS tile [64*SPL + 1].
template <bool CURVES, int SPL>
MP_DEV void walker_eval(const DevShared &sh, const LaunchArgs &a, int walker, double (&par)[MP_MAX_NDIM], double *Lbuf,
                        double &lnp_out, int &status_out, int &sweeps_out) {
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
    bool L_valid = true;          // L_s holds the luminosity at the current tile start
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
        luminosity(sh, w, d_s, ov, Lt0, Lp0, Ld0);
        L_s = Lt0[0]; Lp_s = Lp0[0]; Ld_s = Ld0[0];
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
            Vd<kSPL> tb, h;
            Vb<kSPL> active;
#pragma unroll
            for (int s = 0; s < kSPL; ++s) {
                tb[s] = tb_next[s];
                tb_next[s] = sh.tgrid[min(i0 + kTile + s + 1, nsteps)];
                active[s] = i0 + s < nsteps;
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
                double A = 1.0, B = 0.0;                  // composition of this lane's step maps
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    am[s] = pm.e[s];      // padding steps: h = 0 -> e = 1, inc = 0
                    bm[s] = inc[s];
                    B = fma(am[s], B, bm[s]);
                    A = A * am[s];
                }
                scan_affine(A, B);
                const double Ax = lane_prev(A, 1.0), Bx = lane_prev(B, 0.0);   // exclusive prefix
                double Mc = fma(Ax, M_s, Bx);            // Mdisc at this lane's first step start
#pragma unroll
                for (int s = 0; s < kSPL; ++s) { Mc = fma(am[s], Mc, bm[s]); M1[s] = Mc; }
            }
            const DiscPt<kSPL> d1 = disc_point(sh, w, M1);

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
            int sweep = 0;
            Ew[2] = om_s;
            while (true) {
                ++sweep;
                {
                    bool wild = false;
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) wild = wild || !(wg[s] > 0.0);
                    if (__any(wild)) {   // keep the iteration alive after a wild or NaN guess (rare)
#pragma unroll
                        for (int s = 0; s < kSPL; ++s)
                            if (!(wg[s] > 0.0)) wg[s] = Ew[2] > 0.0 ? Ew[2] : om_s;
                    }
                }
                Vd<kSPL> rot, lam;
                const Vd<kSPL> f1 = omega_rhs<true>(sh, w, d1, wg, rot, lam);
                bool flg = false;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    Ef[3 + s] = f1[s];
                    Ew[3 + s] = wg[s];
                    flg = flg || (active[s] && rot[s] > 0.27);
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
                Vd<kSPL> zw, n0, n1, n2, n3;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    zw[s] = h[s] * lam[s];
                    n0[s] = fma(-lam[s], Ew[3 + s], Ef[3 + s]);
                    n1[s] = fma(-lam[s], Ew[2 + s], Ef[2 + s]);
                    n2[s] = fma(-lam[s], Ew[1 + s], Ef[1 + s]);
                    n3[s] = fma(-lam[s], Ew[s], Ef[s]);
                }
                const Phi<kSPL> pw_ = phi1234(zw);
                const Vd<kSPL> inc = eam4_increment(sh, pw_, h, n0, n1, n2, n3);
                Vd<kSPL> aw, bw;
                double A = 1.0, B = 0.0;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    aw[s] = pw_.e[s];
                    bw[s] = inc[s];
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
                    all_ok = all_ok && dw <= sh.sweep_tol * mag;
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
            // (tiles that hold no observation and are not written out skip the luminosity stage altogether)
            const bool mine = ob_tile == tile;
            int j0 = 0, j1 = 0;
            if (long_lc) { j0 = max(tptr[tile * kSPL], 64); j1 = tptr[min((tile + 1) * kSPL, sh.n_tiles)]; }   // 64-step buckets
            const bool tile_has_obs = __any(mine) || j1 > j0;
            Vd<kSPL> Lt, Lp, Ld;
            if (CURVES || tile_has_obs) luminosity(sh, w, d1, wg, Lt, Lp, Ld);
#pragma unroll
            for (int s = 0; s < kSPL; ++s) {
                if (CURVES && active[s]) {
                    const size_t o = row + (size_t)(i0 + s) + 1;
                    if (a.ltot) a.ltot[o] = Lt[s] / 1.0e50;
                    if (a.lprop) a.lprop[o] = Lp[s] / 1.0e50;
                    if (a.ldip) a.ldip[o] = Ld[s] / 1.0e50;
                    if (a.mdisc) a.mdisc[o] = M1[s];
                    if (a.omega) a.omega[o] = wg[s];
                }
            }
            {
                if (tile_has_obs) {
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) Lbuf[lane * kSPL + s + 1] = Lt[s];
                    if (!L_valid) {   // the previous tile skipped its luminosity stage: evaluate its end point now
                        const Vd<1> Mv{{M_s}}, ov{{om_s}};
                        const DiscPt<1> dps = disc_point(sh, w, Mv);
                        Vd<1> l0, l1, l2;
                        luminosity(sh, w, dps, ov, l0, l1, l2);
                        L_s = l0[0];
                        L_valid = true;
                    }
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
                L_valid = CURVES || tile_has_obs;
                if (L_valid) L_s = lane_bcast(Lt[kSPL - 1], 63);
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

// ---------------------------------------------------------------- W wavefronts per walker
// Small batches cannot give every SIMD a walker (256 CUs x 4 SIMDs): the stretch move only ever has half an
// ensemble in flight, and the reference's own configuration has 24 walkers.  walker_eval_mw spreads ONE walker
// over the W wavefronts of a 64*W-thread workgroup: a tile is 64*W*SPL steps, every lane still owns SPL
// consecutive steps, the wavefront scans stay in DPP, and what has to cross wavefronts goes through LDS:
//   - the step history (Mdotfb, omega_dot, omega at the three previous grid points) is read from an LDS image
//     of the tile instead of the neighbouring lane;
//   - the affine scan is completed with the per-wavefront totals;
//   - loop exits are agreed with __syncthreads_or.
// Same scheme, same arithmetic per step as walker_eval; only the tile length differs (results agree to
// rounding, like the SPL variants).  lds: [2*(kTile+3) + 2*W + (kTile+1) + 16] doubles, see MwLds.
template <int SPL, int W>
struct MwLds {
    static constexpr int kTile = 64 * W * SPL;
    double s[kTile + 3];        // Mdotfb at step ends e = -3..kTile-1, stored at [e + 3]
    double f[kTile + 3];        // omega_dot
    double w[kTile + 3];        // omega
    double tot[2][2 * W];       // per-wavefront scan totals (a, b); double-buffered by use
    int flags[2][W];            // per-wavefront (pending | flagged << 1); double-buffered by sweep
    double L[2][kTile + 1];     // model light curve of the tile; double-buffered by tile
    double carry[2][16];        // tile-end state for the next tile; double-buffered by tile
    int fail[2][2 * W];         // per-wavefront first non-finite / first over-limit lane; double-buffered by tile
};

// Complete a wavefront-level affine scan across the W wavefronts of the workgroup: x_wave = value at this
// wavefront's first step start, given the tile's start value x0.  One barrier.
template <int W>
MP_DEV void scan_affine_block(double &A, double &B, double (&tot)[2 * W], int wave, int lane, double x0, double &x_wave) {
    scan_affine(A, B);
    if (lane == 63) { tot[2 * wave] = A; tot[2 * wave + 1] = B; }
    __syncthreads();
    double xw = x0;
#pragma unroll
    for (int v = 0; v < W; ++v)
        if (v < wave) xw = fma(tot[2 * v], xw, tot[2 * v + 1]);
    x_wave = xw;
}

template <int SPL, int W>
MP_DEV void walker_eval_mw(const DevShared &sh, const LaunchArgs &a, int walker, double (&par)[MP_MAX_NDIM], MwLds<SPL, W> &lds,
                           double &lnp_out, int &status_out, int &sweeps_out) {
    constexpr int kSPL = SPL, kTile = 64 * W * SPL, kMaxSweeps = kTile + kMaxSweepsMargin;
    const int n_tiles = (sh.n_grid - 1 + kTile - 1) / kTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = threadIdx.x;
    const int nsteps = sh.n_grid - 1;

    Walker w;
    int status = walker_setup(sh, a, par, w);

    const double t0 = sh.tgrid[0];
    double M_s = par[2] * kMsol;
    double om_s = (2.0 * M_PI) / (1.0e-3 * par[1]);
    double cS0, cS1, cS2;
    {
        const Vd<3> tg{{t0, t0 * sh.inv_q, t0 * sh.inv_q * sh.inv_q}};
        const Vd<3> Sg = mdot_fb(w, tg);
        cS0 = Sg[0]; cS1 = Sg[1]; cS2 = Sg[2];
    }
    double cf0, cf1, cf2, cw1 = om_s, cw2 = om_s, cw3 = om_s, cw4 = om_s;
    double L_s;
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
        luminosity(sh, w, d_s, ov, Lt0, Lp0, Ld0);
        L_s = Lt0[0];
    }

    const int dsid = a.ds_id ? a.ds_id[walker] : 0;
    const DsDesc dsd = sh.ds ? sh.ds[(dsid >= 0 && dsid < sh.n_ds) ? dsid : 0] : DsDesc{0, 0, 0, 0};
    const int32_t *tptr = sh.tile_ptr + dsd.tile_off;
    constexpr int kRes = 64 * W;                     // observations resident in registers (one per lane of the workgroup)
    int ob_g = -1;
    double ob_dx = 0.0, ob_idt = 0.0, ob_y = 0.0, ob_ye = 1.0;
    if (a.want_chi2 && gl < dsd.n_obs) {
        const int jj = dsd.obs_off + gl;
        ob_g = sh.obs_g[jj]; ob_dx = sh.obs_dx[jj]; ob_idt = sh.obs_idt[jj]; ob_y = sh.obs_y[jj]; ob_ye = sh.obs_yerr[jj];
    }
    const int ob_tile = ob_g >= 0 ? ob_g / kTile : -1;
    const bool long_lc = a.want_chi2 && dsd.n_obs > kRes;
    double chi = 0.0;
    int sweeps_total = 0;
    int tp = 0;      // buffer parity of the scan totals
    int fp = 0;      // buffer parity of the sweep flags

    if (status == MP_STATUS_OK) {
        double tb_next[kSPL];
#pragma unroll
        for (int s = 0; s < kSPL; ++s) tb_next[s] = sh.tgrid[min(gl * kSPL + s + 1, nsteps)];
        double ta_next = sh.tgrid[min(gl * kSPL, nsteps)];

        for (int tile = 0; tile < n_tiles; ++tile) {
            const int i0 = tile * kTile + gl * kSPL;
            const int e0 = gl * kSPL;                    // index of this lane's first step inside the tile
            const int tq = tile & 1;                     // buffer parity of the per-tile LDS regions
            Vd<kSPL> tb, h;
#pragma unroll
            for (int s = 0; s < kSPL; ++s) {
                tb[s] = tb_next[s];
                tb_next[s] = sh.tgrid[min(i0 + kTile + s + 1, nsteps)];
            }
            const double ta0 = ta_next;
            ta_next = sh.tgrid[min(i0 + kTile, nsteps)];
#pragma unroll
            for (int s = 0; s < kSPL; ++s) h[s] = tb[s] - (s == 0 ? ta0 : tb[s - 1]);   // 0 for the padding steps

            // ---------------- Mdisc (2 barriers)
            Vd<kSPL> M1;
            double ES[kSPL + 3];
            {
                const Vd<kSPL> S1 = mdot_fb(w, tb);
#pragma unroll
                for (int s = 0; s < kSPL; ++s) { ES[3 + s] = S1[s]; lds.s[e0 + s + 3] = S1[s]; }
                if (gl == 0) { lds.s[2] = cS0; lds.s[1] = cS1; lds.s[0] = cS2; }
                __syncthreads();
                ES[2] = lds.s[e0 + 2]; ES[1] = lds.s[e0 + 1]; ES[0] = lds.s[e0];
                Vd<kSPL> zm, v0, v1, v2, v3;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    zm[s] = -h[s] * w.inv_tau;
                    v0[s] = ES[3 + s]; v1[s] = ES[2 + s]; v2[s] = ES[1 + s]; v3[s] = ES[s];
                }
                const Phi<kSPL> pm = phi1234(zm);
                const Vd<kSPL> inc = eam4_increment(sh, pm, h, v0, v1, v2, v3);
                double A = 1.0, B = 0.0;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) { B = fma(pm.e[s], B, inc[s]); A = A * pm.e[s]; }
                double M_wave;
                scan_affine_block<W>(A, B, lds.tot[tp], wave, lane, M_s, M_wave);
                tp ^= 1;
                const double Ax = lane_prev(A, 1.0), Bx = lane_prev(B, 0.0);
                double Mc = fma(Ax, M_wave, Bx);
#pragma unroll
                for (int s = 0; s < kSPL; ++s) { Mc = fma(pm.e[s], Mc, inc[s]); M1[s] = Mc; }
            }
            const DiscPt<kSPL> d1 = disc_point(sh, w, M1);

            // ---------------- omega: predictor
            Vd<kSPL> wg;
            {
                const double g1 = om_s - cw1, g2 = g1 - (cw1 - cw2);
                const double d2b = (cw1 - cw2) - (cw2 - cw3);
                const double g3 = tile == 0 ? 0.0 : g2 - d2b;
                const double g4 = tile == 0 ? 0.0 : g3 - (d2b - ((cw2 - cw3) - (cw3 - cw4)));
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    const double k = (double)(e0 + s + 1);
                    const double c2 = 0.5 * k * (k + 1.0);
                    const double c3 = c2 * (k + 2.0) * (1.0 / 3.0);
                    wg[s] = fma(k, g1, fma(c2, g2, fma(c3, g3, fma(c3 * (k + 3.0) * 0.25, g4, om_s))));
                }
            }
            // ---------------- Newton sweeps.  Each pass starts with the right-hand side at the current values and ONE
            // barrier that publishes (omega_dot, omega) for the neighbours together with every wavefront's verdict on the
            // previous pass; when nobody is pending those values are final (and omega_dot is exact at them).  A full pass
            // adds a second barrier for the scan totals.
            Vd<kSPL> f1;
            bool flagged = false, pending = true, not_ok = false, settled = false;
            int sweep = 0;
            double w_guard = om_s;
            while (true) {
                {
                    bool wild = false;
#pragma unroll
                    for (int s = 0; s < kSPL; ++s) wild = wild || !(wg[s] > 0.0);
                    if (__any(wild)) {
#pragma unroll
                        for (int s = 0; s < kSPL; ++s)
                            if (!(wg[s] > 0.0)) wg[s] = w_guard > 0.0 ? w_guard : om_s;
                    }
                }
                Vd<kSPL> rot, lam;
                f1 = omega_rhs<true>(sh, w, d1, wg, rot, lam);
                bool flg = false;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    lds.f[e0 + s + 3] = f1[s];
                    lds.w[e0 + s + 3] = wg[s];
                    flg = flg || (i0 + s < nsteps && rot[s] > 0.27);
                }
                if (gl == 0) {
                    double h1 = cf1, h2 = cf2, u1 = cw1, u2 = cw2;
                    if (tile == 0) {   // start-up ghosts: linear continuation of points 0 and 1 in the index
                        h1 = 2.0 * cf0 - f1[0]; u1 = 2.0 * om_s - wg[0];
                        h2 = 3.0 * cf0 - 2.0 * f1[0]; u2 = 3.0 * om_s - 2.0 * wg[0];
                    }
                    lds.f[2] = cf0; lds.f[1] = h1; lds.f[0] = h2;
                    lds.w[2] = om_s; lds.w[1] = u1; lds.w[0] = u2;
                }
                {
                    const int word = (__any(not_ok) ? 1 : 0) | (__any(settled && flg) ? 2 : 0);
                    if (lane == 0) lds.flags[fp][wave] = word;
                }
                __syncthreads();
                {
                    int all = 0;
#pragma unroll
                    for (int v = 0; v < W; ++v) all |= lds.flags[fp][v];
                    fp ^= 1;
                    pending = sweep == 0 || (all & 1);
                    flagged = flagged || (all & 2);
                }
                if (!pending || flagged || sweep >= kMaxSweeps) break;
                ++sweep;
                double Ef[kSPL + 3], Ew[kSPL + 3];
#pragma unroll
                for (int s = 0; s < kSPL; ++s) { Ef[3 + s] = f1[s]; Ew[3 + s] = wg[s]; }
                Ef[2] = lds.f[e0 + 2]; Ef[1] = lds.f[e0 + 1]; Ef[0] = lds.f[e0];
                Ew[2] = lds.w[e0 + 2]; Ew[1] = lds.w[e0 + 1]; Ew[0] = lds.w[e0];
                w_guard = Ew[2];
                Vd<kSPL> zw, n0, n1, n2, n3;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    zw[s] = h[s] * lam[s];
                    n0[s] = fma(-lam[s], Ew[3 + s], Ef[3 + s]);
                    n1[s] = fma(-lam[s], Ew[2 + s], Ef[2 + s]);
                    n2[s] = fma(-lam[s], Ew[1 + s], Ef[1 + s]);
                    n3[s] = fma(-lam[s], Ew[s], Ef[s]);
                }
                const Phi<kSPL> pw_ = phi1234(zw);
                const Vd<kSPL> inc = eam4_increment(sh, pw_, h, n0, n1, n2, n3);
                double A = 1.0, B = 0.0;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) { B = fma(pw_.e[s], B, inc[s]); A = A * pw_.e[s]; }
                double om_wave;
                scan_affine_block<W>(A, B, lds.tot[tp], wave, lane, om_s, om_wave);   // barrier: also orders the LDS image reads
                tp ^= 1;                                                              // before the next pass overwrites it
                const double Ax = lane_prev(A, 1.0), Bx = lane_prev(B, 0.0);
                double wc = fma(Ax, om_wave, Bx);
                bool all_ok = true, all_settled = true;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    wc = fma(pw_.e[s], wc, inc[s]);
                    const double dw = fabs(wc - wg[s]), mag = fabs(wc);
                    all_settled = all_settled && (dw <= 1.0e-3 * mag);
                    all_ok = all_ok && dw <= sh.sweep_tol * mag;
                    wg[s] = wc;
                }
                settled = all_settled;
                not_ok = !all_ok;
            }
            sweeps_total += sweep;

            // ---------------- luminosity; then ONE barrier publishes the light-curve tile, the carries for the next tile
            // and every wavefront's failure verdict
            Vd<kSPL> Lt, Lp, Ld;
            luminosity(sh, w, d1, wg, Lt, Lp, Ld);
            {
                bool bad = false, over = false;
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    const bool act = i0 + s < nsteps;
                    bad = bad || (act && (!(isfinite(M1[s]) && isfinite(wg[s])) || M1[s] <= 0.0 || wg[s] <= 0.0));
                    over = over || (act && sh.crot * wg[s] * wg[s] > 0.27);
                }
                const unsigned long long mb = __ballot(bad), mo = __ballot(over);
                if (lane == 0) {
                    lds.fail[tq][2 * wave] = mb ? wave * 64 + __ffsll(mb) - 1 : 0x7fffffff;
                    lds.fail[tq][2 * wave + 1] = mo ? wave * 64 + __ffsll(mo) - 1 : 0x7fffffff;
                }
#pragma unroll
                for (int s = 0; s < kSPL; ++s) {
                    lds.L[tq][e0 + s + 1] = Lt[s];
                    const int back = kTile - 1 - (e0 + s);          // 0 = last step end of the tile
                    if (back < 3) { lds.carry[tq][back] = ES[3 + s]; lds.carry[tq][3 + back] = f1[s]; }
                    if (back < 5) lds.carry[tq][6 + back] = wg[s];
                    if (back == 0) { lds.carry[tq][11] = M1[s]; lds.carry[tq][12] = Lt[s]; }
                }
                if (gl == 0) lds.L[tq][0] = L_s;
                __syncthreads();
                int first_bad = 0x7fffffff, first_flag = 0x7fffffff;
#pragma unroll
                for (int v = 0; v < W; ++v) { first_bad = min(first_bad, lds.fail[tq][2 * v]); first_flag = min(first_flag, lds.fail[tq][2 * v + 1]); }
                // a tile whose sweeps flagged an iterate or never settled: the reference's 'flag' (walker_eval)
                if (flagged || pending) first_flag = 0;
                if (first_bad != 0x7fffffff || first_flag != 0x7fffffff) {
                    status = first_flag <= first_bad ? MP_STATUS_FLAG : MP_STATUS_NONFINITE;
                    break;
                }
                const bool mine = ob_tile == tile;
                int j0 = 0, j1 = 0;
                if (long_lc) { j0 = max(tptr[tile * kSPL * W], kRes); j1 = tptr[min((tile + 1) * kSPL * W, sh.n_tiles)]; }
                if (mine) {
                    const int g = ob_g - tile * kTile;
                    const double La = lds.L[tq][g], Lb = lds.L[tq][g + 1];
                    const double mod = fma((Lb - La) * ob_idt, ob_dx, La) / 1.0e50;
                    const double res = (ob_y - mod) / ob_ye;
                    chi = fma(res, res, chi);
                }
                for (int j = j0 + gl; j < j1; j += kRes) {
                    const int jj = dsd.obs_off + j;
                    const int g = sh.obs_g[jj] - tile * kTile;
                    const double La = lds.L[tq][g], Lb = lds.L[tq][g + 1];
                    const double mod = fma((Lb - La) * sh.obs_idt[jj], sh.obs_dx[jj], La) / 1.0e50;
                    const double res = (sh.obs_y[jj] - mod) / sh.obs_yerr[jj];
                    chi = fma(res, res, chi);
                }
                cS0 = lds.carry[tq][0]; cS1 = lds.carry[tq][1]; cS2 = lds.carry[tq][2];
                cf0 = lds.carry[tq][3]; cf1 = lds.carry[tq][4]; cf2 = lds.carry[tq][5];
                om_s = lds.carry[tq][6]; cw1 = lds.carry[tq][7]; cw2 = lds.carry[tq][8]; cw3 = lds.carry[tq][9]; cw4 = lds.carry[tq][10];
                M_s = lds.carry[tq][11]; L_s = lds.carry[tq][12];
            }
        }
    }

    double lnp = -INFINITY;
    if (status == MP_STATUS_OK) {
        const double part = wave_sum(chi);
        __syncthreads();
        if (lane == 0) lds.tot[0][wave] = part;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int v = 0; v < W; ++v) tot += lds.tot[0][v];
        lnp = -0.5 * tot;
        if (!isfinite(lnp)) { lnp = -INFINITY; status = MP_STATUS_NONFINITE; }
    }
    lnp_out = lnp;
    status_out = status;
    sweeps_out = sweeps_total;
}

// ---------------------------------------------------------------- batched log-posterior kernel
template <bool CURVES, int SPL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, SPL >= 4 ? 1 : 2))) void lnprob_kernel(const DevShared sh, const LaunchArgs a) {
    __shared__ double Lbuf[64 * SPL + 1];
    const int walker = blockIdx.x;
    double par[MP_MAX_NDIM];
    const double *pw = a.pars + (size_t)walker * a.ndim;
#pragma unroll
    for (int i = 0; i < MP_MAX_NDIM; ++i) par[i] = i < a.ndim ? pw[i] : 0.0;
    double lnp;
    int status, sweeps;
    walker_eval<CURVES, SPL>(sh, a, walker, par, Lbuf, lnp, status, sweeps);
    if (threadIdx.x == 0) {
        a.lnprob[walker] = lnp;
        if (a.status) a.status[walker] = status;
        if (a.sweeps) a.sweeps[walker] = sweeps;
    }
}

template <int SPL, int W>
__global__ __launch_bounds__(64 * W) void lnprob_mw_kernel(const DevShared sh, const LaunchArgs a) {
    __shared__ MwLds<SPL, W> lds;
    const int walker = blockIdx.x;
    double par[MP_MAX_NDIM];
    const double *pw = a.pars + (size_t)walker * a.ndim;
#pragma unroll
    for (int i = 0; i < MP_MAX_NDIM; ++i) par[i] = i < a.ndim ? pw[i] : 0.0;
    double lnp;
    int status, sweeps;
    walker_eval_mw<SPL, W>(sh, a, walker, par, lds, lnp, status, sweeps);
    if (threadIdx.x == 0) {
        a.lnprob[walker] = lnp;
        if (a.status) a.status[walker] = status;
        if (a.sweeps) a.sweeps[walker] = sweeps;
    }
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
template <int SPL, int W>
__global__ __launch_bounds__(64 * W) void stretch_kernel(const DevShared sh, const StretchArgs g) {
    __shared__ typename std::conditional<(W > 1), MwLds<SPL, W>, double[64 * SPL + 1]>::type lds;
    const int w_ens = blockIdx.x / g.n_half;                       // which ensemble
    const int slot = blockIdx.x - w_ens * g.n_half;                // which walker of the active half
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
    double prop[MP_MAX_NDIM];
#pragma unroll
    for (int i = 0; i < MP_MAX_NDIM; ++i) prop[i] = par[i];
    LaunchArgs a{};
    a.ds_id = g.ds_id;
    a.ndim = g.ndim;
    a.physical = 0;
    a.want_chi2 = 1;
    double lnp;
    int status, sweeps;
    if (g.target == 1) {   // isotropic unit Gaussian: exercises the move itself (tests)
        lnp = 0.0;
#pragma unroll
        for (int i = 0; i < MP_MAX_NDIM; ++i) lnp = i < g.ndim ? sub_rn(lnp, mul_rn(mul_rn(0.5, prop[i]), prop[i])) : lnp;
        status = MP_STATUS_OK;
    } else {
        if constexpr (W > 1) walker_eval_mw<SPL, W>(sh, a, k, par, lds, lnp, status, sweeps);
        else walker_eval<false, SPL>(sh, a, k, par, lds, lnp, status, sweeps);
    }
    if (threadIdx.x == 0) {
        const double lnp_old = g.lnprob[k];
        const double lnpdiff = sub_rn(add_rn(mul_rn(g.ndim - 1.0, log(zz)), lnp), lnp_old);
        const bool accept = lnpdiff > log(u01(r2[0], r2[1]));      // false for NaN / -inf proposals
        if (accept) {
            for (int i = 0; i < g.ndim; ++i) g.pos[(size_t)k * g.ndim + i] = prop[i];
            g.lnprob[k] = lnp;
            g.n_accepted[k] += 1;
        }
        if (g.chain) {
            double *c = g.chain + ((size_t)g.chain_row * g.n_total + k) * g.ndim;
            for (int i = 0; i < g.ndim; ++i) c[i] = accept ? prop[i] : g.pos[(size_t)k * g.ndim + i];
            g.chain_lnp[(size_t)g.chain_row * g.n_total + k] = accept ? lnp : lnp_old;
        }
    }
}

int launch_lnprob(const DevShared &sh, const LaunchArgs &a, void *stream) {
    if (a.n <= 0) return 0;
    const bool curves = a.ltot || a.lprop || a.ldip || a.mdisc || a.omega;
    dim3 grid((unsigned)a.n), block(64);
    // Variants (results agree to rounding):
    //  - batches that leave SIMDs idle (256 CUs x 4 SIMDs) put 4 or 2 wavefronts on every walker;
    //  - up to 1536 walkers: one wavefront per walker, four steps per lane (256-step tiles amortise the
    //    wavefront scans best; needs the whole register file of a SIMD);
    //  - beyond: two steps per lane, which keeps two waves resident per SIMD.
    const bool wide = kernel_spl(a.n) == 4;
    hipStream_t st = (hipStream_t)stream;
    const int wpw = curves ? 1 : (sh.force_wpw ? sh.force_wpw : waves_per_walker(a.n));
    if (wpw == 4) {
        hipLaunchKernelGGL((lnprob_mw_kernel<1, 4>), grid, dim3(256), 0, st, sh, a);
    } else if (wpw == 2) {
        hipLaunchKernelGGL((lnprob_mw_kernel<2, 2>), grid, dim3(128), 0, st, sh, a);
    } else if (curves) {
        if (wide) hipLaunchKernelGGL((lnprob_kernel<true, 4>), grid, block, 0, st, sh, a);
        else hipLaunchKernelGGL((lnprob_kernel<true, 2>), grid, block, 0, st, sh, a);
    } else {
        if (wide) hipLaunchKernelGGL((lnprob_kernel<false, 4>), grid, block, 0, st, sh, a);
        else hipLaunchKernelGGL((lnprob_kernel<false, 2>), grid, block, 0, st, sh, a);
    }
    return (int)hipGetLastError();
}

int launch_stretch(const DevShared &sh, const StretchArgs &g, void *stream) {
    const int n_blocks = g.n_half * g.n_ensembles;
    if (n_blocks <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)n_blocks);
    const int wpw = g.target == 1 ? 1 : (sh.force_wpw ? sh.force_wpw : waves_per_walker(n_blocks));
    if (wpw == 4) hipLaunchKernelGGL((stretch_kernel<1, 4>), grid, dim3(256), 0, st, sh, g);
    else if (wpw == 2) hipLaunchKernelGGL((stretch_kernel<2, 2>), grid, dim3(128), 0, st, sh, g);
    else if (kernel_spl(n_blocks) == 4) hipLaunchKernelGGL((stretch_kernel<4, 1>), grid, dim3(64), 0, st, sh, g);
    else hipLaunchKernelGGL((stretch_kernel<2, 1>), grid, dim3(64), 0, st, sh, g);
    return (int)hipGetLastError();
}

}  // namespace mp
