// mp_math.hpp — device-side building blocks of the gfx950 kernels: DPP wavefront primitives (affine-map scan,
// neighbour fetch, broadcast), the per-lane N-vector type, hand-rolled fp64 elementary functions and the
// exponential-integrator phi functions.  Included by mp_kernels.hip only.
#pragma once
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

// exclusive prefix of a scanned affine map: (a, b) of lane-1, the identity (1, 0) in lane 0 (zero dwords by bound_ctrl)
MP_DEV void lane_prev_map(double a, double b, double &pa, double &pb) {
    const int alo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x138, 0xF, 0xF, true);
    const int ahi = __builtin_amdgcn_update_dpp(0x3FF00000, __double2hiint(a), 0x138, 0xF, 0xF, false);
    const int blo = __builtin_amdgcn_update_dpp(0, __double2loint(b), 0x138, 0xF, 0xF, true);
    const int bhi = __builtin_amdgcn_update_dpp(0, __double2hiint(b), 0x138, 0xF, 0xF, true);
    pa = __hiloint2double(ahi, alo);
    pb = __hiloint2double(bhi, blo);
}

// broadcast lane `src` (wave-uniform index) to all lanes
MP_DEV double lane_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// A value that is the same in every lane, moved to scalar registers (the compiler cannot prove uniformity of fp64 results,
// which only the vector ALU computes, and keeps them in vector registers otherwise)
MP_DEV double uniform(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Inclusive scan of affine maps x -> a*x + b over the 64 lanes: afterwards lane l holds
// m_l o m_{l-1} o ... o m_0.  Lanes without a DPP source combine with the identity (1, 0).
// Within-row shifts (all rows enabled) let the hardware supply the zero dwords of the identity (bound_ctrl: a lane
// whose source is outside the row reads 0): only the high dword of 1.0 has to be preset, one v_mov instead of four.
template <int CTRL, int ROW_MASK>
MP_DEV void scan_step(double &a, double &b) {
    double pa, pb;
    if constexpr (ROW_MASK == 0xF) {
        const int alo = __builtin_amdgcn_update_dpp(0, __double2loint(a), CTRL, 0xF, 0xF, true);
        const int ahi = __builtin_amdgcn_update_dpp(0x3FF00000, __double2hiint(a), CTRL, 0xF, 0xF, false);
        const int blo = __builtin_amdgcn_update_dpp(0, __double2loint(b), CTRL, 0xF, 0xF, true);
        const int bhi = __builtin_amdgcn_update_dpp(0, __double2hiint(b), CTRL, 0xF, 0xF, true);
        pa = __hiloint2double(ahi, alo);
        pb = __hiloint2double(bhi, blo);
    } else {
        pa = dpp_move<CTRL, ROW_MASK>(1.0, a);
        pb = dpp_move<CTRL, ROW_MASK>(0.0, b);
    }
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

// Raw hardware max / min (IEEE maxNum / minNum: a NaN operand is dropped).  hipcc's fmax/fmin first canonicalise every
// operand it cannot prove canonical (one v_max_f64 x, x, x each): in the per-lane extrema below that doubles the count.
// Operands are plain VALU results everywhere these are used.
MP_DEV double max_raw(double a, double b) {
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
MP_DEV double min_raw(double a, double b) {
    double d;
    asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
MP_DEV double maxabs_raw(double a, double b) {   // max(|a|, |b|)
    double d;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
MP_DEV double minabs_raw(double a, double b) {   // min(|a|, |b|)
    double d;
    asm("v_min_f64 %0, |%1|, |%2|" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// extrema over the N values of a lane
template <int N>
MP_DEV double lane_maxabs(const double (&v)[N]) {
    double m = N > 1 ? maxabs_raw(v[0], v[1]) : fabs(v[0]);
#pragma unroll
    for (int i = 2; i + 1 < N; i += 2) m = max_raw(m, maxabs_raw(v[i], v[i + 1]));
    if (N > 2 && (N & 1)) m = maxabs_raw(m, v[N - 1]);
    return m;
}
template <int N>
MP_DEV double lane_minabs(const double (&v)[N]) {
    double m = N > 1 ? minabs_raw(v[0], v[1]) : fabs(v[0]);
#pragma unroll
    for (int i = 2; i + 1 < N; i += 2) m = min_raw(m, minabs_raw(v[i], v[i + 1]));
    if (N > 2 && (N & 1)) m = minabs_raw(m, v[N - 1]);
    return m;
}
template <int N>
MP_DEV double lane_max(const double (&v)[N]) {
    double m = v[0];
#pragma unroll
    for (int i = 1; i < N; ++i) m = max_raw(m, v[i]);
    return m;
}
template <int N>
MP_DEV double lane_min(const double (&v)[N]) {
    double m = v[0];
#pragma unroll
    for (int i = 1; i < N; ++i) m = min_raw(m, v[i]);
    return m;
}

// p <- p*x + c on all N chains
template <int N>
MP_DEV void horner(Vd<N> &p, const Vd<N> &x, double c) {
    FORN p[i] = fma3(p[i], x[i], c);
}

// Polynomial coefficients from a small LDS table, fetched with broadcast ds_read_b128 (two coefficients per
// instruction).  fp64 VALU operands cannot be literals on gfx950, so every coefficient written as a C++ constant costs
// two v_mov_b32 per use site (14 % of the Newton-sweep loop).  Measured (tools/ab_run.sh): the table wins 2.6 % where
// two waves share a SIMD (the 2-steps-per-lane kernels: the other wave covers the LDS latency) and loses 1.5 % with one
// wave per SIMD (4 steps per lane, and the 4-wavefront kernel with 1 step per lane: +2.4 %), so it is used for N == 2 only.  Layout: [k] = 1/(16-k)! for k = 0..13 (1/16! .. 1/3!), then
// log2(e), -ln2_hi, -ln2_lo, pad, 1/17!, pad.  Filled by ktab_init() at kernel entry.
typedef double d2v __attribute__((ext_vector_type(2)));
template <int N>
constexpr bool kUseKtab = N == 2;
constexpr int kKtabN = 20;
__shared__ __attribute__((aligned(16))) double g_ktab[kKtabN];

__constant__ double kKtabInit[kKtabN] = {
    1.0 / 20922789888000.0, 1.0 / 1307674368000.0, 1.0 / 87178291200.0, 1.0 / 6227020800.0, 1.0 / 479001600.0,
    1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0,
    1.0 / 24.0, 1.0 / 6.0, 1.4426950408889634074, -6.93147180369123816490e-01, -1.90821492927058770002e-10, 0.0,
    1.0 / 355687428096000.0, 0.0};   // [18] = 1/17! (phi_5 series)

MP_DEV void ktab_init() {   // every thread of the workgroup calls this once, before any table read
    if (threadIdx.x < kKtabN) g_ktab[threadIdx.x] = kKtabInit[threadIdx.x];
    __syncthreads();
}

MP_DEV d2v ktab2(int k) {   // entries k, k+1 (k even); volatile: stays where it is written, inside the loops
    return *(volatile const __attribute__((address_space(3))) d2v *)&g_ktab[k];   // LDS pointer type: ds_read_b128, not flat
}

// The constants of the tile kinds (mp_device.h: kWtabStride doubles per kind), copied from the device table
// at kernel entry and read with broadcast ds_read_b128 where they are used.
__shared__ __attribute__((aligned(16))) double g_wtab[kWtabSize];

MP_DEV void wtab_init(const double *src) {   // every thread of the workgroup calls this once, before any table read
    for (int i = threadIdx.x; i < kWtabSize; i += blockDim.x) g_wtab[i] = src[i];
    __syncthreads();
}

MP_DEV d2v wtab2(int k) {   // entries k, k+1 (k even)
    return *(volatile const __attribute__((address_space(3))) d2v *)&g_wtab[k];
}
MP_DEV double wtab_theta(int kind, int i) { return g_wtab[kind * kWtabStride + kWtabTheta + i]; }

// p <- (p*x + c[k])*x + c[k+1] on all N chains
template <int N>
MP_DEV void horner2(Vd<N> &p, const Vd<N> &x, int k) {
    const d2v c = ktab2(k);
    FORN p[i] = fma3(p[i], x[i], c.x);
    FORN p[i] = fma3(p[i], x[i], c.y);
}

// Hand-rolled for this kernel's argument ranges (positive, normal, far from overflow): hardware
// seed (v_rcp_f64 / v_rsq_f64, ~2^-23) + ONE third-order correction step, without the scaling / fix-up code the
// general-purpose library versions carry.  All are accurate to ~1-2 ulp.
template <int N>
MP_DEV Vd<N> rcp_fast(const Vd<N> &x) {
    // one third-order step from the hardware seed: e = 1 - x r,  r <- r (1 + e + e^2)
    Vd<N> r, e;
    FORN r[i] = __builtin_amdgcn_rcp(x[i]);
    FORN e[i] = fma(-x[i], r[i], 1.0);
    FORN e[i] = fma(e[i], e[i], e[i]);
    FORN r[i] = fma(r[i], e[i], r[i]);
    return r;
}

template <int N>
MP_DEV Vd<N> rsqrt_fast(const Vd<N> &x) {
    // one third-order step from the hardware seed (relative error e0 ~ 2^-23 -> ~0.3 e0^3):
    // e = 1 - x y^2,  y <- y (1 + e/2 + 3 e^2/8)
    Vd<N> y, t, e, p;
    FORN y[i] = __builtin_amdgcn_rsq(x[i]);
    FORN t[i] = x[i] * y[i];
    FORN e[i] = fma(-t[i], y[i], 1.0);
    FORN p[i] = fma(e[i], 0.375, 0.5);
    FORN t[i] = y[i] * e[i];
    FORN y[i] = fma(t[i], p[i], y[i]);
    return y;
}

// e^x for x in [-750, 700]; underflows cleanly to 0 below
template <int N>
MP_DEV Vd<N> exp_fast(const Vd<N> &x) {
    Vd<N> k, r, p;
    if constexpr (kUseKtab<N>) {
        const d2v la = ktab2(14), lb = ktab2(16);    // log2(e), -ln2_hi | -ln2_lo, pad
        const d2v c4 = ktab2(4), c6 = ktab2(6), c8 = ktab2(8), c10 = ktab2(10), c12 = ktab2(12);   // all reads first
        FORN k[i] = __builtin_rint(x[i] * la.x);
        FORN r[i] = fma(k[i], la.y, x[i]);
        FORN r[i] = fma(k[i], lb.x, r[i]);
        FORN p[i] = fma3(r[i], c4.x, c4.y);          // Taylor degree 12 on |r| <= ln2/2 (1.7e-16): r/12! + 1/11!
        FORN p[i] = fma3(p[i], r[i], c6.x);          // 1/10!
        FORN p[i] = fma3(p[i], r[i], c6.y);          // 1/9!
        FORN p[i] = fma3(p[i], r[i], c8.x);          // 1/8!
        FORN p[i] = fma3(p[i], r[i], c8.y);          // 1/7!
        FORN p[i] = fma3(p[i], r[i], c10.x);         // 1/6!
        FORN p[i] = fma3(p[i], r[i], c10.y);         // 1/5!
        FORN p[i] = fma3(p[i], r[i], c12.x);         // 1/4!
        FORN p[i] = fma3(p[i], r[i], c12.y);         // 1/3!
    } else {
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
    }
    horner(p, r, 0.5);
    horner(p, r, 1.0);
    horner(p, r, 1.0);
    FORN p[i] = ldexp(p[i], (int)k[i]);
    return p;
}

// 10^x (un-logging of the sampler coordinates, code/synthetic_datasets/mcmc_eqns.py:16-17) as e^(x ln 10) with the
// product carried in two parts; ~2 ulp, a twentieth of the instructions of the general-purpose pow()
MP_DEV double exp10_fast(double x) {
    const double hi = x * 2.302585092994046;
    const double lo = fma(x, 2.302585092994046, -hi) + x * -2.1707562233822494e-16;
    const Vd<1> a{{fmin(fmax(hi, -750.0), 709.0)}};
    const double e = exp_fast(a)[0];
    return fma(e, lo, e);
}

// x^(-1/3) for positive normal x within float range: v_log_f32/v_exp_f32 seed (~1e-6) + one fourth-order step
template <int N>
MP_DEV Vd<N> rcbrt_fast(const Vd<N> &x) {
    // one fourth-order step: e = 1 - x y^3,  y <- y (1 - e)^(-1/3) = y (1 + e/3 + 2 e^2/9 + 14 e^3/81 + ...)
    // (the f32 log/exp seed is good to ~4e-6 for arguments up to 1e30, so e <~ 1.2e-5 and the e^4 term is < 1e-19)
    Vd<N> y, y3, e, p;
    FORN y[i] = (double)__builtin_amdgcn_exp2f(-0.33333333f * __builtin_amdgcn_logf((float)x[i]));
    FORN y3[i] = y[i] * y[i];
    FORN y3[i] = y3[i] * y[i];
    FORN e[i] = fma(-x[i], y3[i], 1.0);
    FORN p[i] = fma(e[i], 14.0 / 81.0, 2.0 / 9.0);
    FORN p[i] = fma(e[i], p[i], 1.0 / 3.0);
    FORN y3[i] = y[i] * e[i];
    FORN y[i] = fma(y3[i], p[i], y[i]);
    return y;
}

// x^(-1/7) for positive normal x within float range, same construction
template <int N>
MP_DEV Vd<N> pow_m1_7_fast(const Vd<N> &x) {
    // one fourth-order step: e = 1 - x y^7,  y <- y (1 - e)^(-1/7) = y (1 + e/7 + 4 e^2/49 + 20 e^3/343 + ...)
    // (seed good to ~4e-6, e <~ 3e-5: the e^4 term is < 1e-18)
    Vd<N> y, y2, y4, y7, e, p;
    FORN y[i] = (double)__builtin_amdgcn_exp2f(-0.14285715f * __builtin_amdgcn_logf((float)x[i]));
    FORN y2[i] = y[i] * y[i];
    FORN y4[i] = y2[i] * y2[i];
    FORN y7[i] = y4[i] * y2[i];
    FORN y7[i] = y7[i] * y[i];
    FORN e[i] = fma(-x[i], y7[i], 1.0);
    FORN p[i] = fma(e[i], 20.0 / 343.0, 4.0 / 49.0);
    FORN p[i] = fma(e[i], p[i], 1.0 / 7.0);
    FORN y7[i] = y[i] * e[i];
    FORN y[i] = fma(y7[i], p[i], y[i]);
    return y;
}

// ---------------------------------------------------------------- phi functions, order-5 quadrature
// phi_j(z) = sum_k z^k/(k+j)!  : phi_1 = (e^z-1)/z, phi_{j+1} = (phi_j - 1/j!)/z.  Both equations are advanced with the
// order-5 exponential Adams-Moulton formula: h * int_0^1 e^{z(1-theta)} P(theta) dtheta for the quartic P through the node
// values at t_{j+1}, t_j, ..., t_{j-3} (quadrature matrix W5 of the tile's kind, LDS table).
// phi_1..phi_5 of z = h*lambda.  Taylor series of phi_5: 7 terms when every |z| of the wavefront is below 1/32 (< 1e-17
// relative), 13 terms for |z| < 1/2, closed forms elsewhere.
template <int N>
struct Phi5 {
    Vd<N> e, p1, p2, p3, p4, p5;
};

#ifndef MP_PHI_ALL_BIG
#define MP_PHI_ALL_BIG 1
#endif
#ifndef MP_PHI_SKIP_SERIES
#define MP_PHI_SKIP_SERIES 1
#endif
template <int N>
MP_DEV Phi5<N> phi12345(const Vd<N> &z) {
    Vd<N> s;
    const double zmax = lane_maxabs(z.v);          // the lane's largest |z|: one comparison per range instead of one per step
    const bool all_tiny = zmax < 0.03125;
    double inv6 = 1.0 / 6.0, inv24 = 1.0 / 24.0;
#if MP_PHI_ALL_BIG
    // Every step of the wave in the stiff range (|z| >= 1/2: the tiles over 8 grid intervals from t ~ 20 s on): the recurrence
    // from e^z alone, without the series and the selects (the same arithmetic as the mixed path below, bit for bit).
    // (the 2-steps-per-lane kernels, two waves per SIMD, lose 4 % with this third path: absent there at compile time)
    if constexpr (N >= 4) if (__all(lane_minabs(z.v) >= 0.5)) {          // (false for a NaN among the z)
        if constexpr (kUseKtab<N>) { const d2v c = ktab2(12); inv24 = c.x; inv6 = c.y; }
        Phi5<N> r;
        Vd<N> zc;
        FORN zc[i] = fmax(z[i], -750.0);
        r.e = exp_fast(zc);
        const Vd<N> rz = rcp_fast(z);
        FORN {
            r.p1[i] = (r.e[i] - 1.0) * rz[i];
            r.p2[i] = (r.p1[i] - 1.0) * rz[i];
            r.p3[i] = (r.p2[i] - 0.5) * rz[i];
            r.p4[i] = (r.p3[i] - inv6) * rz[i];
            r.p5[i] = (r.p4[i] - inv24) * rz[i];
        }
        return r;
    }
#endif
    if constexpr (kUseKtab<N>) {
        if (__all(all_tiny)) {
            const d2v c4 = ktab2(4), c6 = ktab2(6), c8 = ktab2(8), c10 = ktab2(10), c12 = ktab2(12);
            FORN s[i] = fma3(z[i], c4.y, c6.x);       // z/11! + 1/10!
            FORN s[i] = fma3(s[i], z[i], c6.y);       // 1/9!
            FORN s[i] = fma3(s[i], z[i], c8.x);       // 1/8!
            FORN s[i] = fma3(s[i], z[i], c8.y);       // 1/7!
            FORN s[i] = fma3(s[i], z[i], c10.x);      // 1/6!
            FORN s[i] = fma3(s[i], z[i], c10.y);      // 1/5!
            inv24 = c12.x; inv6 = c12.y;
#if MP_PHI_SKIP_SERIES
        } else if (__all(lane_minabs(z.v) >= 0.5)) {
            // every step of the wave in the stiff range: the series' values would all be replaced below
            const d2v c12 = ktab2(12);
            inv24 = c12.x; inv6 = c12.y;
            FORN s[i] = 0.0;
#endif
        } else {
            {
                const d2v a = ktab2(18), b = ktab2(0);
                FORN s[i] = fma3(z[i], a.x, b.x);     // z/17! + 1/16!
                FORN s[i] = fma3(s[i], z[i], b.y);    // 1/15!
            }
            horner2(s, z, 2);                         // 1/14!, 1/13!
            horner2(s, z, 4);                         // 1/12!, 1/11!
            horner2(s, z, 6);                         // 1/10!, 1/9!
            horner2(s, z, 8);                         // 1/8!, 1/7!
            horner2(s, z, 10);                        // 1/6!, 1/5!
            const d2v c = ktab2(12);
            inv24 = c.x; inv6 = c.y;
        }
    } else if (__all(all_tiny)) {
        FORN s[i] = 1.0 / 39916800.0;             // 1/11!
        horner(s, z, 1.0 / 3628800.0);            // 1/10!
        horner(s, z, 1.0 / 362880.0);             // 1/9!
        horner(s, z, 1.0 / 40320.0);              // 1/8!
        horner(s, z, 1.0 / 5040.0);               // 1/7!
        horner(s, z, 1.0 / 720.0);                // 1/6!
        horner(s, z, 1.0 / 120.0);                // 1/5!
    } else {
        FORN s[i] = 1.0 / 355687428096000.0;      // 1/17!
        horner(s, z, 1.0 / 20922789888000.0);     // 1/16!
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
    }
    Phi5<N> r;
    r.p5 = s;
    FORN r.p4[i] = fma(z[i], s[i], inv24);
    FORN r.p3[i] = fma(z[i], r.p4[i], inv6);
    FORN r.p2[i] = fma(z[i], r.p3[i], 0.5);
    FORN r.p1[i] = fma(z[i], r.p2[i], 1.0);
    FORN r.e[i] = fma(z[i], r.p1[i], 1.0);
    const bool any_big = !(zmax < 0.5);           // (a NaN among the z is not seen by fmax: it stays a NaN in the series below)
    if (__any(any_big)) {                          // wave-uniform: only stiff / late-time tiles pay for this
        Vb<N> big;
        FORN big[i] = !(fabs(z[i]) < 0.5);
        Vd<N> zc, zs;
        FORN zc[i] = fmax(z[i], -750.0);
        FORN zs[i] = big[i] ? z[i] : 1.0;
        const Vd<N> ce = exp_fast(zc);
        const Vd<N> rz = rcp_fast(zs);
        FORN {
            const double c1 = (ce[i] - 1.0) * rz[i];
            const double c2 = (c1 - 1.0) * rz[i];
            const double c3 = (c2 - 0.5) * rz[i];
            const double c4 = (c3 - inv6) * rz[i];
            const double c5 = (c4 - inv24) * rz[i];
            r.e[i] = big[i] ? ce[i] : r.e[i];
            r.p1[i] = big[i] ? c1 : r.p1[i];
            r.p2[i] = big[i] ? c2 : r.p2[i];
            r.p3[i] = big[i] ? c3 : r.p3[i];
            r.p4[i] = big[i] ? c4 : r.p4[i];
            r.p5[i] = big[i] ? c5 : r.p5[i];
        }
    }
    return r;
}

// phi_6(z) next to a Phi5 of the same z (the Mdisc step, once per tile): Taylor series below |z| = 1/2 (12 terms, < 1e-16
// relative), the recurrence phi_6 = (phi_5 - 1/5!)/z elsewhere.
template <int N>
MP_DEV Vd<N> phi6(const Vd<N> &z, const Phi5<N> &p) {
    Vd<N> s;
    FORN s[i] = 1.0 / 355687428096000.0;      // 1/17!
    horner(s, z, 1.0 / 20922789888000.0);     // 1/16!
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
    Vd<N> zs;
    FORN zs[i] = fabs(z[i]) < 0.5 ? 1.0 : z[i];
    const Vd<N> rz = rcp_fast(zs);
    FORN s[i] = fabs(z[i]) < 0.5 ? s[i] : (p.p5[i] - 1.0 / 120.0) * rz[i];
    return s;
}

// The quadrature in node form: c_k = sum_m W[k][m] phi_{m+1}(z), increment = h * sum_k c_k v_k for the quartic through
// the node values v0..v4 at t_{j+1}, t_j, ..., t_{j-3}.  The Newton sweeps use this form because the weights depend on
// z = h*lambda only: a sweep that keeps lambda keeps them too.
template <int N>
struct EamW5 {
    Vd<N> c0, c1, c2, c3, c4;
};

// The same with the table rows read two ahead of their use (one-wavefront-per-SIMD kernels: nothing else covers an LDS round
// trip there).  eam5_rows_begin issues the reads of rows 0 and 1 -- call it BEFORE the phi functions are formed, which
// take longer than the round trip -- and eam5_node_weights_pipelined reads row k + 2 while it forms the weight of row k.
struct EamRows2 {
    d2v a0, b0, e0, a1, b1, e1;
};
MP_DEV EamRows2 eam5_rows_begin(int wbase) {
    EamRows2 r;
    r.a0 = wtab2(wbase); r.b0 = wtab2(wbase + 2); r.e0 = wtab2(wbase + 4);
    r.a1 = wtab2(wbase + 6); r.b1 = wtab2(wbase + 8); r.e1 = wtab2(wbase + 10);
    return r;
}
template <int N>
MP_DEV EamW5<N> eam5_node_weights_pipelined(int wbase, const Phi5<N> &p, const EamRows2 &first) {
    EamW5<N> w;
    d2v a[5], b[5], e[5];
    a[0] = first.a0; b[0] = first.b0; e[0] = first.e0;
    a[1] = first.a1; b[1] = first.b1; e[1] = first.e1;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (k + 2 < 5) { a[k + 2] = wtab2(wbase + 6 * (k + 2)); b[k + 2] = wtab2(wbase + 6 * (k + 2) + 2); e[k + 2] = wtab2(wbase + 6 * (k + 2) + 4); }
        Vd<N> &c = k == 0 ? w.c0 : k == 1 ? w.c1 : k == 2 ? w.c2 : k == 3 ? w.c3 : w.c4;
        FORN c[i] = e[k].x * p.p5[i];
        FORN c[i] = fma(b[k].y, p.p4[i], c[i]);
        FORN c[i] = fma(b[k].x, p.p3[i], c[i]);
        FORN c[i] = fma(a[k].y, p.p2[i], c[i]);
        FORN c[i] = fma(a[k].x, p.p1[i], c[i]);
    }
    return w;
}

template <int N>
MP_DEV EamW5<N> eam5_node_weights(int wbase, const Phi5<N> &p) {
    EamW5<N> w;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        Vd<N> &c = k == 0 ? w.c0 : k == 1 ? w.c1 : k == 2 ? w.c2 : k == 3 ? w.c3 : w.c4;
        const d2v a = wtab2(wbase + 6 * k), b = wtab2(wbase + 6 * k + 2), e = wtab2(wbase + 6 * k + 4);   // W5[k][0..4]
        FORN c[i] = e.x * p.p5[i];
        FORN c[i] = fma(b.y, p.p4[i], c[i]);
        FORN c[i] = fma(b.x, p.p3[i], c[i]);
        FORN c[i] = fma(a.y, p.p2[i], c[i]);
        FORN c[i] = fma(a.x, p.p1[i], c[i]);
    }
    return w;
}

template <int N>
MP_DEV Vd<N> eam5_increment_nodes(const EamW5<N> &w, const Vd<N> &h, const Vd<N> &v0, const Vd<N> &v1, const Vd<N> &v2,
                                  const Vd<N> &v3, const Vd<N> &v4) {
    Vd<N> acc;
    FORN acc[i] = w.c4[i] * v4[i];
    FORN acc[i] = fma(w.c3[i], v3[i], acc[i]);
    FORN acc[i] = fma(w.c2[i], v2[i], acc[i]);
    FORN acc[i] = fma(w.c1[i], v1[i], acc[i]);
    FORN acc[i] = fma(w.c0[i], v0[i], acc[i]);
    FORN acc[i] = h[i] * acc[i];
    return acc;
}

}  // namespace mp
