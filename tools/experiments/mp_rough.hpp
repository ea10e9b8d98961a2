// mp_rough.hpp — EXPERIMENT, NOT PART OF THE PRODUCT (measured in round 5 and not adopted: docs/EXPERIMENTS.md,
// profiles/r05_ab_rough_first_sweep.log): the FIRST Newton sweep of a tile in single precision.
// To repeat it: copy this file next to mp_eval.hpp, include it there behind the definition of DiscPt, and in the sweep loop of
// walker_eval, behind MP_PHASE(4), for W == 1 && kSPL == 4:
//     if (sweep == 1) { double ds, wc; over_sweeps += rough_sweep<kSPL>(sh, w, d1, h, wbase, om_s, cf0, cf1, cf2, cf3, cw1, cw2, cw3,
//                       startup, wg, ds, wc, Ew[3]) != 0ull; dsum_prev = (float)ds; continue; }
// What it showed: 419 instructions instead of ~1 000, bit-compatible decisions, every parity test green except the sweep
// budget -- and MORE sweeps (21 -> 25 per near-truth walker, 29 -> 32 prior-wide): a sweep contracts the error by 1e-2.5 ... 1e-4,
// so from an extrapolated guess that is 1e-2 ... 1e-3 off the double-precision sweep lands at 1e-5 ... 1e-7 and single
// precision at its floor over a 256-step scan, ~1e-6 x omega: the tiles that used to converge in two sweeps need a third.
//
// A sweep is a Newton-type step on the whole tile: what it returns is as good as the SQUARE of what it was given (plus the
// one-sweep lag of the history nodes).  The first sweep of a tile starts from an extrapolated guess that is 1e-1 ... 1e-3 off
// (tests/diagnostics/predictor_study.py: 5 % median over a tile of 8-interval steps), so its result is 1e-3 ... 1e-5 off
// whatever the arithmetic, and double precision buys it nothing.  In single precision the same sweep runs on packed
// instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two steps of the lane per instruction, i.e. twice the fp64 rate) and
// on the hardware's own 1 / x, 1 / sqrt(x) and 2^x (one quarter-rate instruction each where the fp64 code spends 6 - 25).
// The sweeps that follow are the double-precision ones and end on the same tests as before: the fixed point they converge
// to does not depend on how the iterate they start from was obtained, so the results move by what the sweep tolerance allows
// (<= 0.01 x sweep_tol per tile), like between any two kernel variants.  Nothing of the first sweep is kept but the new guess
// and the size of its correction; no decision is taken on it (a tile never ends on its first sweep).
// Serial restatement: none needed -- oracle/mp_oracle.c solves every step to convergence and knows no sweeps.
#pragma once
#include "mp_math.hpp"

namespace mp {

typedef float f2v __attribute__((ext_vector_type(2)));

template <int CTRL, int ROW_MASK>
MP_DEV float dpp_move32(float keep, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(src), CTRL, ROW_MASK, 0xF, false));
}
MP_DEV float lane_prev32(float v, float first) { return dpp_move32<0x138, 0xF>(first, v); }
MP_DEV float lane_bcast32(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

template <int CTRL, int ROW_MASK>
MP_DEV void scan_step32(float &a, float &b) {
    float pa, pb;
    if constexpr (ROW_MASK == 0xF) {   // (lanes without a source inside the row: the identity -- 1.0f preset, 0 by bound_ctrl)
        pa = __int_as_float(__builtin_amdgcn_update_dpp(0x3F800000, __float_as_int(a), CTRL, 0xF, 0xF, false));
        pb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), CTRL, 0xF, 0xF, true));
    } else {
        pa = dpp_move32<CTRL, ROW_MASK>(1.0f, a);
        pb = dpp_move32<CTRL, ROW_MASK>(0.0f, b);
    }
    b = fmaf(a, pb, b);
    a = a * pa;
}
MP_DEV void scan_affine32(float &a, float &b) {
    scan_step32<0x111, 0xF>(a, b);
    scan_step32<0x112, 0xF>(a, b);
    scan_step32<0x114, 0xF>(a, b);
    scan_step32<0x118, 0xF>(a, b);
    scan_step32<0x142, 0xA>(a, b);
    scan_step32<0x143, 0xC>(a, b);
}

MP_DEV f2v f2(float x) { return f2v{x, x}; }
MP_DEV f2v rsq2(f2v x) { return f2v{__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)}; }
MP_DEV f2v rcp2(f2v x) { return f2v{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
MP_DEV f2v exp2_2(f2v x) { return f2v{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
MP_DEV f2v abs2(f2v x) { return f2v{fabsf(x.x), fabsf(x.y)}; }
MP_DEV f2v fma2(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }
// per-component select
MP_DEV f2v sel2(bool cx, bool cy, f2v a, f2v b) { return f2v{cx ? a.x : b.x, cy ? a.y : b.y}; }

// One sweep over the N (even) steps of this lane in single precision: wg (the guess of omega at the lane's step ends) is
// replaced by the sweep's result; dsum = sum over the lane's steps of |new - old|, wc = the lane's last new value,
// ew3 = the previous lane's last OLD value (what the wild-guess repair of the next sweep falls back to), return value: the
// lanes in which an old value was beyond the break-up limit.  Same formulas as omega_rhs<true> / phi12345 / eam5_* / the
// affine scan of walker_eval (mp_eval.hpp), statement by statement; Nacc's prefactor sqrt(GM)/I x Mdisc/tvisc is formed in
// double precision before it is rounded (its factors are 1e-32 and 1e28).
template <int N>
MP_DEV unsigned long long rough_sweep(const DevShared &sh, const Walker &w, const DiscPt<N> &d1, const Vd<N> &h, int wbase, double om_s,
                                      double cf0, double cf1, double cf2, double cf3, double cw1, double cw2, double cw3, bool startup,
                                      Vd<N> &wg, double &dsum_out, double &wc_out, double &ew3) {
    static_assert(N % 2 == 0, "pairs of steps");
    constexpr int P = N / 2;
#define FORP _Pragma("unroll") for (int p = 0; p < P; ++p)
    f2v W[P], H[P], MA[P], RMU[P], SQU[P], QU[P];
    FORP {
        W[p] = f2v{(float)wg[2 * p], (float)wg[2 * p + 1]};
        H[p] = f2v{(float)h[2 * p], (float)h[2 * p + 1]};
        MA[p] = f2v{(float)(w.armI * d1.mdot[2 * p]), (float)(w.armI * d1.mdot[2 * p + 1])};
        RMU[p] = f2v{(float)d1.rmu[2 * p], (float)d1.rmu[2 * p + 1]};
        SQU[p] = f2v{(float)d1.squ[2 * p], (float)d1.squ[2 * p + 1]};
        QU[p] = f2v{(float)d1.qu[2 * p], (float)d1.qu[2 * p + 1]};
    }
    const float kc = (float)w.kc, sqrt_kc = (float)w.sqrt_kc, Kc = (float)w.Kc, DI = (float)w.DI, n = (float)sh.cfg.n_ode;
    const float crot = (float)sh.crot, sqrtR = (float)sh.sqrtR;
    // ---- omega_dot and its derivative (omega_rhs<true>)
    f2v F[P], LAM[P];
    float rot_max = 0.0f;
    FORP {
        const f2v Y = rsq2(W[p]), IO = Y * Y, RLC = f2(kc) * IO;
        const bool cx = RMU[p].x >= RLC.x, cy = RMU[p].y >= RLC.y;
        const f2v SQ = sel2(cx, cy, f2(sqrt_kc) * Y, SQU[p]);
        const f2v FAST = sel2(cx, cy, f2(Kc) * Y, W[p] * QU[p]);
        const f2v X = fma2(f2(n), FAST, f2(-n));
        const f2v E = exp2_2(f2(-2.885390082f) * abs2(X));               // e^(-2 |x|)
        const f2v R = rcp2(f2(1.0f) + E);
        const f2v T = (f2(1.0f) - E) * R;
        const f2v TH = f2v{copysignf(T.x, X.x), copysignf(T.y, X.y)};    // tanh(x)
        const f2v OM2 = W[p] * W[p], ROT = f2(crot) * OM2;
        rot_max = fmaxf(rot_max, fmaxf(ROT.x, ROT.y));
        const f2v ARM = f2v{fmaxf(SQ.x, sqrtR), fmaxf(SQ.y, sqrtR)};
        const f2v A = sel2(ROT.x > 0.27f, ROT.y > 0.27f, f2(0.0f), MA[p] * ARM);   // (sqrt(GM max(Rm, R)) / I) Mdisc / tvisc, 0 beyond break-up
        F[p] = fma2(f2(-DI) * OM2, W[p], -(A * TH));
        const f2v DFAST = sel2(cx, cy, f2(-0.5f), f2(1.0f)) * FAST * IO;
        const f2v DTH = f2(4.0f * n) * E * R * R * DFAST;                // n sech^2(x) dfast
        const f2v CD = sel2(cx && SQ.x >= sqrtR, cy && SQ.y >= sqrtR, f2(-0.5f), f2(0.0f));
        LAM[p] = fma2(f2(-3.0f * DI), OM2, -(A * fma2(CD * IO, TH, DTH)));
    }
    const unsigned long long over_now = __ballot(rot_max > 0.27f);
    // ---- the four points before every step: the own earlier steps, the previous lane's last four, the history in front of the tile
    float Ef[N + 4], Ew[N + 4];
    FORP { Ef[4 + 2 * p] = F[p].x; Ef[5 + 2 * p] = F[p].y; Ew[4 + 2 * p] = W[p].x; Ew[5 + 2 * p] = W[p].y; }
    float h1 = (float)cf1, h2 = (float)cf2, h3 = (float)cf3, u1 = (float)cw1, u2 = (float)cw2, u3 = (float)cw3;
    const float f0 = (float)cf0, w0 = (float)om_s;
    if (startup) {   // the three points before the grid continue points 0 and 1 linearly in the index
        const float fp1 = lane_bcast32(Ef[4], 0), wp1 = lane_bcast32(Ew[4], 0);
        h1 = 2.0f * f0 - fp1; u1 = 2.0f * w0 - wp1;
        h2 = 3.0f * f0 - 2.0f * fp1; u2 = 3.0f * w0 - 2.0f * wp1;
        h3 = 4.0f * f0 - 3.0f * fp1; u3 = 4.0f * w0 - 3.0f * wp1;
    }
    Ef[3] = lane_prev32(Ef[N + 3], f0);  Ew[3] = lane_prev32(Ew[N + 3], w0);
    Ef[2] = lane_prev32(Ef[N + 2], h1);  Ew[2] = lane_prev32(Ew[N + 2], u1);
    Ef[1] = lane_prev32(Ef[N + 1], h2);  Ew[1] = lane_prev32(Ew[N + 1], u2);
    Ef[0] = lane_prev32(Ef[N + 0], h3);  Ew[0] = lane_prev32(Ew[N + 0], u3);
    ew3 = (double)Ew[3];
    // ---- phi functions of z = h lambda: upward from the series of phi_5 below |z| = 1, downward from e^z above
    // the quadrature weights of the kind (wave-uniform: LDS table, rounded here)
    float Wq[5][5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const d2v a = wtab2(wbase + 6 * k), b = wtab2(wbase + 6 * k + 2), e = wtab2(wbase + 6 * k + 4);
        Wq[k][0] = (float)a.x; Wq[k][1] = (float)a.y; Wq[k][2] = (float)b.x; Wq[k][3] = (float)b.y; Wq[k][4] = (float)e.x;
    }
    f2v AZ[P], INC[P];
    FORP {
        const f2v Zr = H[p] * LAM[p];
        const f2v Z = f2v{fminf(fmaxf(Zr.x, -80.0f), 80.0f), fminf(fmaxf(Zr.y, -80.0f), 80.0f)};
        f2v s = fma2(Z, f2(1.0f / 3628800.0f), f2(1.0f / 362880.0f));
        s = fma2(s, Z, f2(1.0f / 40320.0f));
        s = fma2(s, Z, f2(1.0f / 5040.0f));
        s = fma2(s, Z, f2(1.0f / 720.0f));
        s = fma2(s, Z, f2(1.0f / 120.0f));
        const f2v q4 = fma2(Z, s, f2(1.0f / 24.0f)), q3 = fma2(Z, q4, f2(1.0f / 6.0f)), q2 = fma2(Z, q3, f2(0.5f));
        const f2v q1 = fma2(Z, q2, f2(1.0f)), qe = fma2(Z, q1, f2(1.0f));
        const bool bx = !(fabsf(Z.x) < 1.0f), by = !(fabsf(Z.y) < 1.0f);
        const f2v ce = exp2_2(f2(1.442695041f) * Z);
        const f2v rz = rcp2(sel2(bx, by, Z, f2(1.0f)));
        const f2v c1 = (ce - f2(1.0f)) * rz, c2 = (c1 - f2(1.0f)) * rz, c3 = (c2 - f2(0.5f)) * rz;
        const f2v c4 = (c3 - f2(1.0f / 6.0f)) * rz, c5 = (c4 - f2(1.0f / 24.0f)) * rz;
        const f2v p1 = sel2(bx, by, c1, q1), p2 = sel2(bx, by, c2, q2), p3 = sel2(bx, by, c3, q3), p4 = sel2(bx, by, c4, q4);
        const f2v p5 = sel2(bx, by, c5, s);
        AZ[p] = sel2(bx, by, ce, qe);
        // increment = h sum_k c_k (f - lambda omega)_k, c_k = sum_m W[k][m] phi_{m+1}
        f2v acc = f2(0.0f);
#pragma unroll
        for (int k = 4; k >= 0; --k) {
            f2v c = f2(Wq[k][4]) * p5;
            c = fma2(f2(Wq[k][3]), p4, c);
            c = fma2(f2(Wq[k][2]), p3, c);
            c = fma2(f2(Wq[k][1]), p2, c);
            c = fma2(f2(Wq[k][0]), p1, c);
            const f2v ef = f2v{Ef[4 + 2 * p - k], Ef[5 + 2 * p - k]}, ew = f2v{Ew[4 + 2 * p - k], Ew[5 + 2 * p - k]};
            acc = fma2(c, fma2(-LAM[p], ew, ef), acc);
        }
        INC[p] = H[p] * acc;
    }
    // ---- the lane's steps composed, the wavefront's maps scanned, the tile's start value propagated
    float A = 1.0f, B = 0.0f;
    FORP {
        B = fmaf(AZ[p].x, B, INC[p].x); A = A * AZ[p].x;
        B = fmaf(AZ[p].y, B, INC[p].y); A = A * AZ[p].y;
    }
    scan_affine32(A, B);
    const float Ax = __int_as_float(__builtin_amdgcn_update_dpp(0x3F800000, __float_as_int(A), 0x138, 0xF, 0xF, false));
    const float Bx = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(B), 0x138, 0xF, 0xF, true));
    float wc = fmaf(Ax, w0, Bx), dsum = 0.0f;
    FORP {
        wc = fmaf(AZ[p].x, wc, INC[p].x); dsum += fabsf(wc - W[p].x); wg[2 * p] = (double)wc;
        wc = fmaf(AZ[p].y, wc, INC[p].y); dsum += fabsf(wc - W[p].y); wg[2 * p + 1] = (double)wc;
    }
#undef FORP
    dsum_out = (double)dsum;
    wc_out = (double)wc;
    return over_now;
}

}  // namespace mp
