/*
 * oracle/mp_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, serial, fp64 CPU restatement of the magprop hot path (lnprior -> ODE
 * integration -> luminosity light curve -> linear interpolation -> -0.5*chi^2).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this; the product (magprop_amd/) never does and fails loudly without its HIP
 * library.
 *
 * What is restated, and from where (paths relative to the reference checkout):
 *   walker constants, RHS       code/synthetic_datasets/funcs.py:75-142, magnetar/funcs.py:33-101
 *   initial conditions          code/synthetic_datasets/funcs.py:51-71,  magnetar/funcs.py:17-29
 *   luminosity stage            code/synthetic_datasets/funcs.py:175-229, magnetar/funcs.py:157-210
 *   interpolation + 1e50 scale  code/synthetic_datasets/funcs.py:233-236, magnetar/funcs.py:213-217
 *   chi^2                       code/synthetic_datasets/mcmc_eqns.py:25,  magnetar/mcmc_eqns.py:37
 *   box prior                   code/synthetic_datasets/mcmc_eqns.py:28-49, magnetar/mcmc_eqns.py:40-84
 *   un-logging of pars[2:]      code/synthetic_datasets/mcmc_eqns.py:16-17
 *   6/7/8/9-parameter dispatch  magnetar/mcmc_eqns.py:22-34
 * The RHS and the luminosity stage keep the reference's formulas literally (pow, cbrt-as-pow,
 * tanh from libm), so any algebraic shortcut taken by the HIP kernel is checked against the
 * un-simplified physics.
 *
 * Third-party arithmetic on the path: the reference integrates with scipy.integrate.odeint
 * (ODEPACK LSODA, scipy==1.3.0 pinned in requirements.txt:9; default rtol=atol~1.49e-8) — an
 * adaptive multistep method that cannot be reproduced bit-for-bit and is itself only ~1e-7
 * accurate.  The deterministic scheme used here and by the HIP kernel:
 *   - Mdisc obeys dMdisc/dt = Mdotfb(t) - Mdisc/tvisc exactly (eta1+eta2 == 1,
 *     code/synthetic_datasets/funcs.py:122-129): linear and omega-independent.  It is advanced
 *     with an exponential (integrating-factor) step whose source term is the quadratic interpolant
 *     of Mdotfb through t_i, t_i+h/2, t_i+h — unconditionally stable for h >> tvisc.
 *   [cross-check scheme, mpo_trajectory_etd4rk]
 *   - omega is advanced with the exponential fourth-order Runge-Kutta scheme of Krogstad
 *     (ETD4RK, J. Comput. Phys. 203 (2005) 72) around the per-step frozen Jacobian
 *     lambda_i = d(omega_dot)/d(omega) at (t_i, omega_i), one step per grid interval, fed Mdisc at
 *     t_i, t_i+h/2, t_i+h.  For h*lambda -> 0 it IS classical RK4; it stays accurate where strong
 *     accretion torques pin the spin to its equilibrium and make the omega equation stiff
 *     (|h*lambda| up to ~20 inside the prior box, where plain RK4 on the output grid is 24 % off).
 *   - failure ('flag', code/synthetic_datasets/funcs.py:172-173) is the deterministic rule
 *     "rotation parameter exceeds 0.27 at a grid point or RK stage", which is where LSODA
 *     chatters on the Nacc discontinuity and gives up.
 * Pinning: tests/test_oracle.py checks this file against golden vectors produced by importing the
 * real reference (tests/golden/make_golden.py) and against the reference's own test fixtures
 * (the CSVs under tests/test_data, decimated copies under tests/golden/).
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC; loaded through ctypes by oracle/c_oracle.py)
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* same field order as mp_model_cfg in include/magprop_amd.h (kept separate on purpose) */
typedef struct mpo_cfg {
    double inertia_factor, rm_massflow_factor, n_ode, n_lum, alpha, cs7, k;
    double dipeff, propeff, f_beam, nacc_lum_threshold;
    int32_t lprop_gm_term;
    int32_t dipole_torque;   /* 0: Ndip = -mu^2 omega^3 / (6 c^3) (magnetar/funcs.py:78; code/figure_3.py:79, "Piro & Ott");
                                1: Ndip = -(2/3) (mu^2 omega^3 / c^3) (Rlc / Rm)^3, Rm after the cap (code/figure_3.py:140-141, "Bucciantini") */
} mpo_cfg;

enum { MPO_OK = 0, MPO_FLAG = 1, MPO_NONFINITE = 2, MPO_PRIOR = 3 };

/* magnetar/funcs.py:7-13 */
static const double G_ = 6.674e-8, C_ = 3.0e10, R_ = 1.0e6, MSOL_ = 1.99e33;

typedef struct {
    double M, GM, I, modW;            /* star */
    double tvisc, mu, M0, tfb;        /* per walker, funcs.py:57-61 */
    double dipeff, propeff, f_beam;   /* per walker (7/8/9-parameter likelihoods) */
} wk;

static void walker_setup(const mpo_cfg *c, const double *p, int ndim, wk *w) {
    double B = p[0], MdiscI = p[2], RdiscI = p[3], epsilon = p[4], delta = p[5];
    w->M = 1.4 * MSOL_;
    w->GM = G_ * w->M;
    w->I = c->inertia_factor * w->M * pow(R_, 2.0);
    /* binding energy, magnetar/funcs.py:75-76 */
    w->modW = 0.6 * w->M * pow(C_, 2.0) *
              ((w->GM / (R_ * pow(C_, 2.0))) / (1.0 - 0.5 * (w->GM / (R_ * pow(C_, 2.0)))));
    double Rdisc = RdiscI * 1.0e5;
    w->tvisc = Rdisc / (c->alpha * c->cs7 * 1.0e7);
    w->mu = 1.0e15 * B * pow(R_, 3.0);
    w->M0 = delta * MdiscI * MSOL_;
    w->tfb = epsilon * w->tvisc;
    w->dipeff = c->dipeff;
    w->propeff = c->propeff;
    w->f_beam = c->f_beam;
    /* magnetar/mcmc_eqns.py:22-34 */
    if (ndim == 7) w->f_beam = p[6];
    if (ndim == 8) { w->dipeff = p[6]; w->propeff = p[7]; }
    if (ndim == 9) { w->dipeff = p[6]; w->propeff = p[7]; w->f_beam = p[8]; }
}

/* fallback rate, code/synthetic_datasets/funcs.py:128 */
static double mdot_fb(const wk *w, double t) {
    return (w->M0 / w->tfb) * pow((t + w->tfb) / w->tfb, -5.0 / 3.0);
}

/* shared by the RHS and the luminosity stage: radii, fastness, efficiencies */
typedef struct { double Rm, eta2, Mdotprop, Mdotacc, rot_param; } flow;

static void flow_state(const mpo_cfg *c, const wk *w, double n, double Mdisc, double omega, flow *f) {
    double Rm = pow(w->mu, 4.0 / 7.0) * pow(w->GM, -1.0 / 7.0) *
                pow((c->rm_massflow_factor * Mdisc) / w->tvisc, -2.0 / 7.0);
    double Rc = pow(w->GM / pow(omega, 2.0), 1.0 / 3.0);
    double Rlc = C_ / omega;
    if (Rm >= c->k * Rlc) Rm = c->k * Rlc;
    double fast = pow(Rm / Rc, 3.0 / 2.0);
    double bigT = 0.5 * w->I * pow(omega, 2.0);
    f->rot_param = bigT / w->modW;
    f->eta2 = 0.5 * (1.0 + tanh(n * (fast - 1.0)));
    double eta1 = 1.0 - f->eta2;
    f->Mdotprop = f->eta2 * (Mdisc / w->tvisc);
    f->Mdotacc = eta1 * (Mdisc / w->tvisc);
    f->Rm = Rm;
}

/*
 * d(omega)/dt, code/synthetic_datasets/funcs.py:119,131-140.  *rot receives the rotation parameter,
 * *dfdw (if not NULL) the analytic Jacobian d(omega_dot)/d(omega), obtained by the chain rule on the
 * same literal formulas (Mdisc held fixed).
 */
static double omega_dot(const mpo_cfg *c, const wk *w, double Mdisc, double omega, double *rot, double *dfdw) {
    flow f;
    flow_state(c, w, c->n_ode, Mdisc, omega, &f);
    double Ndip = (-1.0 * pow(w->mu, 2.0) * pow(omega, 3.0)) / (6.0 * pow(C_, 3.0));
    if (c->dipole_torque == 1) {   /* code/figure_3.py:140-141 */
        double Rlc = C_ / omega;
        Ndip = (-2.0 / 3.0) * ((pow(w->mu, 2.0) * pow(omega, 3.0)) / pow(C_, 3.0)) * pow(Rlc / f.Rm, 3.0);
    }
    double Nacc;
    int branch;
    if (f.rot_param > 0.27) {
        Nacc = 0.0; branch = 0;
    } else if (f.Rm >= R_) {
        Nacc = pow(w->GM * f.Rm, 0.5) * (f.Mdotacc - f.Mdotprop); branch = 1;
    } else {
        Nacc = pow(w->GM * R_, 0.5) * (f.Mdotacc - f.Mdotprop); branch = 2;
    }
    *rot = f.rot_param;
    if (dfdw) {
        double Rm_u = pow(w->mu, 4.0 / 7.0) * pow(w->GM, -1.0 / 7.0) *
                      pow((c->rm_massflow_factor * Mdisc) / w->tvisc, -2.0 / 7.0);
        int capped = Rm_u >= c->k * (C_ / omega);
        double dRm = capped ? -c->k * C_ / (omega * omega) : 0.0;
        double Rc = pow(w->GM / pow(omega, 2.0), 1.0 / 3.0);
        double dRc = -(2.0 / 3.0) * Rc / omega;
        double fast = pow(f.Rm / Rc, 1.5);
        double dfast = 1.5 * fast * (dRm / f.Rm - dRc / Rc);
        double ch = cosh(c->n_ode * (fast - 1.0));
        double deta2 = 0.5 * c->n_ode * dfast / (ch * ch);      /* sech^2; cosh overflow -> 0 */
        double ddiff = -2.0 * deta2 * (Mdisc / w->tvisc);       /* d(Mdotacc - Mdotprop) */
        double dNacc = 0.0;
        if (branch == 1)
            dNacc = 0.5 * pow(w->GM / f.Rm, 0.5) * dRm * (f.Mdotacc - f.Mdotprop) + pow(w->GM * f.Rm, 0.5) * ddiff;
        else if (branch == 2)
            dNacc = pow(w->GM * R_, 0.5) * ddiff;
        double dNdip = (-3.0 * pow(w->mu, 2.0) * pow(omega, 2.0)) / (6.0 * pow(C_, 3.0));
        if (c->dipole_torque == 1)   /* Ndip = -(2/3) mu^2 / Rm^3: depends on omega through the capped radius only */
            dNdip = 2.0 * pow(w->mu, 2.0) * dRm / pow(f.Rm, 4.0);
        *dfdw = (dNacc + dNdip) / w->I;
    }
    return (Nacc + Ndip) / w->I;
}

/*
 * The right-hand side itself at one state: out = (dMdisc/dt, domega/dt), magnetar/funcs.py:33-101,
 * code/synthetic_datasets/funcs.py:75-142 (PHYSICAL parameters).  *lam (optional) = d(omega_dot)/d(omega).
 */
void mpo_rhs(const mpo_cfg *c, const double *pars, int ndim, double t, double Mdisc, double omega, double *out,
             double *lam) {
    wk w;
    walker_setup(c, pars, ndim, &w);
    flow f;
    flow_state(c, &w, c->n_ode, Mdisc, omega, &f);
    out[0] = mdot_fb(&w, t) - f.Mdotprop - f.Mdotacc;          /* funcs.py:129 */
    double rot;
    out[1] = omega_dot(c, &w, Mdisc, omega, &rot, lam);
}

/* phi_1..3(z) = sum_k z^k/(k+j)!  (phi_1 = (e^z-1)/z ...): Taylor below |z| = 0.5, closed forms above */
static void phi123(double z, double *ez, double *p1, double *p2, double *p3) {
    if (fabs(z) < 0.5) {
        double s = 0.0, term = 1.0 / 6.0;
        for (int k = 0; k < 20; ++k) { s += term; term *= z / (double)(k + 4); }
        *p3 = s; *p2 = z * s + 0.5; *p1 = z * *p2 + 1.0; *ez = z * *p1 + 1.0;
    } else {
        *ez = exp(z);
        *p1 = expm1(z) / z;
        *p2 = (*p1 - 1.0) / z;
        *p3 = (*p2 - 0.5) / z;
    }
}

/*
 * Weights of the exponential Mdisc step: I_k(Z) = int_0^1 exp(-Z(1-u)) u^k du, k = 0,1,2
 * = phi_1(-Z), phi_2(-Z), 2*phi_3(-Z).
 */
static void etd_weights(double Z, double *I0, double *I1, double *I2) {
    double e, p1, p2, p3;
    phi123(-Z, &e, &p1, &p2, &p3);
    *I0 = p1; *I1 = p2; *I2 = 2.0 * p3;
}

/*
 * Integrate over tgrid[0..n).  nsub >= 1 RK4/exponential sub-steps per grid interval (1 is the
 * production scheme; larger values are for convergence studies in the tests).
 * Mout/Wout (may be NULL) receive the state at the grid points.  Returns a status.
 */
int mpo_trajectory_etd4rk(const mpo_cfg *c, const double *pars, int ndim, const double *tgrid, int n,
                          int nsub, double *Mout, double *Wout) {
    wk w;
    walker_setup(c, pars, ndim, &w);
    /* code/synthetic_datasets/funcs.py:66-69 */
    double M = pars[2] * MSOL_;
    double om = (2.0 * M_PI) / (1.0e-3 * pars[1]);
    double rot, rmax = 0.0;
    int status = MPO_OK;
    if (nsub < 1) nsub = 1;
    for (int i = 0; i < n; ++i) {
        if (Mout) Mout[i] = M;
        if (Wout) Wout[i] = om;
        if (!(isfinite(M) && isfinite(om)) || M <= 0.0 || om <= 0.0) { status = MPO_NONFINITE; }
        else {
            rot = (0.5 * w.I * om * om) / w.modW;
            if (rot > 0.27) status = MPO_FLAG;
        }
        if (status != MPO_OK || i == n - 1) {
            if (status != MPO_OK) {
                for (int j = i + 1; j < n; ++j) { if (Mout) Mout[j] = NAN; if (Wout) Wout[j] = NAN; }
            }
            break;
        }
        double H = tgrid[i + 1] - tgrid[i];
        double h = H / nsub;
        for (int s = 0; s < nsub; ++s) {
            double t0 = tgrid[i] + s * h;
            /* Mdisc: exponential step with quadratic source interpolant */
            double S0 = mdot_fb(&w, t0), S1 = mdot_fb(&w, t0 + 0.5 * h), S2 = mdot_fb(&w, t0 + h);
            double c1 = -3.0 * S0 + 4.0 * S1 - S2, c2 = 2.0 * S0 - 4.0 * S1 + 2.0 * S2;
            double z = h / w.tvisc, I0, I1, I2, J0, J1, J2;
            etd_weights(z, &I0, &I1, &I2);
            etd_weights(0.5 * z, &J0, &J1, &J2);
            double Mh = exp(-0.5 * z) * M + 0.5 * h * (S0 * J0 + 0.5 * c1 * J1 + 0.25 * c2 * J2);
            double M1 = exp(-z) * M + h * (S0 * I0 + c1 * I1 + c2 * I2);
            /* omega: Krogstad ETD4RK around lambda = d(omega_dot)/d(omega) at the step start */
            double lam, r0, r2, r3, r4;
            double f0 = omega_dot(c, &w, M, om, &r0, &lam);
            double zz = h * lam, e1, p1, p2, p3, eh, q1, q2, q3;
            phi123(zz, &e1, &p1, &p2, &p3);
            phi123(0.5 * zz, &eh, &q1, &q2, &q3);
            double N0 = f0 - lam * om;
            double U2 = eh * om + 0.5 * h * q1 * N0;
            double N2 = omega_dot(c, &w, Mh, U2, &r2, NULL) - lam * U2;
            double U3 = eh * om + 0.5 * h * (q1 - 2.0 * q2) * N0 + h * q2 * N2;
            double N3 = omega_dot(c, &w, Mh, U3, &r3, NULL) - lam * U3;
            double U4 = e1 * om + h * (p1 - 2.0 * p2) * N0 + 2.0 * h * p2 * N3;
            double N4 = omega_dot(c, &w, M1, U4, &r4, NULL) - lam * U4;
            om = e1 * om + h * ((p1 - 3.0 * p2 + 4.0 * p3) * N0 + (2.0 * p2 - 4.0 * p3) * (N2 + N3) +
                                (4.0 * p3 - p2) * N4);
            if (r0 > rmax) rmax = r0;
            if (r2 > rmax) rmax = r2;
            if (r3 > rmax) rmax = r3;
            if (r4 > rmax) rmax = r4;
            M = M1;
            if (rmax > 0.27) { status = MPO_FLAG; }
        }
        if (status != MPO_OK) {
            for (int j = i + 1; j < n; ++j) { if (Mout) Mout[j] = NAN; if (Wout) Wout[j] = NAN; }
            break;
        }
    }
    return status;
}

/*
 * Quadrature matrix of the exponential Adams-Moulton step on a geometric grid of ratio q.
 * Nodes (in units of the step h_j, origin t_j): t_{j+1}, t_j, t_{j-1}, t_{j-2}  ->  1, 0, -1/q, -(1/q + 1/q^2).
 * With N(theta) = sum_k N_k l_k(theta) the cubic through the node values,
 *   int_0^1 exp(z (1 - theta)) N(theta) dtheta = sum_m phi_{m+1}(z) sum_k Wm[k][m] N_k ,  Wm[k][m] = m! [theta^m] l_k.
 */
void mpo_eam4_weights(double q, double Wm[4][4]) {
    const double x[4] = {1.0, 0.0, -1.0 / q, -(1.0 / q + 1.0 / (q * q))};
    const double fact[4] = {1.0, 1.0, 2.0, 6.0};
    for (int k = 0; k < 4; ++k) {
        double c[4] = {1.0, 0.0, 0.0, 0.0}; /* polynomial coefficients, ascending */
        int deg = 0;
        double denom = 1.0;
        for (int j = 0; j < 4; ++j) {
            if (j == k) continue;
            /* c(theta) *= (theta - x_j) */
            for (int m = deg + 1; m >= 1; --m) c[m] = c[m - 1] - x[j] * c[m];
            c[0] = -x[j] * c[0];
            ++deg;
            denom *= x[k] - x[j];
        }
        for (int m = 0; m < 4; ++m) Wm[k][m] = fact[m] * c[m] / denom;
    }
}

/* phi_1..phi_K(z): Taylor below |z| = 0.5, closed forms above */
static void phi_upto(double z, double *ez, double *ph, int K) {
    if (fabs(z) < 0.5) {
        double s = 0.0, term = 1.0;
        for (int i = 2; i <= K; ++i) term /= (double)i;                       /* 1/K! */
        for (int k = 0; k < 30; ++k) { s += term; term *= z / (double)(k + K + 1); }
        ph[K - 1] = s;
        for (int j = K - 2; j >= 0; --j) {
            double fact = 1.0;
            for (int i = 2; i <= j + 1; ++i) fact *= (double)i;               /* (j+1)! */
            ph[j] = z * ph[j + 1] + 1.0 / fact;
        }
        *ez = z * ph[0] + 1.0;
    } else {
        *ez = exp(z);
        ph[0] = expm1(z) / z;
        double fact = 1.0;
        for (int j = 1; j < K; ++j) { ph[j] = (ph[j - 1] - 1.0 / fact) / z; fact *= (double)(j + 1); }
    }
}

/*
 * Quadrature matrix of the order-K exponential Adams-Moulton step on a geometric grid of ratio q: nodes (in units of the
 * step h_j, origin t_j) t_{j+1}, t_j, t_{j-1}, ... -> 1, 0, -1/q, -(1/q + 1/q^2), ...;  W[k*K + m] = m! [theta^m] l_k(theta).
 */
void mpo_eam_weights(double q, int K, double *W) {
    double x[8];
    x[0] = 1.0; x[1] = 0.0;
    { double acc = 0.0, f = 1.0; for (int k = 2; k < K; ++k) { f /= q; acc -= f; x[k] = acc; } }
    for (int k = 0; k < K; ++k) {
        double c[8] = {1.0, 0, 0, 0, 0, 0, 0, 0};
        int deg = 0;
        double denom = 1.0;
        for (int j = 0; j < K; ++j) {
            if (j == k) continue;
            for (int m = deg + 1; m >= 1; --m) c[m] = c[m - 1] - x[j] * c[m];
            c[0] = -x[j] * c[0];
            ++deg;
            denom *= x[k] - x[j];
        }
        double fact = 1.0;
        for (int m = 0; m < K; ++m) { if (m > 1) fact *= (double)m; W[k * K + m] = fact * c[m] / denom; }
    }
}

/* cubic Hermite on [0, 1] (theta), step h: value from the end values y0, y1 and end derivatives d0, d1 */
static double hermite(double th, double h, double y0, double d0, double y1, double d1) {
    const double D = y1 - y0;
    return y0 + th * (h * d0 + th * ((3.0 * D - h * (2.0 * d0 + d1)) + th * (h * (d0 + d1) - 2.0 * D)));
}
static double hermite_d(double th, double h, double y0, double d0, double y1, double d1) {   /* d/dt of the same cubic */
    const double D = y1 - y0;
    return (h * d0 + th * (2.0 * (3.0 * D - h * (2.0 * d0 + d1)) + th * 3.0 * (h * (d0 + d1) - 2.0 * D))) / h;
}

/* cubic Lagrange interpolation on four nodes (values only) */
static double lagrange4(const double x[4], const double y[4], double t) {
    double p = 0.0;
    for (int k = 0; k < 4; ++k) {
        double l = 1.0;
        for (int m = 0; m < 4; ++m) if (m != k) l *= (t - x[m]) / (x[k] - x[m]);
        p += l * y[k];
    }
    return p;
}

/* quintic Hermite: values, first and second derivatives at both ends (Mdisc: its derivatives are analytic) */
static double hermite5(double th, double h, double y0, double d0, double e0, double y1, double d1, double e1) {
    const double t2 = th * th, t3 = t2 * th, t4 = t3 * th, t5 = t4 * th;
    return (1.0 - 10.0 * t3 + 15.0 * t4 - 6.0 * t5) * y0 + (10.0 * t3 - 15.0 * t4 + 6.0 * t5) * y1 +
           h * ((th - 6.0 * t3 + 8.0 * t4 - 3.0 * t5) * d0 + (-4.0 * t3 + 7.0 * t4 - 3.0 * t5) * d1) +
           h * h * ((0.5 * t2 - 1.5 * t3 + 1.5 * t4 - 0.5 * t5) * e0 + (0.5 * t3 - t4 + 0.5 * t5) * e1);
}
static double mdot_fb_dt(const wk *w, double t) {   /* d/dt of the fallback rate */
    return (-5.0 / 3.0) * mdot_fb(w, t) / (t + w->tfb);
}

/* which smooth branch of the right-hand side a state is on: bit 0 = Alfven radius capped at k*Rlc
 * (code/synthetic_datasets/funcs.py:109-110), bit 1 = Rm >= R (the torque-arm branch, :133-138) */
static int branch_flags(const mpo_cfg *c, const wk *w, double Mdisc, double omega) {
    double Rm_u = pow(w->mu, 4.0 / 7.0) * pow(w->GM, -1.0 / 7.0) * pow((c->rm_massflow_factor * Mdisc) / w->tvisc, -2.0 / 7.0);
    int capped = Rm_u >= c->k * (C_ / omega);
    double Rm = capped ? c->k * C_ / omega : Rm_u;
    return capped | ((Rm >= R_) ? 2 : 0);
}

/*
 * PRODUCTION SCHEME (what the HIP kernels implement): exponential Adams-Moulton of order 5 for both
 * equations (the source of the linear Mdisc equation is analytic) on geometric grids:
 *   Mdisc_{j+1} = e^{-z} Mdisc_j + h Mdotfb(t_j) sum_k c_k (-x)^k phi_{k+1}(-z),   z = h/tvisc, x = h/(t_j + tfb),
 *                 c_k = Gamma(k + 5/3)/Gamma(5/3), k = 0..5  (the power-law source expanded about the step start)
 *   omega_{j+1} = e^{h lam} omega_j + h sum_m phi_{m+1}(h lam) sum_k W5[k][m] (f_{j+1-k} - lam omega_{j+1-k})
 * with f_p = omega_dot(Mdisc_p, omega_p) and lam = d(omega_dot)/d(omega) frozen at the NEW point (implicit in omega_{j+1},
 * solved by fixed-point iteration).  The integration proceeds in TILES of 64*spl steps (one wavefront, spl steps per lane
 * on the GPU); the step of a tile is a fixed multiple of the output grid's:
 *   - the first 32 grid intervals are covered with 1/8-interval sub-steps (the spin-up transient of heavy discs around
 *     strongly magnetised stars is faster than the output grid at t = 1 s);
 *   - mode 0 (FIXED): every later tile steps over single grid intervals;
 *   - mode 1 (ADAPTIVE, the product default): tiles step over 1, 2, 4 or 8 grid intervals.  A tile at stride > 1 is kept only up
 *     to the first lane in which (a) the solution changes the smooth branch of the right-hand side (Alfven-radius cap, torque
 *     arm: a kink no multistep formula can cross at a coarse step) or (b) the smoothness indicator
 *     120 |phi_5(h lam)| h |4th difference of (f - lam omega)| / omega (the formula's own error term) exceeds stride_tol
 *     (a tenth of it for tiles over 8 intervals and for the tile behind the sub-steps); what follows is redone finer.
 *     The stride that is tried adapts per walker (opt_s, trouble).  Values at skipped grid points: omega from the
 *     step's cubic Hermite interpolant in (omega, f); Mdisc from the quintic in (Mdisc, dMdisc/dt, d2Mdisc/dt2) while the
 *     step resolves tvisc, from the cubic Lagrange interpolant of Mdisc / (tvisc Mdotfb) through four nodes beyond.
 * History for a tile whose step differs from its predecessor's: the Hermite interpolant on the predecessor's steps
 * (exact where the points coincide).  Start-up: the missing (f, omega) points continue points 0 and 1 linearly.  Failure ('flag'): the rotation parameter of an iterate at a step end exceeds
 * 0.27 in a tile at stride <= 1; at a coarser stride the tile is redone finer first.
 * (The kernels additionally stop the sweeps of a coarse tile early when its first lanes show that nothing will be kept, and
 * cut a tile whose sweeps are slow where they have converged; both lead to the same kind of redo as here.)
 */
typedef struct { double t, M, dM, ddM, w, f; } mpo_node;
typedef struct {
    int tiles, tiles_pre, tiles_s1, tiles_s2, tiles_s4, tiles_cut;   /* tile solves by kind (s4: stride >= 4); tiles not (fully) kept */
    int steps_kept;
} mpo_stats;

#define MPO_PRE_FINE 32   /* grid intervals covered by the sub-stepped tiles */
#define MPO_PRE_SUB 8     /* sub-steps per grid interval there */
#define MPO_PRE_EARLY_END_FACTOR 6553600.0   /* mp_device.h MP_PRE_EARLY_END_FACTOR: 65 536 x 100, the calm-first-tile test holds the scaled indicator to a HUNDREDTH of the bound */
#define MPO_EARLY_HOLD_SECONDS 4.0   /* coarse tiles starting before this time are held to a tenth of stride_tol */
#define MPO_MIN_KEEP 8    /* a coarse tile is kept if at least this many lanes precede the first offending one */
#define MPO_CUT_WINDOW 24 /* lanes from a cut on over which the excess of the indicator decides the next stride (mp_eval.hpp MP_CUT_BY_RATIO) */

/* omega, f, Mdisc, dMdisc/dt at time t <= the last accepted node (Hermite on the accepted steps; exact at nodes) */
static int node_lookup(const mpo_node *nd, int nn, double t, double *wv, double *fv) {
    int hi = nn - 1;
    while (hi > 0 && nd[hi - 1].t > t * (1.0 + 1e-13)) --hi;
    if (fabs(t / nd[hi].t - 1.0) < 1e-13) { *wv = nd[hi].w; *fv = nd[hi].f; return 1; }
    if (hi == 0) return 0;                                    /* before the first node */
    const mpo_node *a = &nd[hi - 1], *b = &nd[hi];
    if (fabs(t / a->t - 1.0) < 1e-13) { *wv = a->w; *fv = a->f; return 1; }
    const double h = b->t - a->t, th = (t - a->t) / h;
    *wv = hermite(th, h, a->w, a->f, b->w, b->f);
    *fv = hermite_d(th, h, a->w, a->f, b->w, b->f);
    return 1;
}

int mpo_trajectory_mode(const mpo_cfg *c, const double *pars, int ndim, const double *tgrid, int n, int mode, int spl,
                        double stride_tol, int max_stride, double *Mout, double *Wout, mpo_stats *stats) {
    enum { P = 5 };
    wk w;
    walker_setup(c, pars, ndim, &w);
    const int TILE = 64 * spl, nsteps = n - 1;
    const double q = exp(log(tgrid[n - 1] / tgrid[0]) / (double)nsteps);
    if (stride_tol <= 0.0) stride_tol = 1.0e-7;
    if (max_stride <= 0) max_stride = 8;
    mpo_stats st_ = {0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < n; ++j) { if (Mout) Mout[j] = NAN; if (Wout) Wout[j] = NAN; }
    mpo_node *nd = (mpo_node *)malloc(sizeof(mpo_node) * (size_t)(n + MPO_PRE_FINE * MPO_PRE_SUB + TILE + 8));
    mpo_node *tn = (mpo_node *)malloc(sizeof(mpo_node) * (size_t)(TILE + 1));
    double *ind = (double *)malloc(sizeof(double) * (size_t)TILE);
    int *brk = (int *)malloc(sizeof(int) * (size_t)TILE);
    int nn = 0, status = MPO_OK;
    double M = pars[2] * MSOL_;                        /* code/synthetic_datasets/funcs.py:66-69 */
    double om = (2.0 * M_PI) / (1.0e-3 * pars[1]);
    double rot, fcur = omega_dot(c, &w, M, om, &rot, NULL);
    if (!(isfinite(M) && isfinite(om)) || M <= 0.0 || om <= 0.0) status = MPO_NONFINITE;
    else if (rot > 0.27) status = MPO_FLAG;
    {
        const double dM0 = mdot_fb(&w, tgrid[0]) - M / w.tvisc;
        nd[nn++] = (mpo_node){tgrid[0], M, dM0, mdot_fb_dt(&w, tgrid[0]) - dM0 / w.tvisc, om, fcur};
    }
    if (Mout) Mout[0] = M;
    if (Wout) Wout[0] = om;
    int pre_fine = nsteps < MPO_PRE_FINE ? nsteps : MPO_PRE_FINE;
    int i0 = 0;          /* grid index of the tile start */
    int sub_done = 0;    /* sub-steps of the pre-phase done */
    int s = 1;           /* stride of the next grid tile */
    int last_was_pre = 0;
    int cool = 0;        /* tiles over single intervals for which the scaled indicator decides about coarsening */
    int trouble = 0;     /* coarse tiles that kept nothing: after two of them stride 8 is no longer tried, only reached when the
                            indicator of a full tile over 4 intervals promotes it */
    int opt_s = 0;       /* the stride that is tried after a calm tile over single intervals: lowered when such an attempt fails
                            outright, raised when the indicator promotes a tile (0: max_stride) */
    while (status == MPO_OK && i0 < nsteps) {
        const int pre = i0 < pre_fine;
        const int after_pre = !pre && last_was_pre;
        int nc;
        double Q;
        if (pre) {
            const int left = pre_fine * MPO_PRE_SUB - sub_done;
            nc = left < TILE ? left : TILE;
            Q = pow(q, 1.0 / MPO_PRE_SUB);
            s = 1;
        } else {
            if (mode == 0) s = 1;
            /* (whole steps only; and at least three of them, for the dense output's four nodes) */
            while (s > 1 && ((nsteps - i0) % s != 0 || (nsteps - i0) / s < 3)) s /= 2;
            nc = ((nsteps - i0) / s < TILE) ? (nsteps - i0) / s : TILE;
            Q = pow(q, (double)s);
        }
        double W5[25];
        mpo_eam_weights(Q, P, W5);
        const double t0 = nd[nn - 1].t;
        double fh[P], wh[P];
        fh[0] = fcur; wh[0] = om;
        int startup = (nn == 1);
        for (int k = 1; k <= P - 2 && !startup; ++k)
            if (!node_lookup(nd, nn, t0 / pow(Q, (double)k), &wh[k], &fh[k])) startup = 1;
        const double f0 = fh[0], w0 = om;
        double f1 = 0.0, w1 = 0.0, Mt = M, ot = om;
        const int flags0 = branch_flags(c, &w, M, om);
        int tstatus = MPO_OK, J;
        for (J = 0; J < nc; ++J) {
            double tJ1;
            if (pre) { const int k = sub_done + J + 1; tJ1 = (k % MPO_PRE_SUB == 0) ? tgrid[k / MPO_PRE_SUB] : tgrid[0] * pow(Q, (double)k); }
            else tJ1 = tgrid[i0 + (J + 1) * s];
            const double tJ = (J == 0) ? t0 : tn[J - 1].t, h = tJ1 - tJ;
            /* ---- Mdisc */
            const double S1 = mdot_fb(&w, tJ1);
            double ez, ph[8], acc = 0.0;
            phi_upto(-h / w.tvisc, &ez, ph, 6);
            /* the source is a power law of t + tfb: inside the step Mdotfb(tJ + theta h) = Mdotfb(tJ) (1 + theta x)^(-5/3),
               x = h / (tJ + tfb) <= 1 - 1/Q; its binomial series integrates term by term against the exponential kernel,
               int_0^1 e^{-z (1 - theta)} theta^k dtheta = k! phi_{k+1}(-z).  Terms up to x^5: where the step is much longer
               than tvisc (phi_{k+1}(-z) -> 1/(k! z)) the series is the binomial series of (1 + x)^(-5/3) itself and the
               first neglected term is 4 x^6 = 7e-12 at a stride of 8 grid intervals (1e-13 at 4). */
            {
                const double x = h / (tJ + w.tfb);
                acc = mdot_fb(&w, tJ) * (ph[0] + x * (-(5.0 / 3.0) * ph[1] + x * ((40.0 / 9.0) * ph[2] +
                                         x * (-(440.0 / 27.0) * ph[3] + x * ((6160.0 / 81.0) * ph[4] - x * (104720.0 / 243.0) * ph[5])))));
            }
            const double M1 = ez * Mt + h * acc;
            /* ---- omega: fixed-point iteration on the implicit step */
            double wn = ot + h * fh[0], fnew = 0.0, lam = 0.0, Nv[P];
            int flagged = 0;
            for (int it = 0; it < 200; ++it) {
                double r1;
                fnew = omega_dot(c, &w, M1, wn, &r1, &lam);
                if (r1 > 0.27) { flagged = 1; break; }
                Nv[0] = fnew - lam * wn;
                Nv[1] = fh[0] - lam * wh[0];
                for (int k = 1; k <= P - 2; ++k) {
                    double hf, hw;
                    if (startup && J < k) {   /* point j-k lies before the grid: points 0 and 1 continued linearly (point -m) */
                        const double m_ = (double)(k - J), fp1 = (J == 0) ? fnew : f1, wp1 = (J == 0) ? wn : w1;
                        hf = (1.0 + m_) * f0 - m_ * fp1; hw = (1.0 + m_) * w0 - m_ * wp1;
                    } else { hf = fh[k]; hw = wh[k]; }
                    Nv[k + 1] = hf - lam * hw;
                }
                phi_upto(h * lam, &ez, ph, P);
                acc = 0.0;
                for (int m = 0; m < P; ++m) { double g = 0.0; for (int k = 0; k < P; ++k) g += W5[k * P + m] * Nv[k]; acc += ph[m] * g; }
                const double wnew = ez * ot + h * acc, d = fabs(wnew - wn);
                wn = wnew;
                if (!(d > 1e-15 * fabs(wn))) break; /* converged, or NaN */
            }
            if (flagged) { tstatus = MPO_FLAG; break; }
            fnew = omega_dot(c, &w, M1, wn, &rot, NULL);
            if (!(isfinite(M1) && isfinite(wn)) || M1 <= 0.0 || wn <= 0.0) { tstatus = MPO_NONFINITE; break; }
            if ((0.5 * w.I * wn * wn) / w.modW > 0.27) { tstatus = MPO_FLAG; break; }
            brk[J] = branch_flags(c, &w, M1, wn) != flags0;
            /* the formula's error term is h phi_5(h lam) x the 4th difference of the node values: phi_5(0) = 1/120, and
               -> 1/(24 |h lam|) where the equation is stiff and the exponential integrator tracks the spin equilibrium */
            ind[J] = (startup && J < P - 2) ? 0.0
                                            : 120.0 * fabs(ph[4]) * h * fabs(Nv[0] - 4.0 * Nv[1] + 6.0 * Nv[2] - 4.0 * Nv[3] + Nv[4]) / fabs(wn);
            tn[J] = (mpo_node){tJ1, M1, S1 - M1 / w.tvisc, mdot_fb_dt(&w, tJ1) - (S1 - M1 / w.tvisc) / w.tvisc, wn, fnew};
            if (startup && J == 0) { f1 = fnew; w1 = wn; }
            for (int k = P - 2; k >= 1; --k) { fh[k] = fh[k - 1]; wh[k] = wh[k - 1]; }
            fh[0] = fnew; wh[0] = wn;
            Mt = M1; ot = wn;
        }
        const int nsolved = J;                       /* steps before a failure */
        ++st_.tiles;
        if (pre) ++st_.tiles_pre; else if (s == 1) ++st_.tiles_s1; else if (s == 2) ++st_.tiles_s2; else ++st_.tiles_s4;
        /* ---- what is kept (lane granularity: lane l owns steps l*spl .. l*spl + spl - 1) */
        int keep = nc, next_s = s;
        if (pre || s == 1) {
            if (tstatus != MPO_OK) { status = tstatus; break; }            /* a verdict only at stride <= 1 */
        } else {
            /* the tile tried at a coarse stride right behind the sub-steps has no calm predecessor to vouch for it: a tenth;
               steps over 8 intervals: likewise (with the plain bound the prior-wide golden points come out up to 1.0e-7 off
               the reference's tight values instead of 0.5e-7; with 0.3 one soak walker in 32 768 is 1.01e-7 off the fixed steps) */
            /* coarse tiles that start before t = MPO_EARLY_HOLD_SECONDS (where the spin-up transients live) too:
               of 262 144 soak walkers one, a model far above its data whose Lprop cancels to 1e-2, came out 1.45e-7 off the
               fixed steps through stride-4 tiles at t < 16 s that pass the plain bound; held to the tenth: 7.3e-8, no cost.
               (Round 3 expressed this as grid index 1 024 = 4.1 s on the reference's "L" grid; on its "S" grid that index is
               at 8 ms.  A time protects both.) */
            const double tile_tol = (after_pre || s >= 8 || t0 < MPO_EARLY_HOLD_SECONDS) ? 0.1 * stride_tol : stride_tol;
            int first = (nsolved < nc) ? nsolved / spl : 64;
            for (int l = 0; l < first && l * spl < nc; ++l)
                for (int e = l * spl; e < (l + 1) * spl && e < nsolved; ++e)
                    if (brk[e] || ind[e] > tile_tol) { if (l < first) first = l; }
            if (first * spl >= nc) first = 64;
            if (first < 64) {
                ++st_.tiles_cut;
                if (first < 2 * MPO_MIN_KEEP) cool = 3;                      /* a coarse attempt that failed early */
                if (first < MPO_MIN_KEEP) {
                    /* nothing worth keeping: redone at the next finer stride at once (over single intervals after a kink) */
                    int kink = 0;
                    for (int e = 0; e < MPO_MIN_KEEP * spl && e < nsolved; ++e) kink |= brk[e];
                    opt_s = s / 2 > 2 ? s / 2 : 2;
                    ++trouble;
                    s = (s > 2 && !kink) ? s / 2 : 1;
                    continue;
                }
                keep = first * spl;
                next_s = 1;                                                  /* the offending region gets single intervals ... */
                /* ... unless it is a fast feature of the solution, not a kink (round 4, mp_eval.hpp MP_CUT_BY_RATIO): then the
                   stride its excess over the bound asks for -- order 5: a halving of the step buys 32 x, margin 2 as in the
                   promotions below -- judged over the MPO_CUT_WINDOW lanes from the cut on; a kink among them: single intervals */
                {
                    int kink = 0;
                    double excess = 0.0;
                    for (int e = first * spl; e < (first + MPO_CUT_WINDOW) * spl && e < nsolved; ++e) {
                        kink |= brk[e];
                        if (ind[e] / tile_tol > excess) excess = ind[e] / tile_tol;
                    }
                    int cut_by_ind = 0;
                    for (int e = first * spl; e < (first + 1) * spl && e < nsolved; ++e) cut_by_ind |= (!brk[e] && ind[e] > tile_tol);
                    for (int e = first * spl; e < (first + 1) * spl && e < nsolved; ++e) if (brk[e]) cut_by_ind = 0;
                    if (cut_by_ind && !kink && (first + 1) * spl <= nsolved)
                        next_s = excess <= 16.0 ? s / 2 : (excess <= 512.0 ? (s / 4 > 1 ? s / 4 : 1) : 1);
                }
            }
        }
        /* ---- stride of the next tile (mode 1): coarsen when the kept steps were calm (5th-order scaling of the indicator) */
        if (mode == 1 && !pre && keep == nc && next_s == s) {
            double imax = 0.0, ipost = 0.0;
            int lastbrk = -1;
            for (int e = 0; e < keep; ++e) { if (ind[e] > imax) imax = ind[e]; if (brk[e] && lastbrk < 0) lastbrk = e / spl; }
            if (s == 1) {
                /* no kink in this tile: the scaled indicator decides while a recent coarse attempt has failed (cool > 0);
                   otherwise the coarse stride is simply tried (the tile is cut where it does not hold): the indicator of
                   a tile whose sweeps stopped at the tolerance carries their residual, amplified by the 4th difference */
                if (lastbrk < 0) {
                    if (cool > 0) {
                        --cool;
                        next_s = (imax * 65536.0 < stride_tol) ? 8 : (imax * 2048.0 < stride_tol) ? 4 : (imax * 64.0 < stride_tol) ? 2 : 1;
                    } else {
                        next_s = (imax > stride_tol) ? 1 : (opt_s ? opt_s : max_stride);
                        if (trouble >= 2 && next_s > 4) next_s = 4;
                    }
                }
                else {
                    /* a kink inside this tile: the history of a coarse successor must lie behind it */
                    const int first_clean = (lastbrk + 2) * spl, tail = keep - first_clean;   /* steps after the kink lane + 1 */
                    for (int e = first_clean > 0 ? first_clean : 0; e < keep; ++e) if (ind[e] > ipost) ipost = ind[e];
                    if (tail >= 3 * 8 + spl && ipost * 65536.0 < stride_tol) next_s = 8;
                    else if (tail >= 3 * 4 + spl && ipost * 2048.0 < stride_tol) next_s = 4;
                    else if (tail >= 3 * 2 + spl && ipost * 64.0 < stride_tol) next_s = 2;
                    else next_s = 1;
                }
            } else {
                /* (a tile over 8 intervals is held to a tenth of the bound: its promotion asks the same of the scaled indicator) */
                next_s = (imax * (s == 4 ? 640.0 : 64.0) < stride_tol && 2 * s <= max_stride) ? 2 * s : s;
                if (opt_s && next_s > opt_s) opt_s = next_s;
            }
            if (next_s > max_stride) next_s = max_stride;
        }
        /* 128-step tiles (spl < 4), adaptive stride: the sub-stepped start ends after a tile that was calm -- its indicator scaled
           to a step of a whole interval (8^5, margin 2: 65 536) and held to a hundredth stays below the bound everywhere, no kink in
           it (mp_eval.hpp MP_PRE_EARLY_END) */
        if (pre && mode == 1 && spl < 4 && max_stride > 1 && keep == nc && nc == TILE && sub_done + keep < pre_fine * MPO_PRE_SUB) {
            int calm = 1;
            for (int e = 0; e < keep; ++e) if (brk[e] || MPO_PRE_EARLY_END_FACTOR * ind[e] > stride_tol) calm = 0;
            if (calm) pre_fine = (sub_done + keep) / MPO_PRE_SUB;
        }
        if (pre) {
            next_s = (mode == 1 && sub_done + keep >= pre_fine * MPO_PRE_SUB) ? max_stride : 1;   /* optimistic after the sub-steps ... */
            /* ... as far as the last sub-stepped tile reaches back (the kernels' history is that tile's record: a successor over s
               intervals needs points 3 s intervals before its start) */
            const int reach = keep / MPO_PRE_SUB;
            while (next_s > 1 && 3 * next_s > reach) next_s /= 2;
        }
        /* ---- commit the kept steps: nodes, and the grid points they contain */
        mpo_node prev = nd[nn - 1];
        const mpo_node tile_start = prev;
        for (J = 0; J < keep; ++J) {
            if (pre) {
                const int k = sub_done + J + 1;
                if (k % MPO_PRE_SUB == 0) { if (Mout) Mout[k / MPO_PRE_SUB] = tn[J].M; if (Wout) Wout[k / MPO_PRE_SUB] = tn[J].w; }
            } else {
                const int ib = i0 + J * s;
                const double h = tn[J].t - prev.t;
                /* N[0] = the tile's start, N[j + 1] = step end j: the four nodes around this step (first / last step of the
                   kept part: the first / last four) */
#define NODE(j) ((j) == 0 ? &tile_start : &tn[(j) - 1])
                int l0 = J - 1;
                if (l0 + 3 > keep) l0 = keep - 3;
                if (l0 < 0) l0 = 0;
                for (int i = 1; i < s; ++i) {
                    const double tg = tgrid[ib + i], th = (tg - prev.t) / h;
                    /* Mdisc: the quintic while the step resolves the viscous time.  Beyond, Mdisc follows the fallback rate
                       quasi-steadily and dMdisc/dt = Mdotfb - Mdisc/tvisc, a difference of nearly equal terms, carries the
                       node's rounding amplified by h/tvisc: the smooth ratio Mdisc / (tvisc Mdotfb) = 1 + O(tvisc/t) is
                       interpolated instead, through four node values (1.7e-10 at a stride of 8 where h = tvisc, falling as 1/t;
                       the two-node cubic in Mdisc itself is 3e-9 off there).  Fewer than three kept steps: that cubic. */
                    if (Mout) {
                        if (h / w.tvisc < 1.0) Mout[ib + i] = hermite5(th, h, prev.M, prev.dM, prev.ddM, tn[J].M, tn[J].dM, tn[J].ddM);
                        else if (keep >= 3) {
                            double xl[4], rl[4];
                            for (int k = 0; k < 4; ++k) {
                                const mpo_node *nk = NODE(l0 + k);
                                xl[k] = nk->t;
                                rl[k] = nk->M / (nk->M + w.tvisc * nk->dM);              /* Mdisc / (tvisc Mdotfb) at the node */
                            }
                            Mout[ib + i] = lagrange4(xl, rl, tg) * w.tvisc * mdot_fb(&w, tg);
                        } else Mout[ib + i] = hermite(th, h, prev.M, prev.dM, tn[J].M, tn[J].dM);
                    }
                    if (Wout) Wout[ib + i] = hermite(th, h, prev.w, prev.f, tn[J].w, tn[J].f);
                }
#undef NODE
                if (Mout) Mout[ib + s] = tn[J].M;
                if (Wout) Wout[ib + s] = tn[J].w;
            }
            nd[nn++] = tn[J];
            prev = tn[J];
        }
        st_.steps_kept += keep;
        if (pre) { sub_done += keep; i0 = sub_done / MPO_PRE_SUB; } else i0 += keep * s;
        M = tn[keep - 1].M; om = tn[keep - 1].w; fcur = tn[keep - 1].f;
        s = next_s;
        last_was_pre = pre;
    }
    free(nd); free(tn); free(ind); free(brk);
    if (status != MPO_OK) for (int j = 0; j < n; ++j) { if (Mout) Mout[j] = NAN; if (Wout) Wout[j] = NAN; }
    if (stats) *stats = st_;
    return status;
}

/* The FIXED production scheme (mode 0).  nsub > 1: every step refined geometrically nsub-fold (convergence studies). */
int mpo_trajectory(const mpo_cfg *c, const double *pars, int ndim, const double *tgrid, int n,
                   int nsub, double *Mout, double *Wout) {
    if (nsub <= 1) return mpo_trajectory_mode(c, pars, ndim, tgrid, n, 0, 4, 0.0, 1, Mout, Wout, NULL);
    const int nf = (n - 1) * nsub + 1;
    const double q = exp(log(tgrid[n - 1] / tgrid[0]) / (double)(nf - 1));
    double *tf = (double *)malloc(sizeof(double) * 3 * (size_t)nf), *Mf = tf + nf, *Wf = tf + 2 * nf;
    for (int i = 0; i < nf; ++i) tf[i] = (i % nsub == 0) ? tgrid[i / nsub] : tgrid[0] * pow(q, (double)i);
    const int st = mpo_trajectory_mode(c, pars, ndim, tf, nf, 0, 4, 0.0, 1, Mf, Wf, NULL);
    for (int j = 0; j < n; ++j) { if (Mout) Mout[j] = Mf[(size_t)j * nsub]; if (Wout) Wout[j] = Wf[(size_t)j * nsub]; }
    free(tf);
    return st;
}

/* luminosities at one grid point, in erg/s (not yet /1e50) */
static void luminosity(const mpo_cfg *c, const wk *w, double Mdisc, double omega,
                       double *Ltot, double *Lprop_o, double *Ldip_o) {
    flow f;
    flow_state(c, w, c->n_lum, Mdisc, omega, &f);
    double Nacc;
    if (f.rot_param > c->nacc_lum_threshold) {
        Nacc = 0.0;
    } else if (f.Rm >= R_) {
        Nacc = pow(w->GM * f.Rm, 0.5) * (f.Mdotacc - f.Mdotprop);
    } else {
        Nacc = pow(w->GM * R_, 0.5) * (f.Mdotacc - f.Mdotprop);
    }
    double Ldip = w->dipeff * ((pow(w->mu, 2.0) * pow(omega, 4.0)) / (6.0 * pow(C_, 3.0)));
    if (Ldip <= 0.0) Ldip = 0.0;
    if (!isfinite(Ldip)) Ldip = 0.0;
    double Lprop;
    if (c->lprop_gm_term)
        Lprop = w->propeff * ((-1.0 * Nacc * omega) - ((w->GM / f.Rm) * f.eta2 * (Mdisc / w->tvisc)));
    else
        Lprop = w->propeff * (-1.0 * Nacc * omega);
    if (Lprop <= 0.0) Lprop = 0.0;
    if (!isfinite(Lprop)) Lprop = 0.0;
    *Ltot = w->f_beam * (Ldip + Lprop);
    *Lprop_o = Lprop;
    *Ldip_o = Ldip;
}

/*
 * model_lc / model_lum with xdata=None: out[4][n] = tarr, Ltot, Lprop, Ldip, luminosities /1e50.
 * traj (may be NULL) = [2][n] Mdisc, omega.  pars are PHYSICAL (no un-logging).
 */
int mpo_model_lc(const mpo_cfg *c, const double *pars, int ndim, const double *tgrid, int n,
                 int nsub, double *out, double *traj) {
    double *M = (double *)malloc(sizeof(double) * 2 * (size_t)n), *W = M + n;
    int st = mpo_trajectory(c, pars, ndim, tgrid, n, nsub, M, W);
    wk w;
    walker_setup(c, pars, ndim, &w);
    for (int i = 0; i < n; ++i) {
        double lt = NAN, lp = NAN, ld = NAN;
        if (st == MPO_OK) { luminosity(c, &w, M[i], W[i], &lt, &lp, &ld); lt /= 1.0e50; lp /= 1.0e50; ld /= 1.0e50; }
        out[i] = tgrid[i]; out[n + i] = lt; out[2 * n + i] = lp; out[3 * n + i] = ld;
    }
    if (traj) memcpy(traj, M, sizeof(double) * 2 * (size_t)n);
    free(M);
    return st;
}

/* np.interp semantics (what scipy's interp1d(kind='linear') evaluates): x must lie inside the grid */
static double interp_lin(const double *xp, const double *fp, int n, double x) {
    if (x == xp[n - 1]) return fp[n - 1];
    int lo = 0, hi = n - 1; /* invariant xp[lo] <= x < xp[hi] */
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (xp[mid] <= x) lo = mid; else hi = mid; }
    double slope = (fp[lo + 1] - fp[lo]) / (xp[lo + 1] - xp[lo]);
    return slope * (x - xp[lo]) + fp[lo];
}

/* lnlike for PHYSICAL parameters.  *status: MPO_OK / MPO_FLAG / MPO_NONFINITE.  -inf on failure. */
double mpo_lnlike_mode(const mpo_cfg *c, const double *pars, int ndim, const double *tgrid, int n,
                       const double *x, const double *y, const double *yerr, int nobs, int *status, int mode, int spl,
                       mpo_stats *stats) {
    double *M = (double *)malloc(sizeof(double) * 3 * (size_t)n), *W = M + n, *L = M + 2 * n;
    int st = mpo_trajectory_mode(c, pars, ndim, tgrid, n, mode, spl, 0.0, 0, M, W, stats);
    double ll = -INFINITY;
    if (st == MPO_OK) {
        wk w;
        walker_setup(c, pars, ndim, &w);
        for (int i = 0; i < n; ++i) { double lp, ld; luminosity(c, &w, M[i], W[i], &L[i], &lp, &ld); }
        double acc = 0.0;
        for (int j = 0; j < nobs; ++j) {
            double mod = interp_lin(tgrid, L, n, x[j]) / 1.0e50;
            double r = (y[j] - mod) / yerr[j];
            acc += r * r;
        }
        ll = -0.5 * acc;
        if (!isfinite(ll)) { ll = -INFINITY; st = MPO_NONFINITE; }
    }
    free(M);
    if (status) *status = st;
    return ll;
}

double mpo_lnlike(const mpo_cfg *c, const double *pars, int ndim, const double *tgrid, int n,
                  const double *x, const double *y, const double *yerr, int nobs, int *status) {
    return mpo_lnlike_mode(c, pars, ndim, tgrid, n, x, y, yerr, nobs, status, 0, 4, NULL);
}

/*
 * lnprob in sampler coordinates: inclusive box prior, un-log the coordinates in log_mask, lnlike.
 * nprior = 0 disables the prior.  mode / spl: mpo_trajectory_mode (0 = fixed steps, 1 = adaptive stride; lanes own spl steps).
 */
double mpo_lnprob_mode(const mpo_cfg *c, const double *pars, int ndim, const double *lower, const double *upper,
                       int nprior, uint32_t log_mask, const double *tgrid, int n,
                       const double *x, const double *y, const double *yerr, int nobs, int *status, int mode, int spl,
                       mpo_stats *stats) {
    double p[9];
    if (stats) memset(stats, 0, sizeof *stats);
    for (int i = 0; i < nprior; ++i) {
        if (!(pars[i] >= lower[i]) || !(pars[i] <= upper[i])) { if (status) *status = MPO_PRIOR; return -INFINITY; }
    }
    for (int i = 0; i < ndim; ++i) p[i] = (log_mask >> i) & 1u ? pow(10.0, pars[i]) : pars[i];
    return mpo_lnlike_mode(c, p, ndim, tgrid, n, x, y, yerr, nobs, status, mode, spl, stats);
}

double mpo_lnprob(const mpo_cfg *c, const double *pars, int ndim, const double *lower, const double *upper,
                  int nprior, uint32_t log_mask, const double *tgrid, int n,
                  const double *x, const double *y, const double *yerr, int nobs, int *status) {
    return mpo_lnprob_mode(c, pars, ndim, lower, upper, nprior, log_mask, tgrid, n, x, y, yerr, nobs, status, 0, 4, NULL);
}

/* tiles[nwalk][3] (optional): tile solves, tiles cut short or redone, steps kept */
void mpo_lnprob_batch_mode(const mpo_cfg *c, const double *pars, int nwalk, int ndim, const double *lower,
                           const double *upper, int nprior, uint32_t log_mask, const double *tgrid, int n,
                           const double *x, const double *y, const double *yerr, int nobs,
                           double *lnprob, int32_t *status, int mode, int spl, int32_t *tiles) {
    for (int i = 0; i < nwalk; ++i) {
        int st = 0;
        mpo_stats ss;
        lnprob[i] = mpo_lnprob_mode(c, pars + (size_t)i * ndim, ndim, lower, upper, nprior, log_mask, tgrid, n,
                                    x, y, yerr, nobs, &st, mode, spl, &ss);
        if (status) status[i] = st;
        if (tiles) { tiles[3 * i] = ss.tiles; tiles[3 * i + 1] = ss.tiles_cut; tiles[3 * i + 2] = ss.steps_kept; }
    }
}

void mpo_lnprob_batch(const mpo_cfg *c, const double *pars, int nwalk, int ndim, const double *lower,
                      const double *upper, int nprior, uint32_t log_mask, const double *tgrid, int n,
                      const double *x, const double *y, const double *yerr, int nobs,
                      double *lnprob, int32_t *status) {
    mpo_lnprob_batch_mode(c, pars, nwalk, ndim, lower, upper, nprior, log_mask, tgrid, n, x, y, yerr, nobs, lnprob, status,
                          0, 4, NULL);
}
