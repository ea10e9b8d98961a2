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
    int32_t lprop_gm_term, reserved;
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

/* phi_1..4(z): Taylor below |z| = 0.5, closed forms above */
static void phi1234(double z, double *ez, double ph[4]) {
    if (fabs(z) < 0.5) {
        double s = 0.0, term = 1.0 / 24.0;
        for (int k = 0; k < 22; ++k) { s += term; term *= z / (double)(k + 5); }
        ph[3] = s; ph[2] = z * s + 1.0 / 6.0; ph[1] = z * ph[2] + 0.5; ph[0] = z * ph[1] + 1.0; *ez = z * ph[0] + 1.0;
    } else {
        *ez = exp(z);
        ph[0] = expm1(z) / z;
        ph[1] = (ph[0] - 1.0) / z;
        ph[2] = (ph[1] - 0.5) / z;
        ph[3] = (ph[2] - 1.0 / 6.0) / z;
    }
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

/*
 * PRODUCTION SCHEME (the one the HIP kernel implements): exponential Adams-Moulton of order 4 on the
 * geometric output grid, one step per grid interval.
 *   Mdisc_{j+1} = e^{-z} Mdisc_j + h sum_m phi_{m+1}(-z) sum_k Wm[k][m] Mdotfb(t_{j+1-k}),        z = h/tvisc
 *   omega_{j+1} = e^{h lam} omega_j + h sum_m phi_{m+1}(h lam) sum_k Wm[k][m] (f_{j+1-k} - lam omega_{j+1-k})
 * with f_p = omega_dot(Mdisc_p, omega_p) and lam = d(omega_dot)/d(omega) frozen at the NEW point
 * (t_{j+1}, omega_{j+1}): implicit in omega_{j+1}, solved by fixed-point iteration (contraction ~ h*|d lam|).
 * History before the first grid point: Mdotfb is analytic (the grid is continued geometrically backwards);
 * for (f, omega) the two missing points are the linear continuation in the step index through points 0 and 1.
 * Failure ('flag'): the rotation parameter of any iterate at a grid point exceeds 0.27.
 * nsub > 1 refines the grid geometrically (convergence studies only).
 */
int mpo_trajectory(const mpo_cfg *c, const double *pars, int ndim, const double *tgrid, int n,
                   int nsub, double *Mout, double *Wout) {
    wk w;
    walker_setup(c, pars, ndim, &w);
    if (nsub < 1) nsub = 1;
    const int nf = (n - 1) * nsub + 1;
    const double q = exp(log(tgrid[n - 1] / tgrid[0]) / (double)(nf - 1));
    double Wm[4][4];
    mpo_eam4_weights(q, Wm);
    double *tf = (double *)malloc(sizeof(double) * (size_t)nf);
    if (nsub == 1) memcpy(tf, tgrid, sizeof(double) * (size_t)n);
    else for (int i = 0; i < nf; ++i) tf[i] = (i % nsub == 0) ? tgrid[i / nsub] : tgrid[0] * pow(q, (double)i);
    int status = MPO_OK;
    for (int j = 0; j < n; ++j) { if (Mout) Mout[j] = NAN; if (Wout) Wout[j] = NAN; } /* undefined after a failure */
    double M = pars[2] * MSOL_;                        /* code/synthetic_datasets/funcs.py:66-69 */
    double om = (2.0 * M_PI) / (1.0e-3 * pars[1]);
    /* history: index 0 = point j, 1 = j-1, 2 = j-2 */
    double Sh[3] = {mdot_fb(&w, tf[0]), mdot_fb(&w, tf[0] / q), mdot_fb(&w, tf[0] / (q * q))};
    double fh[3] = {0.0, 0.0, 0.0}, wh[3] = {0.0, 0.0, 0.0}, rot;
    fh[0] = omega_dot(c, &w, M, om, &rot, NULL);
    wh[0] = om;
    double f0 = fh[0], w0 = om, f1 = 0.0, w1 = 0.0; /* points 0 and 1, for the start-up ghosts */
    for (int i = 0; i < nf; ++i) {
        if (i % nsub == 0) { if (Mout) Mout[i / nsub] = M; if (Wout) Wout[i / nsub] = om; }
        if (!(isfinite(M) && isfinite(om)) || M <= 0.0 || om <= 0.0) status = MPO_NONFINITE;
        else if ((0.5 * w.I * om * om) / w.modW > 0.27) status = MPO_FLAG;
        if (status != MPO_OK || i == nf - 1) break;
        const double h = tf[i + 1] - tf[i];
        /* ---- Mdisc */
        const double S1 = mdot_fb(&w, tf[i + 1]);
        double ez, ph[4];
        phi1234(-h / w.tvisc, &ez, ph);
        double acc = 0.0;
        for (int m = 0; m < 4; ++m)
            acc += ph[m] * (Wm[0][m] * S1 + Wm[1][m] * Sh[0] + Wm[2][m] * Sh[1] + Wm[3][m] * Sh[2]);
        const double M1 = ez * M + h * acc;
        /* ---- omega: fixed-point iteration on the implicit step */
        double wn = om + h * fh[0], fnew = 0.0;
        int flagged = 0;
        for (int it = 0; it < 200; ++it) {
            double lam, r1;
            fnew = omega_dot(c, &w, M1, wn, &r1, &lam);
            if (r1 > 0.27) { flagged = 1; break; }
            double hf[2], hw[2]; /* points j-1, j-2 (ghosts during start-up) */
            if (i == 0) { hf[0] = 2.0 * f0 - fnew; hw[0] = 2.0 * w0 - wn; hf[1] = 3.0 * f0 - 2.0 * fnew; hw[1] = 3.0 * w0 - 2.0 * wn; }
            else if (i == 1) { hf[0] = fh[1]; hw[0] = wh[1]; hf[1] = 2.0 * f0 - f1; hw[1] = 2.0 * w0 - w1; }
            else { hf[0] = fh[1]; hw[0] = wh[1]; hf[1] = fh[2]; hw[1] = wh[2]; }
            const double N0 = fnew - lam * wn, N1 = fh[0] - lam * wh[0], N2 = hf[0] - lam * hw[0], N3 = hf[1] - lam * hw[1];
            phi1234(h * lam, &ez, ph);
            acc = 0.0;
            for (int m = 0; m < 4; ++m) acc += ph[m] * (Wm[0][m] * N0 + Wm[1][m] * N1 + Wm[2][m] * N2 + Wm[3][m] * N3);
            const double wnew = ez * om + h * acc;
            const double d = fabs(wnew - wn);
            wn = wnew;
            if (!(d > 1e-15 * fabs(wn))) break; /* converged, or NaN */
        }
        if (flagged) { status = MPO_FLAG; break; }
        fnew = omega_dot(c, &w, M1, wn, &rot, NULL);
        if (i == 0) { f1 = fnew; w1 = wn; }
        Sh[2] = Sh[1]; Sh[1] = Sh[0]; Sh[0] = S1;
        fh[2] = fh[1]; fh[1] = fh[0]; fh[0] = fnew;
        wh[2] = wh[1]; wh[1] = wh[0]; wh[0] = wn;
        M = M1;
        om = wn;
    }
    free(tf);
    return status;
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
double mpo_lnlike(const mpo_cfg *c, const double *pars, int ndim, const double *tgrid, int n,
                  const double *x, const double *y, const double *yerr, int nobs, int *status) {
    double *M = (double *)malloc(sizeof(double) * 3 * (size_t)n), *W = M + n, *L = M + 2 * n;
    int st = mpo_trajectory(c, pars, ndim, tgrid, n, 1, M, W);
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

/*
 * lnprob in sampler coordinates: inclusive box prior, un-log the coordinates in log_mask, lnlike.
 * nprior = 0 disables the prior.
 */
double mpo_lnprob(const mpo_cfg *c, const double *pars, int ndim, const double *lower, const double *upper,
                  int nprior, uint32_t log_mask, const double *tgrid, int n,
                  const double *x, const double *y, const double *yerr, int nobs, int *status) {
    double p[9];
    for (int i = 0; i < nprior; ++i) {
        if (!(pars[i] >= lower[i]) || !(pars[i] <= upper[i])) { if (status) *status = MPO_PRIOR; return -INFINITY; }
    }
    for (int i = 0; i < ndim; ++i) p[i] = (log_mask >> i) & 1u ? pow(10.0, pars[i]) : pars[i];
    return mpo_lnlike(c, p, ndim, tgrid, n, x, y, yerr, nobs, status);
}

void mpo_lnprob_batch(const mpo_cfg *c, const double *pars, int nwalk, int ndim, const double *lower,
                      const double *upper, int nprior, uint32_t log_mask, const double *tgrid, int n,
                      const double *x, const double *y, const double *yerr, int nobs,
                      double *lnprob, int32_t *status) {
    for (int i = 0; i < nwalk; ++i) {
        int st = 0;
        lnprob[i] = mpo_lnprob(c, pars + (size_t)i * ndim, ndim, lower, upper, nprior, log_mask, tgrid, n,
                               x, y, yerr, nobs, &st);
        if (status) status[i] = st;
    }
}
