// mp_capi.cpp — host side of the C ABI declared in include/magprop_amd.h.
//
// Plain HIP runtime only (no torch, no hipBLAS): device buffers, one stream per handle, the
// host-side digestion of observed light curves into the tile-bucketed layout the kernel reads,
// and thin launch wrappers.  There is NO CPU fallback: without a HIP device mp_create() fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "mp_device.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(MP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DeviceScope {  // make the handle's device current for the duration of a call
    int prev = -1;
    bool ok = true;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

struct HostDataset {
    bool set = false;
    std::vector<int32_t> g, tile_ptr;
    std::vector<double> dx, idt, y, yerr;
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return MP_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max<size_t>(n, 16);
        HIP_TRY(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        return MP_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Page-locked host staging area of the host-buffer entry points: copies to and from it are true asynchronous DMA
// (pageable user buffers would be staged by the runtime copy by copy).
struct PinnedBuf {
    unsigned char *p = nullptr;
    unsigned char *dev = nullptr;   // the same memory as the device sees it (mapped, coherent): kernels may read and write it in place
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return MP_OK;
        if (p) (void)hipHostFree(p);
        p = dev = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(n, 4096);
        HIP_TRY(hipHostMalloc((void **)&p, want, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer((void **)&dev, p, 0));
        cap = want;
        return MP_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = dev = nullptr;
        cap = 0;
    }
};

}  // namespace

struct mp_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<double> tgrid;
    int n_tiles = 0;
    HostDataset ds[MP_MAX_DATASETS];
    mp::DevShared sh{};
    // device copies of the shared data
    DevBuf<double> d_wtab;
    DevBuf<double> d_tgrid, d_obs_dx, d_obs_idt, d_obs_y, d_obs_yerr;
    DevBuf<int32_t> d_obs_g, d_tile_ptr;
    DevBuf<mp::DsDesc> d_ds;
    // The packed observation arrays are an append-only arena: a new light curve goes behind the last one (obs_used /
    // tp_used entries are live or stale), a replaced one leaves its old entries behind as garbage until the next rebuild.
    size_t obs_used = 0, tp_used = 0;
    std::vector<mp::DsDesc> desc;   // host mirror of d_ds
    // workspace of the host-buffer entry points
    DevBuf<double> w_pars, w_lnprob, w_curves;
    DevBuf<int32_t> w_dsid, w_status, w_sweeps;
    double last_mean_tiles = 0.0;
    bool tile_log_on = false;
    DevBuf<int32_t> w_tile_log;
    std::vector<int32_t> last_tile_log;
    PinnedBuf h_io;               // mp_lnprob_batch: [pars | ds_id] in, [lnprob | status | sweeps | tiles] out, read and written in place by the kernel
    double last_mean_sweeps = 0.0;
    std::vector<int32_t> last_sweeps, last_tiles;   // per walker, most recent host-buffer batch (diagnostic)
    // Launch order of mixed-length batches (mp_kernels.hip order_kernel): a ring of index buffers, one per launch in flight.
    // A slot is written by the launch that takes it and read by that launch's workgroups as they start; the launch that
    // takes it kOrderRing launches later waits (stream-level, on the event recorded behind the earlier launch) for that
    // launch to have finished -- launches fewer than kOrderRing apart share nothing.
    static constexpr int kOrderRing = 8;
    DevBuf<int32_t> order[kOrderRing];
    hipEvent_t order_done[kOrderRing] = {};
    unsigned order_next = 0;
    // Threading / stream contract (include/magprop_amd.h): every entry point that takes a handle or a sampler holds
    // `mu` for its duration.  Launches share nothing writable but their own outputs (round 4: no per-walker scratch rows),
    // so launches of one handle on different streams may overlap freely.
    std::recursive_mutex mu;
    // Multi-device handle (mp_create_multi): one evaluator per listed device; this object then holds no device state of
    // its own -- datasets and prior are forwarded to every evaluator, a host-buffer batch is dealt out in contiguous blocks.
    std::vector<mp_handle *> sub;
    int pend_n = 0;               // rows of the host-buffer batch between batch_begin and batch_end
    size_t pend_in_bytes = 0;
    double last_tot_sweeps = 0.0, last_tot_tiles = 0.0;   // over the walkers of the last batch that finished (status ok) ...
    int last_cnt_ok = 0;                                  // ... and how many those were
};

namespace {
using Lock = std::lock_guard<std::recursive_mutex>;

// Launch the log-posterior kernel over a batch.  A batch that refers to light curves of more than 64 points next to short
// ones and needs more than one round of the device's wave slots (two per SIMD) is evaluated longest light curves first.
int launch_lnprob_ordered(mp_handle *h, const mp::LaunchArgs &a_in, hipStream_t st) {
    mp::LaunchArgs a = a_in;
    int slot = -1;
    const bool curves = a.ltot || a.lprop || a.ldip || a.mdisc || a.omega;
    if (h->sh.has_long && h->sh.n_ds > 1 && a.ds_id && a.want_chi2 && !curves && a.n > 2 * h->sh.n_simd) {
        slot = (int)(h->order_next++ % mp_handle::kOrderRing);
        if (h->order[slot].cap < (size_t)a.n) {
            // (growing frees the old buffer, which waits for the device: nothing in flight reads it any more)
            const int rc = h->order[slot].ensure((size_t)a.n);
            if (rc) return rc;
        }
        if (!h->order_done[slot]) HIP_TRY(hipEventCreateWithFlags(&h->order_done[slot], hipEventDisableTiming));
        else HIP_TRY(hipStreamWaitEvent(st, h->order_done[slot], 0));
        const int eo = mp::launch_order(h->sh, a.ds_id, a.n, h->order[slot].p, (void *)st);
        if (eo) return fail(MP_EHIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)eo));
        a.order = h->order[slot].p;
    }
    const int e = mp::launch_lnprob(h->sh, a, (void *)st);
    if (e) return fail(MP_EHIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    if (slot >= 0) HIP_TRY(hipEventRecord(h->order_done[slot], st));
    return MP_OK;
}
}  // namespace

static void publish_datasets(mp_handle *h) {
    int n_ds = 0, extra = 0;
    for (int d = 0; d < MP_MAX_DATASETS; ++d)
        if (h->ds[d].set) { n_ds = d + 1; extra = std::max(extra, (int)h->ds[d].g.size() - 64); }
    h->sh.ds = h->d_ds.p;
    h->sh.n_ds = n_ds;
    h->sh.tile_ptr = h->d_tile_ptr.p;
    h->sh.obs_g = h->d_obs_g.p;
    h->sh.obs_dx = h->d_obs_dx.p;
    h->sh.obs_idt = h->d_obs_idt.p;
    h->sh.obs_y = h->d_obs_y.p;
    h->sh.obs_yerr = h->d_obs_yerr.p;
    h->sh.has_long = extra > 0 ? 1 : 0;
}

// Copy light curve d behind the arena's last entry and publish its descriptor (slots never set keep n_obs = 0, which
// the kernels answer with MP_STATUS_BADDATASET).  Nothing in flight reads the target regions: no synchronisation.
static int append_dataset(mp_handle *h, int d) {
    const HostDataset &s = h->ds[d];
    const size_t n = s.g.size(), o = h->obs_used, t = h->tp_used;
    HIP_TRY(hipMemcpy(h->d_obs_g.p + o, s.g.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_obs_dx.p + o, s.dx.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_obs_idt.p + o, s.idt.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_obs_y.p + o, s.y.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_obs_yerr.p + o, s.yerr.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_tile_ptr.p + t, s.tile_ptr.data(), s.tile_ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    int32_t flags = mp::kDsOnKnots;
    for (size_t j = 0; j < n; ++j) if (s.dx[j] != 0.0) { flags = 0; break; }
    h->desc[d] = mp::DsDesc{(int32_t)n, (int32_t)o, (int32_t)t, flags};
    HIP_TRY(hipMemcpy(h->d_ds.p + d, &h->desc[d], sizeof(mp::DsDesc), hipMemcpyHostToDevice));
    h->obs_used = o + n;
    h->tp_used = t + s.tile_ptr.size();
    return MP_OK;
}

// Rebuild the arena from the host copies (first use, growth, or a replaced slot): waits for the device, because
// kernels in flight may still read the old buffers, and leaves room for the sets to come.
static int rebuild_datasets(mp_handle *h) {
    size_t n_obs = 0, n_tp = 0;
    for (int d = 0; d < MP_MAX_DATASETS; ++d)
        if (h->ds[d].set) { n_obs += h->ds[d].g.size(); n_tp += h->ds[d].tile_ptr.size(); }
    HIP_TRY(hipDeviceSynchronize());
    const size_t cap_obs = std::max<size_t>(2 * n_obs, 4096), cap_tp = std::max<size_t>(2 * n_tp, 8 * ((size_t)h->n_tiles + 1));
    int rc;
    if ((rc = h->d_obs_g.ensure(cap_obs)) || (rc = h->d_obs_dx.ensure(cap_obs)) || (rc = h->d_obs_idt.ensure(cap_obs)) ||
        (rc = h->d_obs_y.ensure(cap_obs)) || (rc = h->d_obs_yerr.ensure(cap_obs)) || (rc = h->d_tile_ptr.ensure(cap_tp)) ||
        (rc = h->d_ds.ensure(MP_MAX_DATASETS)))
        return rc;
    h->desc.assign(MP_MAX_DATASETS, mp::DsDesc{0, 0, 0, 0});
    HIP_TRY(hipMemset(h->d_ds.p, 0, MP_MAX_DATASETS * sizeof(mp::DsDesc)));
    h->obs_used = h->tp_used = 0;
    for (int d = 0; d < MP_MAX_DATASETS; ++d)
        if (h->ds[d].set && (rc = append_dataset(h, d))) return rc;
    publish_datasets(h);
    return MP_OK;
}

// After h->ds[d] has been (re)set on the host.
static int upload_dataset(mp_handle *h, int d, bool replaced) {
    const HostDataset &s = h->ds[d];
    const bool fits = h->d_ds.p && h->obs_used + s.g.size() <= h->d_obs_g.cap && h->tp_used + s.tile_ptr.size() <= h->d_tile_ptr.cap;
    if (replaced || !fits) return rebuild_datasets(h);
    const int rc = append_dataset(h, d);
    if (rc) return rc;
    publish_datasets(h);
    return MP_OK;
}

extern "C" {

int mp_abi_version(void) { return MP_ABI_VERSION; }

const char *mp_last_error(void) { return g_err.c_str(); }

void mp_cfg_synth(mp_model_cfg *c) {
    if (!c) return;
    *c = mp_model_cfg{0.35, 3.0, 10.0, 10.0, 0.1, 1.0, 0.9, 1.0, 1.0, 1.0, 0.27, 1, 0, 0.0, 0.0, 0, 0};
}

void mp_cfg_lib(mp_model_cfg *c) {
    if (!c) return;
    *c = mp_model_cfg{0.8, 1.0, 1.0, 1.0, 0.1, 1.0, 0.9, 0.05, 0.4, 1.0, 0.0, 0, 0, 0.0, 0.0, 0, 0};
}

mp_handle *mp_create(const mp_model_cfg *cfg, const double *tgrid, int n_grid, int device) {
    if (!cfg || !tgrid || n_grid < 2) {
        fail(MP_EINVAL, "mp_create: cfg/tgrid NULL or n_grid < 2");
        return nullptr;
    }
    for (int i = 1; i < n_grid; ++i)
        if (!(tgrid[i] > tgrid[i - 1]) || !std::isfinite(tgrid[i])) {
            fail(MP_EINVAL, "mp_create: tgrid must be finite and strictly increasing (index %d)", i);
            return nullptr;
        }
    // The integrator is built on a geometric grid (np.logspace), the only kind the reference uses
    // (magnetar/funcs.py:132-137, code/synthetic_datasets/funcs.py:19).
    if (!(tgrid[0] > 0.0)) {
        fail(MP_EINVAL, "mp_create: tgrid must be positive");
        return nullptr;
    }
    // the kernels generate the step end times themselves, t_i = t_0 q^i: the grid has to be geometric to rounding
    const double lnq_check = std::log(tgrid[n_grid - 1] / tgrid[0]) / (double)(n_grid - 1);
    for (int i = 1; i < n_grid; ++i)
        if (std::fabs(tgrid[i] / (tgrid[0] * std::exp((double)i * lnq_check)) - 1.0) > 1.0e-12) {
            fail(MP_EINVAL, "mp_create: tgrid must be log-spaced (np.logspace); it leaves t0*q^i at index %d", i);
            return nullptr;
        }
    if (!(cfg->inertia_factor > 0) || !(cfg->alpha > 0) || !(cfg->cs7 > 0) || !(cfg->k > 0) ||
        !(cfg->rm_massflow_factor > 0)) {
        fail(MP_EINVAL, "mp_create: non-positive model constant in cfg");
        return nullptr;
    }
    if (!(cfg->sweep_tol >= 0.0) || cfg->sweep_tol > 1.0e-3) {
        fail(MP_EINVAL, "mp_create: cfg.sweep_tol must be 0 (library default) or in (0, 1e-3]");
        return nullptr;
    }
    if (!(cfg->stride_tol >= 0.0) || cfg->stride_tol > 1.0e-3 ||
        !(cfg->max_stride == 0 || cfg->max_stride == 1 || cfg->max_stride == 2 || cfg->max_stride == 4 ||
          cfg->max_stride == 8)) {
        fail(MP_EINVAL, "mp_create: cfg.max_stride must be 0, 1, 2, 4 or 8 and cfg.stride_tol 0 or in (0, 1e-3]");
        return nullptr;
    }
    if (cfg->dipole_torque != 0 && cfg->dipole_torque != 1) {
        fail(MP_EINVAL, "mp_create: cfg.dipole_torque must be 0 (the packages' dipole torque) or 1 (code/figure_3.py's Bucciantini law)");
        return nullptr;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        fail(MP_ENODEV, "mp_create: no HIP device visible (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    if (device >= count) {
        fail(MP_ENODEV, "mp_create: device %d out of range (%d visible)", device, count);
        return nullptr;
    }
    mp_handle *h = new mp_handle();
    h->device = device;
    DeviceScope scope(device);
    hipDeviceProp_t prop;
    if (!scope.ok || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipGetDeviceProperties(&prop, device) != hipSuccess) {
        fail(MP_EHIP, "mp_create: cannot select device %d / create stream", device);
        mp_destroy(h);
        return nullptr;
    }
    h->tgrid.assign(tgrid, tgrid + n_grid);
    h->n_tiles = (n_grid - 1 + mp::kTile - 1) / mp::kTile;
    if (h->d_tgrid.ensure((size_t)n_grid) != MP_OK ||
        hipMemcpy(h->d_tgrid.p, tgrid, sizeof(double) * (size_t)n_grid, hipMemcpyHostToDevice) != hipSuccess) {
        fail(MP_EHIP, "mp_create: cannot upload the time grid");
        mp_destroy(h);
        return nullptr;
    }
    mp::DevShared &s = h->sh;
    s.tgrid = h->d_tgrid.p;
    s.n_grid = n_grid;
    s.n_tiles = h->n_tiles;
    s.cfg = *cfg;
    s.n_prior = 0;
    s.log_mask = 0;
    // star constants, magnetar/funcs.py:7-13,75-76
    const double M = 1.4 * mp::kMsol;
    s.GM = mp::kG * M;
    s.inertia = cfg->inertia_factor * M * mp::kR * mp::kR;
    s.inv_inertia = 1.0 / s.inertia;
    const double beta = s.GM / (mp::kR * mp::kC * mp::kC);
    const double modW = 0.6 * M * mp::kC * mp::kC * (beta / (1.0 - 0.5 * beta));
    s.crot = 0.5 * s.inertia / modW;
    s.sqrtGM = std::sqrt(s.GM);
    s.inv_sqrtGM = 1.0 / s.sqrtGM;
    s.sqrtR = std::sqrt(mp::kR);
    s.crm_unit = std::pow(1.0e15 * mp::kR * mp::kR * mp::kR, 4.0 / 7.0) * std::pow(s.GM, -1.0 / 7.0) *
                 std::pow(cfg->rm_massflow_factor, -2.0 / 7.0);
    s.t0 = tgrid[0];
    const double lnq = std::log(tgrid[n_grid - 1] / tgrid[0]) / (double)(n_grid - 1);
    s.lnq8 = lnq / 8.0;
    s.pre_fine = std::min(32, n_grid - 1);                         // oracle/mp_oracle.c MPO_PRE_FINE
    s.sweep_tol = cfg->sweep_tol > 0.0 ? cfg->sweep_tol : MP_SWEEP_TOL_DEFAULT;
    s.stride_tol = cfg->stride_tol > 0.0 ? cfg->stride_tol : MP_STRIDE_TOL_DEFAULT;
    {
        const int ms = cfg->max_stride > 0 ? cfg->max_stride : MP_MAX_STRIDE_DEFAULT;
        s.max_kind = ms == 1 ? 1 : (ms == 2 ? 2 : (ms == 4 ? 3 : 4));
    }
    s.n_simd = std::max(1, prop.multiProcessorCount) * 4;         // 4 SIMDs per CU (1 024 on MI355X)
    s.force_spl = 0;
    s.force_waves = 0;
    // The constants of the stride policy (oracle/mp_oracle.c carries the same ones).  They are not
    // settings: the shipped library takes them from here only, mp_get_policy() reports them.
    s.coarse_max_sweeps = 5;
    s.fine_max_sweeps = 12;                                        // (round 3: 8.  A tile over single intervals that is cut after 8 sweeps
                                                                   // restarts with a fresh extrapolation inside the slow zone -- where the
                                                                   // Jacobian changes from step to step and the converged region grows by
                                                                   // about one lane per sweep -- and the restart can be far worse than going
                                                                   // on: 1 walker in 1 000 prior-wide ones then needed 120 - 160 sweeps in one
                                                                   // tile.  12: worst walker 83 sweeps, means unchanged; profiles/r04_tail.log)
    s.trouble_limit = 2;
    s.early_hold_t = MP_EARLY_HOLD_SECONDS;                       // oracle/mp_oracle.c MPO_EARLY_HOLD_SECONDS
    s.coarse_tol_factor = 0.1;
    s.k4_tol_factor = 0.1;
    s.stop_factor = MP_STOP_FACTOR;                               // (mp_eval.hpp: the end of a tile's sweeps)
#ifdef MP_EXPERIMENTS
    // Developer build only (`make -C magprop_amd/csrc experiments` -> libmagprop_amd_exp.so, selected with MAGPROP_AMD_LIB):
    // environment overrides of the policy constants for A/B runs (tools/).  The shipped library does not read the
    // environment, so no benchmark or soak artefact can be produced with loosened settings and leave no trace
    // (mp_get_policy()[MP_POLICY_EXPERIMENTS] tells the two builds apart; bench.py prints it).
    {
        auto env_i = [](const char *name, int lo, int hi, int32_t &dst) { if (const char *e = std::getenv(name)) { const int v = std::atoi(e); if (v >= lo && v <= hi) dst = v; } };
        auto env_d = [](const char *name, double lo, double hi, double &dst) { if (const char *e = std::getenv(name)) { const double v = std::atof(e); if (v >= lo && v <= hi) dst = v; } };
        if (const char *e = std::getenv("MAGPROP_AMD_MAX_STRIDE")) {
            const int v = std::atoi(e);
            if (v == 1 || v == 2 || v == 4 || v == 8) s.max_kind = v == 1 ? 1 : (v == 2 ? 2 : (v == 4 ? 3 : 4));
        }
        if (const char *e = std::getenv("MAGPROP_AMD_SPL")) { const int v = std::atoi(e); if (v == 2 || v == 4) s.force_spl = v; }
        if (const char *e = std::getenv("MAGPROP_AMD_WAVES")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4) s.force_waves = v; }
        env_d("MAGPROP_AMD_SWEEP_TOL", 1.0e-14, 1.0e-3, s.sweep_tol);       // (the cfg validation range)
        env_d("MAGPROP_AMD_STRIDE_TOL", 1.0e-14, 1.0e-3, s.stride_tol);
        env_d("MAGPROP_AMD_EARLY_HOLD_SECONDS", 0.0, 1.0e6, s.early_hold_t);
        env_i("MAGPROP_AMD_TROUBLE_LIMIT", 0, 1000, s.trouble_limit);
        env_i("MAGPROP_AMD_COARSE_MAX_SWEEPS", 2, 64, s.coarse_max_sweeps);
        env_i("MAGPROP_AMD_FINE_MAX_SWEEPS", 2, 272, s.fine_max_sweeps);
        env_d("MAGPROP_AMD_COARSE_TOL_FACTOR", 1.0e-3, 1.0, s.coarse_tol_factor);
        env_d("MAGPROP_AMD_K4_TOL_FACTOR", 1.0e-3, 1.0, s.k4_tol_factor);
        env_d("MAGPROP_AMD_STOP_FACTOR", 0.0, 1.0, s.stop_factor);        // (0: every tile runs its verification sweep)
    }
#endif
    // Constants of the tile kinds: steps over 1/8, 1, 2, 4, 8 grid intervals (mp_device.h StrideK).
    // Quadrature matrices of the exponential Adams-Moulton formulas on nodes t_{j+1}, t_j, t_{j-1}, ... of a geometric
    // grid of ratio Q (in units of the step, origin t_j: 1, 0, -1/Q, -(1/Q + 1/Q^2), ...):
    // W[k][m] = m! * [theta^m] l_k(theta), l_k the Lagrange basis on the nodes.
    auto quad_weights = [](double Q, int K, double *W) {
        double x[8];
        x[0] = 1.0; x[1] = 0.0;
        { double acc = 0.0, f = 1.0; for (int k = 2; k < K; ++k) { f /= Q; acc -= f; x[k] = acc; } }
        for (int k = 0; k < K; ++k) {
            double co[8] = {1.0, 0, 0, 0, 0, 0, 0, 0};
            int deg = 0;
            double denom = 1.0;
            for (int j = 0; j < K; ++j) {
                if (j == k) continue;
                for (int m = deg + 1; m >= 1; --m) co[m] = co[m - 1] - x[j] * co[m];
                co[0] = -x[j] * co[0];
                ++deg;
                denom *= x[k] - x[j];
            }
            double fact = 1.0;
            for (int m = 0; m < K; ++m) { if (m > 1) fact *= (double)m; W[k * K + m] = fact * co[m] / denom; }
        }
    };
    std::vector<double> wtab((size_t)mp::kWtabSize, 0.0);
    for (int kind = 0; kind < mp::kKinds; ++kind) {
        mp::StrideK &K = s.sk[kind];
        const double lnQ = kind == 0 ? lnq / 8.0 : lnq * (double)(1 << (kind - 1));
        K.lnQ = lnQ;
        K.inv_Q = std::exp(-lnQ);
        K.one_m_invQ = -std::expm1(-lnQ);
        const int ns = kind == 0 ? 1 : (1 << (kind - 1));
        double *T = wtab.data() + (size_t)kind * mp::kWtabStride;
        for (int i = 0; i < 8; ++i)
            T[mp::kWtabTheta + i] = (i < ns && ns > 1) ? std::expm1(lnq * (double)i) / std::expm1(lnQ) : 0.0;
        double W5[25];
        quad_weights(std::exp(lnQ), 5, W5);
        for (int k = 0; k < 5; ++k)
            for (int m = 0; m < 5; ++m) T[6 * k + m] = W5[k * 5 + m];
        // dense output of Mdisc where the step is longer than tvisc (mp_device.h kWtabDense): cubic Lagrange weights at the
        // skipped grid points, the step being the first / middle / last interval of its four nodes (times in units of the
        // step, origin at its start; consecutive steps grow by Q)
        if (kind >= 2) {
            const double Q = std::exp(lnQ);
            const double nodes[3][4] = {{0.0, 1.0, 1.0 + Q, 1.0 + Q + Q * Q},
                                        {-1.0 / Q, 0.0, 1.0, 1.0 + Q},
                                        {-1.0 / Q - 1.0 / (Q * Q), -1.0 / Q, 0.0, 1.0}};
            for (int i = 1; i < ns; ++i) {
                const double th = T[mp::kWtabTheta + i];
                double *D = wtab.data() + mp::kWtabDense + ((kind - 2) * 7 + (i - 1)) * 12;
                for (int v = 0; v < 3; ++v)
                    for (int k = 0; k < 4; ++k) {
                        double l = 1.0;
                        for (int m = 0; m < 4; ++m)
                            if (m != k) l *= (th - nodes[v][m]) / (nodes[v][k] - nodes[v][m]);
                        D[v * 4 + k] = l;
                    }
            }
        }
    }
    // behind it: Q^k of the kinds 1 .. 4 (mp_device.h DevShared::ttab)
    const size_t ttab_off = wtab.size();
    wtab.resize(ttab_off + (size_t)(mp::kKinds - 1) * mp::kTtabN);
    for (int kind = 1; kind < mp::kKinds; ++kind)
        for (int k = 0; k < mp::kTtabN; ++k) wtab[ttab_off + (size_t)(kind - 1) * mp::kTtabN + k] = std::exp((double)k * s.sk[kind].lnQ);
    if (h->d_wtab.ensure(wtab.size()) != MP_OK ||
        hipMemcpy(h->d_wtab.p, wtab.data(), wtab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        fail(MP_EHIP, "mp_create: cannot upload the quadrature tables");
        mp_destroy(h);
        return nullptr;
    }
    s.wtab = h->d_wtab.p;
    s.ttab = h->d_wtab.p + ttab_off;
    if (rebuild_datasets(h) != MP_OK) {
        mp_destroy(h);
        return nullptr;
    }
    return h;
}

mp_handle *mp_create_multi(const mp_model_cfg *cfg, const double *tgrid, int n_grid, const int *devices, int n_devices) {
    if (!devices || n_devices < 1 || n_devices > 64) {
        fail(MP_EINVAL, "mp_create_multi: devices NULL or n_devices outside 1..64");
        return nullptr;
    }
    mp_handle *h = new mp_handle();
    for (int g = 0; g < n_devices; ++g) {
        mp_handle *s = mp_create(cfg, tgrid, n_grid, devices[g]);     // (validates cfg / tgrid; its message stays in mp_last_error)
        if (!s) {
            mp_destroy(h);
            return nullptr;
        }
        h->sub.push_back(s);
    }
    h->device = h->sub[0]->device;
    h->tgrid = h->sub[0]->tgrid;
    h->n_tiles = h->sub[0]->n_tiles;
    h->sh = h->sub[0]->sh;          // the scalar settings (policy, tolerances, n_simd of the first device); no device pointer of it is used
    return h;
}

int mp_n_devices(const mp_handle *h) { return h ? (h->sub.empty() ? 1 : (int)h->sub.size()) : 0; }

int mp_destroy(mp_handle *h) {
    if (!h) return MP_OK;
    if (!h->sub.empty()) {
        for (mp_handle *s : h->sub) (void)mp_destroy(s);
        delete h;
        return MP_OK;
    }
    DeviceScope scope(h->device);
    (void)hipDeviceSynchronize();
    h->d_wtab.release();
    h->d_tgrid.release(); h->d_obs_dx.release(); h->d_obs_idt.release(); h->d_obs_y.release();
    h->d_obs_yerr.release(); h->d_obs_g.release(); h->d_tile_ptr.release(); h->d_ds.release();
    h->w_pars.release(); h->w_lnprob.release(); h->w_curves.release();
    h->w_dsid.release(); h->w_status.release(); h->w_sweeps.release(); h->w_tile_log.release();
    h->h_io.release();
    for (int i = 0; i < mp_handle::kOrderRing; ++i) {
        h->order[i].release();
        if (h->order_done[i]) (void)hipEventDestroy(h->order_done[i]);
    }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MP_OK;
}

int mp_set_dataset(mp_handle *h, int ds_id, const double *x, const double *y, const double *yerr, int n_obs) {
    if (!h || !x || !y || !yerr) return fail(MP_EINVAL, "mp_set_dataset: NULL argument");
    if (ds_id < 0 || ds_id >= MP_MAX_DATASETS) return fail(MP_EINVAL, "mp_set_dataset: ds_id %d out of range", ds_id);
    if (n_obs <= 0) return fail(MP_EINVAL, "mp_set_dataset: n_obs must be positive");
    Lock lock(h->mu);
    if (!h->sub.empty()) {
        for (mp_handle *s : h->sub) { const int rc = mp_set_dataset(s, ds_id, x, y, yerr, n_obs); if (rc) return rc; }
        h->sh.n_ds = h->sub[0]->sh.n_ds;
        return MP_OK;
    }
    const std::vector<double> &t = h->tgrid;
    const int n = (int)t.size();
    for (int j = 0; j < n_obs; ++j) {
        if (!(x[j] >= t.front()) || !(x[j] <= t.back()))  // interp1d(bounds_error=True), magnetar/funcs.py:214-215
            return fail(MP_ERANGE, "A value in x_new is %s the interpolation range.",
                        (x[j] < t.front()) ? "below" : "above");
    }
    std::vector<int> order(n_obs);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return x[a] < x[b]; });
    HostDataset d;
    d.set = true;
    d.tile_ptr.assign((size_t)h->n_tiles + 1, 0);
    for (int k = 0; k < n_obs; ++k) {
        const int j = order[k];
        int g = (int)(std::upper_bound(t.begin(), t.end(), x[j]) - t.begin()) - 1;  // t[g] <= x < t[g+1]
        g = std::min(std::max(g, 0), n - 2);
        d.g.push_back(g);
        d.dx.push_back(x[j] - t[g]);
        d.idt.push_back(1.0 / (t[g + 1] - t[g]));
        d.y.push_back(y[j]);
        d.yerr.push_back(yerr[j]);
        d.tile_ptr[(size_t)(g / mp::kTile) + 1] += 1;
    }
    for (size_t k = 1; k < d.tile_ptr.size(); ++k) d.tile_ptr[k] += d.tile_ptr[k - 1];
    const bool replaced = h->ds[ds_id].set;
    h->ds[ds_id] = std::move(d);
    DeviceScope scope(h->device);
    return upload_dataset(h, ds_id, replaced);
}

int mp_set_prior(mp_handle *h, const double *lower, const double *upper, int ndim, uint32_t log_mask) {
    if (!h) return fail(MP_EINVAL, "mp_set_prior: NULL handle");
    if (ndim < 0 || ndim > MP_MAX_NDIM) return fail(MP_EINVAL, "mp_set_prior: ndim %d out of range", ndim);
    if (ndim > 0 && (!lower || !upper)) return fail(MP_EINVAL, "mp_set_prior: NULL bounds");
    Lock lock(h->mu);
    for (mp_handle *s : h->sub) { const int rc = mp_set_prior(s, lower, upper, ndim, log_mask); if (rc) return rc; }
    for (int i = 0; i < MP_MAX_NDIM; ++i) {
        h->sh.lower[i] = i < ndim ? lower[i] : -INFINITY;
        h->sh.upper[i] = i < ndim ? upper[i] : INFINITY;
    }
    h->sh.n_prior = ndim;
    h->sh.log_mask = log_mask;
    return MP_OK;
}

static int check_batch_args(const mp_handle *h, const void *pars, int n, int ndim, const void *lnprob) {
    if (!h || !pars || !lnprob) return fail(MP_EINVAL, "lnprob batch: NULL argument");
    if (n < 0) return fail(MP_EINVAL, "lnprob batch: negative n");
    if (ndim < 6 || ndim > MP_MAX_NDIM) return fail(MP_EINVAL, "lnprob batch: ndim must be 6..9, got %d", ndim);
    if (h->sh.n_ds <= 0) return fail(MP_ESTATE, "lnprob batch: no dataset registered (mp_set_dataset)");
    return MP_OK;
}

int mp_lnprob_batch_dev(mp_handle *h, const double *d_pars, const int32_t *d_ds_id, int n, int ndim,
                        double *d_lnprob, int32_t *d_status, double *d_ltot, void *stream) {
    int rc = check_batch_args(h, d_pars, n, ndim, d_lnprob);
    if (rc) return rc;
    if (!h->sub.empty()) return fail(MP_ESTATE, "mp_lnprob_batch_dev: device pointers belong to ONE device; a multi-device handle serves the host-buffer entries");
    Lock lock(h->mu);
    if (!d_ds_id && !h->ds[0].set) return fail(MP_ESTATE, "lnprob batch: ds_id is NULL but dataset 0 is not set");
    DeviceScope scope(h->device);
    mp::LaunchArgs a{};
    a.pars = d_pars;
    a.ds_id = d_ds_id;
    a.n = n;
    a.ndim = ndim;
    a.physical = 0;
    a.want_chi2 = 1;
    a.lnprob = d_lnprob;
    a.status = d_status;
    a.ltot = d_ltot;
    return launch_lnprob_ordered(h, a, (hipStream_t)stream);
}

// The host-buffer batch in two halves, so that a multi-device handle can have every device's launch in flight before it
// waits for the first: batch_begin stages the rows and enqueues the kernel on the handle's stream, batch_end waits and
// hands the results over.  (Caller holds h->mu and has validated the arguments.)
static int batch_begin(mp_handle *h, const double *pars, const int32_t *ds_id, int n, int ndim, double *ltot_out) {
    int rc;
    DeviceScope scope(h->device);
    const size_t ng = h->tgrid.size();
    // One page-locked staging block owned by the handle, mapped into the device's address space: [pars n*ndim f64 | ds_id n i32]
    // in, [lnprob n f64 | status n i32 | sweeps n i32 | tiles n i32] out.  The kernel reads a walker's 48 - 72 bytes and writes
    // its 20 bytes IN PLACE over PCIe (round 5): no copy command either way -- one launch and one wait per call where the
    // two asynchronous copies around the kernel cost as much as the kernel itself.
    const size_t in_pars = sizeof(double) * (size_t)n * ndim, in_ids = ds_id ? sizeof(int32_t) * (size_t)n : 0;
    const size_t in_bytes = (in_pars + in_ids + 7) & ~(size_t)7;
    const size_t out_bytes = (sizeof(double) + 4 * sizeof(int32_t)) * (size_t)n;   // lnprob | status | sweeps | tiles (+ pad)
    if ((rc = h->h_io.ensure(in_bytes + out_bytes)) || (ltot_out && (rc = h->w_curves.ensure((size_t)n * ng))))
        return rc;
    hipStream_t st = h->stream;
    std::memcpy(h->h_io.p, pars, in_pars);
    if (ds_id) std::memcpy(h->h_io.p + in_pars, ds_id, in_ids);
    unsigned char *d_out = h->h_io.dev + in_bytes;
    mp::LaunchArgs a{};
    a.pars = (const double *)h->h_io.dev;
    a.ds_id = ds_id ? (const int32_t *)(h->h_io.dev + in_pars) : nullptr;
    a.n = n;
    a.ndim = ndim;
    a.physical = 0;
    a.want_chi2 = 1;
    a.lnprob = (double *)d_out;
    a.status = (int32_t *)(d_out + sizeof(double) * (size_t)n);
    a.sweeps = a.status + n;
    a.tiles = a.sweeps + n;
    a.ltot = ltot_out ? h->w_curves.p : nullptr;   // rows of walkers that fail are NaN-filled by the kernel
    if (h->tile_log_on) {
        if ((rc = h->w_tile_log.ensure((size_t)n * MP_TILE_LOG))) return rc;
        HIP_TRY(hipMemsetAsync(h->w_tile_log.p, 0xFF, (size_t)n * MP_TILE_LOG * sizeof(int32_t), st));
        a.tile_log = h->w_tile_log.p;
    }
    if ((rc = launch_lnprob_ordered(h, a, st))) return rc;
    if (h->tile_log_on) {
        h->last_tile_log.resize((size_t)n * MP_TILE_LOG);
        HIP_TRY(hipMemcpyAsync(h->last_tile_log.data(), h->w_tile_log.p, (size_t)n * MP_TILE_LOG * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    } else h->last_tile_log.clear();
    if (ltot_out)
        HIP_TRY(hipMemcpyAsync(ltot_out, h->w_curves.p, sizeof(double) * (size_t)n * ng, hipMemcpyDeviceToHost, st));
    h->pend_n = n;
    h->pend_in_bytes = in_bytes;
    return MP_OK;
}

static int batch_end(mp_handle *h, double *lnprob_out, int32_t *status_out) {
    DeviceScope scope(h->device);
    const int n = h->pend_n;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const unsigned char *h_out = h->h_io.p + h->pend_in_bytes;
    std::memcpy(lnprob_out, h_out, sizeof(double) * (size_t)n);
    const int32_t *status = (const int32_t *)(h_out + sizeof(double) * (size_t)n), *sweeps = status + n, *tiles = sweeps + n;
    if (status_out) std::memcpy(status_out, status, sizeof(int32_t) * (size_t)n);
    double tot = 0.0, tot_tiles = 0.0;
    int cnt = 0;
    h->last_sweeps.assign(sweeps, sweeps + n);
    h->last_tiles.assign(tiles, tiles + n);
    for (int i = 0; i < n; ++i)
        if (status[i] == MP_STATUS_OK) { tot += sweeps[i]; tot_tiles += tiles[i]; ++cnt; }
    h->last_tot_sweeps = tot; h->last_tot_tiles = tot_tiles; h->last_cnt_ok = cnt;
    h->last_mean_sweeps = tot_tiles > 0.0 ? tot / tot_tiles : 0.0;
    h->last_mean_tiles = cnt ? tot_tiles / (double)cnt : 0.0;
    return MP_OK;
}

int mp_lnprob_batch(mp_handle *h, const double *pars, const int32_t *ds_id, int n, int ndim, double *lnprob_out,
                    int32_t *status_out, double *ltot_out) {
    int rc = check_batch_args(h, pars, n, ndim, lnprob_out);
    if (rc) return rc;
    if (n == 0) return MP_OK;
    Lock lock(h->mu);
    const mp_handle *hd = h->sub.empty() ? h : h->sub[0];     // (the datasets of a multi-device handle live in its evaluators)
    if (ds_id) {
        for (int i = 0; i < n; ++i)
            if (ds_id[i] < 0 || ds_id[i] >= MP_MAX_DATASETS || !hd->ds[ds_id[i]].set)
                return fail(MP_EINVAL, "lnprob batch: walker %d refers to unset dataset %d", i, ds_id[i]);
    } else if (!hd->ds[0].set) {
        return fail(MP_ESTATE, "lnprob batch: ds_id is NULL but dataset 0 is not set");
    }
    if (h->sub.empty()) {
        if ((rc = batch_begin(h, pars, ds_id, n, ndim, ltot_out))) return rc;
        return batch_end(h, lnprob_out, status_out);
    }
    // Multi-device: contiguous blocks of ceil(n / G) rows (SURVEY.md 8(e)'s partitioning), every device's kernel enqueued before
    // the first is waited for; one host thread drives them all (a launch returns in microseconds, the kernels run side by side).
    const int G = (int)h->sub.size(), per = (n + G - 1) / G;
    const size_t ng = h->tgrid.size();
    int used = 0;
    for (int g = 0; g < G; ++g) {
        const int lo = std::min(g * per, n), cnt = std::min(lo + per, n) - lo;
        if (cnt <= 0) break;
        mp_handle *s = h->sub[(size_t)g];
        Lock sl(s->mu);
        if ((rc = batch_begin(s, pars + (size_t)lo * ndim, ds_id ? ds_id + lo : nullptr, cnt, ndim, ltot_out ? ltot_out + (size_t)lo * ng : nullptr))) {
            for (int k = 0; k < used; ++k) (void)hipStreamSynchronize(h->sub[(size_t)k]->stream);   // nothing may still write the caller's buffers
            return rc;
        }
        ++used;
    }
    int first_rc = MP_OK;
    h->last_sweeps.clear();
    h->last_tiles.clear();
    double tot = 0.0, tot_tiles = 0.0;
    int cnt_ok = 0;
    for (int g = 0; g < used; ++g) {
        mp_handle *s = h->sub[(size_t)g];
        const int lo = g * per;
        Lock sl(s->mu);
        rc = batch_end(s, lnprob_out + lo, status_out ? status_out + lo : nullptr);
        if (rc && !first_rc) first_rc = rc;
        h->last_sweeps.insert(h->last_sweeps.end(), s->last_sweeps.begin(), s->last_sweeps.end());
        h->last_tiles.insert(h->last_tiles.end(), s->last_tiles.begin(), s->last_tiles.end());
        tot += s->last_tot_sweeps; tot_tiles += s->last_tot_tiles; cnt_ok += s->last_cnt_ok;
    }
    if (first_rc) return first_rc;
    h->last_mean_sweeps = tot_tiles > 0.0 ? tot / tot_tiles : 0.0;
    h->last_mean_tiles = cnt_ok ? tot_tiles / (double)cnt_ok : 0.0;
    return MP_OK;
}

int mp_model_lc(mp_handle *h, const double *pars, int ndim, double *out, double *traj, int32_t *status) {
    if (!h || !pars || !out) return fail(MP_EINVAL, "mp_model_lc: NULL argument");
    if (ndim < 6 || ndim > MP_MAX_NDIM) return fail(MP_EINVAL, "mp_model_lc: ndim must be 6..9, got %d", ndim);
    if (!h->sub.empty()) return mp_model_lc(h->sub[0], pars, ndim, out, traj, status);   // one walker: the first device
    Lock lock(h->mu);
    DeviceScope scope(h->device);
    const size_t ng = h->tgrid.size();
    int rc;
    if ((rc = h->w_pars.ensure(MP_MAX_NDIM)) || (rc = h->w_lnprob.ensure(1)) || (rc = h->w_status.ensure(1)) ||
        (rc = h->w_curves.ensure(5 * ng)))
        return rc;
    hipStream_t st = h->stream;
    HIP_TRY(hipMemcpyAsync(h->w_pars.p, pars, sizeof(double) * (size_t)ndim, hipMemcpyHostToDevice, st));
    mp::LaunchArgs a{};
    a.pars = h->w_pars.p;
    a.n = 1;
    a.ndim = ndim;
    a.physical = 1;
    a.want_chi2 = 0;
    a.lnprob = h->w_lnprob.p;
    a.status = h->w_status.p;
    a.ltot = h->w_curves.p;
    a.lprop = h->w_curves.p + ng;
    a.ldip = h->w_curves.p + 2 * ng;
    a.mdisc = h->w_curves.p + 3 * ng;
    a.omega = h->w_curves.p + 4 * ng;
    if ((rc = launch_lnprob_ordered(h, a, st))) return rc;
    int32_t stt = 0;
    std::memcpy(out, h->tgrid.data(), sizeof(double) * ng);
    HIP_TRY(hipMemcpyAsync(out + ng, h->w_curves.p, sizeof(double) * 3 * ng, hipMemcpyDeviceToHost, st));
    if (traj) HIP_TRY(hipMemcpyAsync(traj, h->w_curves.p + 3 * ng, sizeof(double) * 2 * ng, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&stt, h->w_status.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (status) *status = stt;
    return MP_OK;
}

int mp_rhs_batch(mp_handle *h, const double *pars, int ndim, const double *t, const double *y, int n, double *dydt,
                 double *lam) {
    if (!h || !pars || !t || !y || !dydt) return fail(MP_EINVAL, "mp_rhs_batch: NULL argument");
    if (ndim < 6 || ndim > MP_MAX_NDIM) return fail(MP_EINVAL, "mp_rhs_batch: ndim must be 6..9, got %d", ndim);
    if (n < 0) return fail(MP_EINVAL, "mp_rhs_batch: negative n");
    if (n == 0) return MP_OK;
    if (!h->sub.empty()) return mp_rhs_batch(h->sub[0], pars, ndim, t, y, n, dydt, lam);
    Lock lock(h->mu);
    DeviceScope scope(h->device);
    // one device block: [pars n*ndim | t n | y 2n] in, [dydt 2n | lam n] out
    const size_t n_in = (size_t)n * (ndim + 3), n_out = (size_t)n * 3;
    int rc;
    if ((rc = h->w_curves.ensure(n_in + n_out))) return rc;
    double *d = h->w_curves.p;
    hipStream_t st = h->stream;
    HIP_TRY(hipMemcpyAsync(d, pars, sizeof(double) * (size_t)n * ndim, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + (size_t)n * ndim, t, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + (size_t)n * (ndim + 1), y, sizeof(double) * 2 * (size_t)n, hipMemcpyHostToDevice, st));
    mp::RhsArgs r{};
    r.pars = d;
    r.t = d + (size_t)n * ndim;
    r.y = d + (size_t)n * (ndim + 1);
    r.dydt = d + n_in;
    r.lam = d + n_in + 2 * (size_t)n;
    r.n = n;
    r.ndim = ndim;
    const int e = mp::launch_rhs(h->sh, r, st);
    if (e) return fail(MP_EHIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    HIP_TRY(hipMemcpyAsync(dydt, r.dydt, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost, st));
    if (lam) HIP_TRY(hipMemcpyAsync(lam, r.lam, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return MP_OK;
}

int mp_synchronize(mp_handle *h) {
    if (!h) return fail(MP_EINVAL, "mp_synchronize: NULL handle");
    Lock lock(h->mu);
    if (!h->sub.empty()) {
        for (mp_handle *s : h->sub) { const int rc = mp_synchronize(s); if (rc) return rc; }
        return MP_OK;
    }
    DeviceScope scope(h->device);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MP_OK;
}

// ---------------------------------------------------------------- ensemble sampler (stretch move)
// Philox4x32-10 (same function as in mp_kernels.hip); the host uses it for the random red/blue split.
static void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct mp_sampler {
    mp_handle *h = nullptr;
    int n_walkers = 0, n_ensembles = 0, n_total = 0, ndim = 0, target = 0;
    uint64_t seed = 0;
    double a = 2.0;
    uint64_t steps_done = 0;
    bool have_state = false;
    DevBuf<double> d_pos, d_lnprob, d_chain, d_chain_lnp, d_bad, d_spec;
    std::vector<int32_t> ens_ds;   // dataset of every ensemble
    int whole_step = 1;   // mp_sampler_run: one launch per step where the ensemble is small enough (mp_sampler_set_whole_step)
    DevBuf<int64_t> d_acc;
    DevBuf<int32_t> d_perm, d_dsid, d_status;
    DevBuf<uint32_t> d_bad_count;
    PinnedBuf h_perm;   // page-locked staging of the random splits: their upload overlaps the running half-steps
    // walker-sharded driving (mp_sampler_halfstep_shard / _apply): splits of kWin steps at a time, double-buffered
    static constexpr int kWin = 32;
    DevBuf<int32_t> d_win[2];
    PinnedBuf h_win[2];
    hipEvent_t win_copied[2] = {nullptr, nullptr};
    int64_t win_id[2] = {-1, -1};
    bool ext_stream_work = false;   // half-steps were enqueued on a caller's stream since the last device-wide wait
    // failed proposals (the reference's fbad file): the device window d_bad is drained into this log
    std::vector<double> bad_log;    // [rows][ndim]
    int64_t n_bad = 0;              // exact count since creation (rows beyond the window between two drains are counted, not kept)
};

// Move the device window of failed proposals into the host log and reset it.  The caller has made sure that no kernel
// of this sampler is in flight.
static int drain_bad(mp_sampler *s) {
    uint32_t cnt = 0;
    HIP_TRY(hipMemcpy(&cnt, s->d_bad_count.p, sizeof cnt, hipMemcpyDeviceToHost));
    if (cnt == 0) return MP_OK;
    const size_t cap = s->d_bad.cap / (size_t)s->ndim, rows = std::min<size_t>(cnt, cap);
    const size_t old = s->bad_log.size();
    s->bad_log.resize(old + rows * (size_t)s->ndim);
    HIP_TRY(hipMemcpy(s->bad_log.data() + old, s->d_bad.p, rows * s->ndim * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(s->d_bad_count.p, 0, sizeof(uint32_t)));
    s->n_bad += (int64_t)cnt;
    return MP_OK;
}

// random split of every ensemble for step `step` (emcee's randomize_split): Fisher-Yates, counter (step, ensemble, i, 'split')
static void draw_split(const mp_sampler *s, uint64_t step64, int32_t *perm) {
    const uint32_t step = (uint32_t)step64;
    for (int e = 0; e < s->n_ensembles; ++e) {
        int32_t *p = perm + (size_t)e * s->n_walkers;
        std::iota(p, p + s->n_walkers, 0);
        for (int i = s->n_walkers - 1; i > 0; --i) {
            uint32_t r[4];
            philox4x32_10((uint32_t)s->seed, (uint32_t)(s->seed >> 32), step, (uint32_t)e, (uint32_t)i, 0x5117u, r);
            const uint64_t r64 = ((uint64_t)r[0] << 32) | r[1];
            std::swap(p[i], p[(size_t)(r64 % (uint64_t)(i + 1))]);
        }
    }
}

// splits of `count` consecutive steps into perm[count][n_total]; the steps are independent (counter-based generator), so
// large ensembles are drawn by a few host threads (8 192 walkers: 0.4 ms per step on one core, more than a half-step of a
// walker-sharded ensemble takes on the GPU)
static void draw_splits(const mp_sampler *s, uint64_t step0, int count, int32_t *perm) {
    const size_t nt = (size_t)s->n_total;
    const int n_thr = (int)std::min<size_t>({(size_t)count, (size_t)4, (nt * (size_t)count) / 8192});
    auto work = [&](int first, int stride) {
        for (int i = first; i < count; i += stride) draw_split(s, step0 + (uint64_t)i, perm + (size_t)i * nt);
    };
    if (n_thr <= 1) { work(0, 1); return; }
    std::vector<std::thread> pool;
    for (int k = 1; k < n_thr; ++k) pool.emplace_back(work, k, n_thr);
    work(0, n_thr);
    for (auto &th : pool) th.join();
}

static mp::StretchArgs stretch_args(const mp_sampler *s, const int32_t *d_perm, uint64_t step, int half) {
    mp::StretchArgs g{};
    g.pos = s->d_pos.p; g.lnprob = s->d_lnprob.p; g.n_accepted = s->d_acc.p;
    g.perm = d_perm;
    g.ds_id = s->d_dsid.p;
    g.n_walkers = s->n_walkers; g.n_half = s->n_walkers / 2; g.n_ensembles = s->n_ensembles;
    g.n_total = s->n_total; g.ndim = s->ndim; g.half = half; g.target = s->target;
    g.step = (uint32_t)step; g.seed = s->seed; g.a = s->a;
    g.bad_log = s->d_bad.p; g.bad_count = s->d_bad_count.p; g.bad_cap = (uint32_t)(s->d_bad.cap / (size_t)std::max(s->ndim, 1));
    // Ensembles on light curves of different lengths (BASELINE config 5: 50 / 410 / 8 / 1 944 points): the half-step launch
    // starts the ensemble with the longest light curve first, so that its waves do not begin last and finish alone.  The order
    // is a function of the datasets only, so every rank of a walker-sharded run derives the same one.
    if (s->target == 0 && s->n_ensembles > 1 && s->n_ensembles <= 16) {
        int idx[16];
        std::iota(idx, idx + s->n_ensembles, 0);
        std::stable_sort(idx, idx + s->n_ensembles, [&](int x, int y) {
            return s->h->ds[s->ens_ds[(size_t)x]].g.size() > s->h->ds[s->ens_ds[(size_t)y]].g.size();
        });
        bool identity = true;
        for (int e = 0; e < s->n_ensembles; ++e) identity = identity && idx[e] == e;
        if (!identity)
            for (int e = 0; e < s->n_ensembles; ++e) g.ens_order |= (uint64_t)idx[e] << (4 * e);
    }
    return g;
}

mp_sampler *mp_sampler_create(mp_handle *h, int n_walkers, int n_ensembles, int ndim, const int32_t *ens_ds_id,
                              uint64_t seed, double a, int target) {
    if (!h) { fail(MP_EINVAL, "mp_sampler_create: NULL handle"); return nullptr; }
    if (!h->sub.empty()) { fail(MP_ESTATE, "mp_sampler_create: the device-resident sampler lives on ONE device (walker sharding across devices: magprop_amd/distributed.py)"); return nullptr; }
    if (n_walkers < 2 || (n_walkers & 1)) { fail(MP_EINVAL, "mp_sampler_create: n_walkers must be even and >= 2"); return nullptr; }
    if (n_ensembles < 1 || ndim < 1 || ndim > MP_MAX_NDIM || (target == 0 && ndim < 6)) {
        fail(MP_EINVAL, "mp_sampler_create: bad n_ensembles / ndim");
        return nullptr;
    }
    if (!(a > 1.0)) { fail(MP_EINVAL, "mp_sampler_create: stretch scale a must exceed 1"); return nullptr; }
    if (target == 0 && h->sh.cfg.dipole_torque != 0) { fail(MP_ESTATE, "mp_sampler_create: the alternative dipole torque (cfg.dipole_torque = 1) is served by the curve kernels only"); return nullptr; }
    Lock lock(h->mu);
    if (target == 0) {
        for (int e = 0; e < n_ensembles; ++e) {
            const int d = ens_ds_id ? ens_ds_id[e] : 0;
            if (d < 0 || d >= MP_MAX_DATASETS || !h->ds[d].set) {
                fail(MP_ESTATE, "mp_sampler_create: ensemble %d refers to unset dataset %d", e, d);
                return nullptr;
            }
        }
    }
    mp_sampler *s = new mp_sampler();
    s->h = h; s->n_walkers = n_walkers; s->n_ensembles = n_ensembles; s->n_total = n_walkers * n_ensembles;
    s->ndim = ndim; s->target = target; s->seed = seed; s->a = a;
    DeviceScope scope(h->device);
    const size_t nt = (size_t)s->n_total;
    std::vector<int32_t> ds(nt, 0);
    s->ens_ds.resize((size_t)n_ensembles);
    for (int e = 0; e < n_ensembles; ++e) s->ens_ds[(size_t)e] = ens_ds_id ? ens_ds_id[e] : 0;
    for (int e = 0; e < n_ensembles; ++e)
        for (int k = 0; k < n_walkers; ++k) ds[(size_t)e * n_walkers + k] = ens_ds_id ? ens_ds_id[e] : 0;
    constexpr size_t kBadRows = MP_BAD_WINDOW;   // device window of failed proposals between two drains (drain_bad)
    if (s->d_pos.ensure(nt * ndim) || s->d_lnprob.ensure(nt) || s->d_acc.ensure(nt) || s->d_dsid.ensure(nt) ||
        s->d_status.ensure(nt) || s->d_bad.ensure(kBadRows * ndim) || s->d_bad_count.ensure(1) ||
        hipMemcpy(s->d_dsid.p, ds.data(), nt * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(s->d_acc.p, 0, nt * sizeof(int64_t)) != hipSuccess ||
        hipMemset(s->d_bad_count.p, 0, sizeof(uint32_t)) != hipSuccess ||
        hipEventCreateWithFlags(&s->win_copied[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->win_copied[1], hipEventDisableTiming) != hipSuccess) {
        fail(MP_EHIP, "mp_sampler_create: device allocation failed");
        mp_sampler_destroy(s);
        return nullptr;
    }
    return s;
}

int mp_sampler_destroy(mp_sampler *s) {
    if (!s) return MP_OK;
    Lock lock(s->h->mu);
    DeviceScope scope(s->h->device);
    (void)hipDeviceSynchronize();
    s->d_pos.release(); s->d_lnprob.release(); s->d_chain.release(); s->d_chain_lnp.release();
    s->d_acc.release(); s->d_perm.release(); s->d_dsid.release(); s->d_status.release(); s->h_perm.release();
    s->d_bad.release(); s->d_bad_count.release();
    for (int b = 0; b < 2; ++b) {
        s->d_win[b].release(); s->h_win[b].release();
        if (s->win_copied[b]) (void)hipEventDestroy(s->win_copied[b]);
    }
    delete s;
    return MP_OK;
}

int mp_sampler_set_positions(mp_sampler *s, const double *pos) {
    if (!s || !pos) return fail(MP_EINVAL, "mp_sampler_set_positions: NULL argument");
    mp_handle *h = s->h;
    Lock lock(h->mu);
    DeviceScope scope(h->device);
    const size_t nt = (size_t)s->n_total;
    for (size_t i = 0; i < nt * s->ndim; ++i)
        if (!std::isfinite(pos[i])) return fail(MP_EINVAL, "mp_sampler_set_positions: non-finite coordinate");
    HIP_TRY(hipDeviceSynchronize());   // the sharded entry points may have work in flight on a caller's stream
    s->ext_stream_work = false;
    HIP_TRY(hipMemcpyAsync(s->d_pos.p, pos, nt * s->ndim * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (s->target == 1) {
        std::vector<double> lp(nt, 0.0);
        for (size_t k = 0; k < nt; ++k)
            for (int i = 0; i < s->ndim; ++i) lp[k] -= 0.5 * pos[k * s->ndim + i] * pos[k * s->ndim + i];
        HIP_TRY(hipMemcpyAsync(s->d_lnprob.p, lp.data(), nt * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    } else {
        mp::LaunchArgs a{};
        a.pars = s->d_pos.p; a.ds_id = s->d_dsid.p; a.n = s->n_total; a.ndim = s->ndim; a.want_chi2 = 1;
        a.lnprob = s->d_lnprob.p; a.status = s->d_status.p;
        const int rc = launch_lnprob_ordered(h, a, h->stream);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    s->have_state = true;
    return MP_OK;
}

int mp_sampler_run(mp_sampler *s, int n_steps, double *chain, double *chain_lnprob) {
    if (!s || n_steps < 0) return fail(MP_EINVAL, "mp_sampler_run: bad argument");
    if (!s->have_state) return fail(MP_ESTATE, "mp_sampler_run: call mp_sampler_set_positions first");
    if ((chain == nullptr) != (chain_lnprob == nullptr)) return fail(MP_EINVAL, "mp_sampler_run: chain and chain_lnprob go together");
    mp_handle *h = s->h;
    Lock lock(h->mu);
    DeviceScope scope(h->device);
    const size_t nt = (size_t)s->n_total, row = nt * s->ndim;
    // chunks of steps so that the device-resident chain slab stays below ~256 MB, the splits below ~64 MB, and the
    // window of failed proposals (drained after every chunk) overflows only if more than 1 in 32 proposals fails
    const size_t perm_cap = std::max<size_t>(1, std::min<size_t>((size_t)16 << 20, (size_t)32 * MP_BAD_WINDOW) / nt);
    const int chunk_max = (int)std::min<size_t>(
        (size_t)std::max(n_steps, 1),
        chain ? std::max<size_t>(1, std::min<size_t>(perm_cap, (256u << 20) / (row * sizeof(double)))) : perm_cap);
    constexpr int kSub = 8;   // steps per batch of splits: the host draws the next batch while the GPU runs this one
    int rc;
    // A whole step per launch (mp_kernels.hip stretch_step_kernel: 3 n/2 evaluations, a third of them speculative) while
    // that beats two half-step launches (mp_device.h stretch_whole_step_fits); larger ensembles fill the device with one half-step
    // at a time.
    const int n_slots = (s->n_walkers / 2) * s->n_ensembles;
    const bool whole = s->whole_step && mp::stretch_whole_step_fits(h->sh, 3 * (long long)n_slots);
    if (whole && (rc = s->d_spec.ensure((size_t)3 * n_slots * (size_t)(s->ndim + mp::kSpecExtra)))) return rc;
    if (s->ext_stream_work) {   // sharded half-steps on a caller's stream may still be updating the state
        HIP_TRY(hipDeviceSynchronize());
        s->ext_stream_work = false;
    }
    for (int done = 0; done < n_steps;) {
        const int chunk = std::min(chunk_max, n_steps - done);
        if ((rc = s->h_perm.ensure((size_t)chunk * nt * sizeof(int32_t))) || (rc = s->d_perm.ensure((size_t)chunk * nt))) return rc;
        int32_t *perm = (int32_t *)s->h_perm.p;
        if (chain) {
            if ((rc = s->d_chain.ensure((size_t)chunk * row)) || (rc = s->d_chain_lnp.ensure((size_t)chunk * nt))) return rc;
        }
        for (int sub = 0; sub < chunk; sub += kSub) {
            const int sub_end = std::min(chunk, sub + kSub);
            draw_splits(s, s->steps_done + (uint64_t)sub, sub_end - sub, perm + (size_t)sub * nt);
            HIP_TRY(hipMemcpyAsync(s->d_perm.p + (size_t)sub * nt, perm + (size_t)sub * nt,
                                   (size_t)(sub_end - sub) * nt * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
            for (int st = sub; st < sub_end; ++st) {
                if (whole) {
                    mp::StretchArgs g = stretch_args(s, s->d_perm.p + (size_t)st * nt, s->steps_done + (uint64_t)st, 0);
                    g.chain = chain ? s->d_chain.p : nullptr;
                    g.chain_lnp = chain ? s->d_chain_lnp.p : nullptr;
                    g.chain_row = st;
                    g.spec = s->d_spec.p;
                    int e = mp::launch_stretch_step(h->sh, g, 3 * n_slots, h->stream);
                    if (!e) e = mp::launch_stretch_step_commit(g, h->stream);
                    if (e) return fail(MP_EHIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
                    continue;
                }
                for (int half = 0; half < 2; ++half) {
                    mp::StretchArgs g = stretch_args(s, s->d_perm.p + (size_t)st * nt, s->steps_done + (uint64_t)st, half);
                    g.chain = chain ? s->d_chain.p : nullptr;
                    g.chain_lnp = chain ? s->d_chain_lnp.p : nullptr;
                    g.chain_row = st;
                    const int e = mp::launch_stretch(h->sh, g, g.n_half * g.n_ensembles, h->stream);
                    if (e) return fail(MP_EHIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
                }
            }
        }
        if (chain) {
            HIP_TRY(hipMemcpyAsync(chain + (size_t)done * row, s->d_chain.p, (size_t)chunk * row * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipMemcpyAsync(chain_lnprob + (size_t)done * nt, s->d_chain_lnp.p, (size_t)chunk * nt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        }
        HIP_TRY(hipStreamSynchronize(h->stream));
        if ((rc = drain_bad(s))) return rc;
        s->steps_done += (uint64_t)chunk;
        done += chunk;
    }
    return MP_OK;
}

int mp_sampler_set_whole_step(mp_sampler *s, int enable) {
    if (!s) return fail(MP_EINVAL, "mp_sampler_set_whole_step: NULL sampler");
    Lock lock(s->h->mu);
    s->whole_step = enable != 0;
    return MP_OK;
}

// ---- walker-sharded driving: the caller (one process per GPU) runs, per half-step,
//        mp_sampler_halfstep_shard(its block of slots) -> all-gather of the outcome rows -> mp_sampler_halfstep_apply.
// Device pointer of the split of the current step; uploads the next window of kWin splits when the step enters it.
static int current_split(mp_sampler *s, hipStream_t st, const int32_t **d_perm) {
    const size_t nt = (size_t)s->n_total;
    const int64_t win = (int64_t)(s->steps_done / mp_sampler::kWin);
    const int b = (int)(win & 1);
    if (s->win_id[b] != win) {
        int rc;
        const size_t bytes = (size_t)mp_sampler::kWin * nt * sizeof(int32_t);
        if ((rc = s->h_win[b].ensure(bytes)) || (rc = s->d_win[b].ensure((size_t)mp_sampler::kWin * nt))) return rc;
        if (s->win_id[b] >= 0) HIP_TRY(hipEventSynchronize(s->win_copied[b]));   // staged two windows ago: long done
        int32_t *perm = (int32_t *)s->h_win[b].p;
        draw_splits(s, (uint64_t)win * mp_sampler::kWin, mp_sampler::kWin, perm);
        // stream order puts the copy behind every kernel that still reads this buffer's previous contents
        HIP_TRY(hipMemcpyAsync(s->d_win[b].p, perm, bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipEventRecord(s->win_copied[b], st));
        s->win_id[b] = win;
    }
    *d_perm = s->d_win[b].p + (size_t)(s->steps_done % mp_sampler::kWin) * nt;
    return MP_OK;
}

int mp_sampler_row_doubles(const mp_sampler *s) { return s ? s->ndim + 3 : 0; }
int mp_sampler_n_slots(const mp_sampler *s) { return s ? (s->n_walkers / 2) * s->n_ensembles : 0; }

int mp_sampler_halfstep_shard(mp_sampler *s, int half, int slot_lo, int slot_hi, double *d_rows, void *stream) {
    if (!s || (half != 0 && half != 1)) return fail(MP_EINVAL, "mp_sampler_halfstep_shard: bad argument");
    if (!s->have_state) return fail(MP_ESTATE, "mp_sampler_halfstep_shard: call mp_sampler_set_positions first");
    const int n_slots = (s->n_walkers / 2) * s->n_ensembles;
    if (slot_lo < 0 || slot_hi > n_slots || slot_lo > slot_hi) return fail(MP_EINVAL, "mp_sampler_halfstep_shard: slots [%d, %d) outside [0, %d)", slot_lo, slot_hi, n_slots);
    if (slot_hi > slot_lo && !d_rows) return fail(MP_EINVAL, "mp_sampler_halfstep_shard: NULL row buffer");
    mp_handle *h = s->h;
    Lock lock(h->mu);
    DeviceScope scope(h->device);
    hipStream_t st = (hipStream_t)stream;
    const int32_t *d_perm = nullptr;
    int rc;
    if ((rc = current_split(s, st, &d_perm))) return rc;
    s->ext_stream_work = true;
    if (slot_hi == slot_lo) return MP_OK;
    mp::StretchArgs g = stretch_args(s, d_perm, s->steps_done, half);
    g.upd = d_rows;
    g.slot_lo = slot_lo;
    const int e = mp::launch_stretch(h->sh, g, slot_hi - slot_lo, stream);
    if (e) return fail(MP_EHIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return MP_OK;
}

int mp_sampler_halfstep_apply(mp_sampler *s, int half, const double *d_rows, double *d_chain_row, double *d_chain_lnp_row,
                              void *stream) {
    if (!s || !d_rows || (half != 0 && half != 1)) return fail(MP_EINVAL, "mp_sampler_halfstep_apply: bad argument");
    if ((d_chain_row == nullptr) != (d_chain_lnp_row == nullptr)) return fail(MP_EINVAL, "mp_sampler_halfstep_apply: chain row and lnprob row go together");
    if (!s->have_state) return fail(MP_ESTATE, "mp_sampler_halfstep_apply: call mp_sampler_set_positions first");
    mp_handle *h = s->h;
    Lock lock(h->mu);
    DeviceScope scope(h->device);
    const int32_t *d_perm = nullptr;
    int rc;
    if ((rc = current_split(s, (hipStream_t)stream, &d_perm))) return rc;
    s->ext_stream_work = true;
    mp::StretchArgs g = stretch_args(s, d_perm, s->steps_done, half);
    g.upd = const_cast<double *>(d_rows);
    g.chain = d_chain_row;
    g.chain_lnp = d_chain_lnp_row;
    g.chain_row = 0;
    const int e = mp::launch_stretch_apply(g, stream);
    if (e) return fail(MP_EHIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    if (half == 1) s->steps_done += 1;
    return MP_OK;
}

// ---- the same for a whole step per launch (mp_kernels.hip stretch_step_kernel): the caller runs, per STEP,
//        mp_sampler_step_shard(its share of the 3 * n_slots blocks) -> ONE all-gather of the rows -> mp_sampler_step_apply.
int mp_sampler_step_blocks(const mp_sampler *s) { return s ? 3 * (s->n_walkers / 2) * s->n_ensembles : 0; }
int mp_sampler_step_row_doubles(const mp_sampler *s) { return s ? s->ndim + mp::kSpecExtra : 0; }

int mp_sampler_step_shard(mp_sampler *s, int block_lo, int block_hi, double *d_rows, void *stream) {
    if (!s) return fail(MP_EINVAL, "mp_sampler_step_shard: NULL sampler");
    if (!s->have_state) return fail(MP_ESTATE, "mp_sampler_step_shard: call mp_sampler_set_positions first");
    const int n_blocks = 3 * (s->n_walkers / 2) * s->n_ensembles;
    if (block_lo < 0 || block_hi > n_blocks || block_lo > block_hi) return fail(MP_EINVAL, "mp_sampler_step_shard: blocks [%d, %d) outside [0, %d)", block_lo, block_hi, n_blocks);
    if (block_hi > block_lo && !d_rows) return fail(MP_EINVAL, "mp_sampler_step_shard: NULL row buffer");
    mp_handle *h = s->h;
    Lock lock(h->mu);
    DeviceScope scope(h->device);
    hipStream_t st = (hipStream_t)stream;
    const int32_t *d_perm = nullptr;
    int rc;
    if ((rc = current_split(s, st, &d_perm))) return rc;
    s->ext_stream_work = true;
    if (block_hi == block_lo) return MP_OK;
    mp::StretchArgs g = stretch_args(s, d_perm, s->steps_done, 0);
    g.spec = d_rows;
    g.slot_lo = block_lo;
    const int e = mp::launch_stretch_step(h->sh, g, block_hi - block_lo, stream);
    if (e) return fail(MP_EHIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return MP_OK;
}

int mp_sampler_step_apply(mp_sampler *s, const double *d_rows, double *d_chain_row, double *d_chain_lnp_row, void *stream) {
    if (!s || !d_rows) return fail(MP_EINVAL, "mp_sampler_step_apply: bad argument");
    if ((d_chain_row == nullptr) != (d_chain_lnp_row == nullptr)) return fail(MP_EINVAL, "mp_sampler_step_apply: chain row and lnprob row go together");
    if (!s->have_state) return fail(MP_ESTATE, "mp_sampler_step_apply: call mp_sampler_set_positions first");
    mp_handle *h = s->h;
    Lock lock(h->mu);
    DeviceScope scope(h->device);
    const int32_t *d_perm = nullptr;
    int rc;
    if ((rc = current_split(s, (hipStream_t)stream, &d_perm))) return rc;
    s->ext_stream_work = true;
    mp::StretchArgs g = stretch_args(s, d_perm, s->steps_done, 0);
    g.spec = const_cast<double *>(d_rows);
    g.chain = d_chain_row;
    g.chain_lnp = d_chain_lnp_row;
    g.chain_row = 0;
    const int e = mp::launch_stretch_step_commit(g, stream);
    if (e) return fail(MP_EHIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    s->steps_done += 1;
    return MP_OK;
}

int mp_sampler_state_ptrs(mp_sampler *s, double **d_pos, double **d_lnprob) {
    if (!s) return fail(MP_EINVAL, "mp_sampler_state_ptrs: NULL sampler");
    if (d_pos) *d_pos = s->d_pos.p;
    if (d_lnprob) *d_lnprob = s->d_lnprob.p;
    return MP_OK;
}

int mp_sampler_get_bad(mp_sampler *s, int64_t first_row, double *pars, int max_rows, int64_t *n_bad, int64_t *n_logged) {
    if (!s || max_rows < 0 || first_row < 0 || (max_rows > 0 && !pars)) return fail(MP_EINVAL, "mp_sampler_get_bad: bad argument");
    mp_handle *h = s->h;
    Lock lock(h->mu);
    DeviceScope scope(h->device);
    HIP_TRY(hipDeviceSynchronize());
    const int rc = drain_bad(s);
    if (rc) return rc;
    const int64_t logged = (int64_t)(s->bad_log.size() / (size_t)s->ndim);
    if (n_bad) *n_bad = s->n_bad;
    if (n_logged) *n_logged = logged;
    const int64_t rows = std::max<int64_t>(0, std::min<int64_t>(logged - first_row, (int64_t)max_rows));
    if (rows) std::memcpy(pars, s->bad_log.data() + (size_t)first_row * s->ndim, (size_t)rows * s->ndim * sizeof(double));
    return (int)rows;
}

int mp_sampler_get_state(mp_sampler *s, double *pos, double *lnprob, int64_t *n_accepted, int64_t *steps_done) {
    if (!s) return fail(MP_EINVAL, "mp_sampler_get_state: NULL sampler");
    mp_handle *h = s->h;
    Lock lock(h->mu);
    DeviceScope scope(h->device);
    const size_t nt = (size_t)s->n_total;
    HIP_TRY(hipDeviceSynchronize());
    if (pos) HIP_TRY(hipMemcpy(pos, s->d_pos.p, nt * s->ndim * sizeof(double), hipMemcpyDeviceToHost));
    if (lnprob) HIP_TRY(hipMemcpy(lnprob, s->d_lnprob.p, nt * sizeof(double), hipMemcpyDeviceToHost));
    if (n_accepted) HIP_TRY(hipMemcpy(n_accepted, s->d_acc.p, nt * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (steps_done) *steps_done = (int64_t)s->steps_done;
    return MP_OK;
}

int mp_device(const mp_handle *h) { return h ? h->device : -1; }
void *mp_stream(const mp_handle *h) { return h ? (void *)h->stream : nullptr; }   // (a multi-device handle has none: NULL)
int mp_n_grid(const mp_handle *h) { return h ? (int)h->tgrid.size() : 0; }
double mp_last_mean_sweeps(const mp_handle *h) { return h ? h->last_mean_sweeps : 0.0; }
double mp_last_mean_tiles(const mp_handle *h) { return h ? h->last_mean_tiles : 0.0; }
int mp_last_sweeps(const mp_handle *h, int32_t *out, int n) {
    if (!h || !out || n < 0) return fail(MP_EINVAL, "mp_last_sweeps: bad argument");
    const int m = std::min<int>(n, (int)h->last_sweeps.size());
    std::copy(h->last_sweeps.begin(), h->last_sweeps.begin() + m, out);
    return m;
}
int mp_tile_log(mp_handle *h, int enable) {
    if (!h) return fail(MP_EINVAL, "mp_tile_log: NULL handle");
    Lock lock(h->mu);
    for (mp_handle *s : h->sub) (void)mp_tile_log(s, enable);
    h->tile_log_on = enable != 0;
    return MP_OK;
}
int mp_last_tile_log(const mp_handle *h, int walker, int32_t *out, int n) {
    if (!h || !out || n < 0 || walker < 0) return fail(MP_EINVAL, "mp_last_tile_log: bad argument");
    if (!h->sub.empty()) {   // the walker's block and its row inside it (contiguous blocks of ceil(n / G) rows)
        const int G = (int)h->sub.size(), total = (int)h->last_tiles.size(), per = (total + G - 1) / std::max(G, 1);
        if (per <= 0 || walker >= total) return 0;
        return mp_last_tile_log(h->sub[(size_t)(walker / per)], walker % per, out, n);
    }
    const size_t off = (size_t)walker * MP_TILE_LOG;
    if (off + MP_TILE_LOG > h->last_tile_log.size()) return 0;
    int m = 0;
    while (m < n && m < MP_TILE_LOG && h->last_tile_log[off + m] != -1) { out[m] = h->last_tile_log[off + m]; ++m; }
    return m;
}
int mp_last_tiles(const mp_handle *h, int32_t *out, int n) {
    if (!h || !out || n < 0) return fail(MP_EINVAL, "mp_last_tiles: bad argument");
    const int m = std::min<int>(n, (int)h->last_tiles.size());
    std::copy(h->last_tiles.begin(), h->last_tiles.begin() + m, out);
    return m;
}
double mp_sweep_tol(const mp_handle *h) { return h ? h->sh.sweep_tol : 0.0; }
int mp_get_policy(const mp_handle *h, double *out, int n) {
    if (!h || !out || n < 0) return fail(MP_EINVAL, "mp_get_policy: bad argument");
    const mp::DevShared &s = h->sh;
    double v[MP_POLICY_COUNT];
    v[MP_POLICY_MAX_STRIDE] = (double)(s.max_kind <= 1 ? 1 : (1 << (s.max_kind - 1)));
    v[MP_POLICY_STRIDE_TOL] = s.stride_tol;
    v[MP_POLICY_SWEEP_TOL] = s.sweep_tol;
    v[MP_POLICY_EARLY_HOLD_SECONDS] = s.early_hold_t;
    v[MP_POLICY_K4_TOL_FACTOR] = s.k4_tol_factor;
    v[MP_POLICY_COARSE_TOL_FACTOR] = s.coarse_tol_factor;
    v[MP_POLICY_COARSE_MAX_SWEEPS] = (double)s.coarse_max_sweeps;
    v[MP_POLICY_FINE_MAX_SWEEPS] = (double)s.fine_max_sweeps;
    v[MP_POLICY_TROUBLE_LIMIT] = (double)s.trouble_limit;
    v[MP_POLICY_STOP_FACTOR] = s.stop_factor;
    v[MP_POLICY_FORCED_STEPS_PER_LANE] = (double)s.force_spl;
#ifdef MP_EXPERIMENTS
    v[MP_POLICY_EXPERIMENTS] = 1.0;
#else
    v[MP_POLICY_EXPERIMENTS] = (MP_POLICY_MACROS_MODIFIED) ? 1.0 : 0.0;   // a -D build of the policy macros is an experiments build too
#endif
    v[MP_POLICY_LIGHT_TOL] = MP_LIGHT_TOL;
    v[MP_POLICY_CUT_BY_RATIO] = (double)MP_CUT_BY_RATIO;
    v[MP_POLICY_ABORT_SKIP_RATIO] = MP_ABORT_SKIP_RATIO;
    v[MP_POLICY_LOGPRED_MIN_KIND] = (double)MP_LOGPRED_MIN_KIND;
    v[MP_POLICY_PRE_EARLY_END_FACTOR] = MP_PRE_EARLY_END_FACTOR;
    const int m = std::min(n, (int)MP_POLICY_COUNT);
    std::copy(v, v + m, out);
    return m;
}
int mp_n_simd(const mp_handle *h) { return h ? h->sh.n_simd : 0; }

}  // extern "C"
