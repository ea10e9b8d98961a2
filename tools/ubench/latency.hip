// Micro-benchmark (gfx950): DEPENDENT-chain cost of the instruction patterns the lnprob kernels' critical path is made of, one
// wave per SIMD (and the LDS exchange / barrier patterns of the team kernels with 2 and 4 wavefronts per workgroup).
// Build: hipcc --offload-arch=gfx950 -O3 latency.hip -o latency ; run: ./latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

__device__ __forceinline__ double bcast0(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 0), hi = __builtin_amdgcn_readlane(__double2hiint(v), 0);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double uni(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_shr1(double keep, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(src), 0x111, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(src), 0x111, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// 16 dependent units per REP16; each unit's instruction count is given to the host for the per-unit / per-instruction figures
template <int KIND>
__global__ void bench(double *out, int iters, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) double lds[4 * 64 * 4 + 64];
    double a = threadIdx.x * 1e-3 + 1.0, a1 = a + 0.5;
    const double b = 0.999999, c = 1e-9;
    float f = (float)a;
    int idx = (threadIdx.x * 8) & 2047;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    lds[threadIdx.x] = 0.0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        else if (KIND == 1) { REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(a1) : "v"(b), "v"(c));) }
        else if (KIND == 2) { REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
        else if (KIND == 3) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));) }
        else if (KIND == 4) { REP16(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));) }
        else if (KIND == 5) { REP16(asm volatile("v_rsq_f64 %0, %0" : "+v"(a));) }
        else if (KIND == 6) { REP16(asm volatile("v_exp_f32 %0, %0" : "+v"(f));) }
        else if (KIND == 7) { REP16(asm volatile("v_cvt_f32_f64 %1, %0\n v_cvt_f64_f32 %0, %1" : "+v"(a), "+v"(f));) }
        else if (KIND == 8) { REP16(asm volatile("v_ldexp_f64 %0, %0, 0" : "+v"(a));) }
        else if (KIND == 9) { REP16(asm volatile("v_rndne_f64 %0, %0" : "+v"(a));) }
        else if (KIND == 10) { REP16(a = fma(dpp_shr1(1.0, a), b, c);) }                      // one scan stage: 2 dpp movs + fma
        else if (KIND == 11) { REP16(a = fma(bcast0(a), b, c);) }                              // readlane round trip + fma
        else if (KIND == 12) { REP16(if (__builtin_amdgcn_ballot_w64(a > 0.5) != 0ull) a = fma(a, b, c); else a = a * b;) }   // vote + branch + fma
        else if (KIND == 13) { REP16(a = a > 0.5 ? fma(a, b, c) : a * b;) }                    // compare + select
        else if (KIND == 14) {                                                                  // ds_read_b64, address depends on the value read
            REP16(asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)\n v_cvt_u32_f64 %1, %0\n v_lshlrev_b32 %1, 3, %1" : "+v"(a), "+v"(idx));)
        } else if (KIND == 15) {                                                                // ds_write_b64 + ds_read_b64 same address (own lane)
            REP16(lds[threadIdx.x] = a; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a = fma(lds[threadIdx.x], b, c); asm volatile("" ::: "memory");)
        } else if (KIND == 16) {                                                                // exchange: write own, barrier, read partner wave's
            REP16(lds[threadIdx.x] = a; __syncthreads(); a = fma(lds[((wave + 1) % nw) * 64 + lane], b, c); __syncthreads();)
        } else if (KIND == 17) {                                                                // exchange, one barrier, double-buffered
            REP4(lds[threadIdx.x] = a; __syncthreads(); a = fma(lds[((wave + 1) % nw) * 64 + lane], b, c);
                 lds[256 + threadIdx.x] = a; __syncthreads(); a = fma(lds[256 + ((wave + 1) % nw) * 64 + lane], b, c);
                 lds[threadIdx.x] = a; __syncthreads(); a = fma(lds[((wave + 1) % nw) * 64 + lane], b, c);
                 lds[256 + threadIdx.x] = a; __syncthreads(); a = fma(lds[256 + ((wave + 1) % nw) * 64 + lane], b, c);)
        } else if (KIND == 18) { REP16(__syncthreads(); a = fma(a, b, c);) }                   // barrier alone
        else if (KIND == 19) { REP16(asm volatile("s_nop 0");) }
        else if (KIND == 20) {                                                                  // v_cmp + s_and of vcc + scalar branch on it, value untouched
            REP16(if (__builtin_amdgcn_ballot_w64(a > 0.5) == ~0ull) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
        } else if (KIND == 21) {                                                               // uniform(): two readfirstlane + use as SGPR operand
            REP16(a = fma(a1, uni(a), c);)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + a1 + f + idx;
}

template <int KIND>
void run(const char *name, int blocks, int threads, double units_per_rep, double inst_per_unit) {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    const int iters = 500;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double units = units_per_rep * iters;
    printf("%-52s wg=%3d : %7.1f ticks/unit  %6.1f ticks/inst   (wall %.1f ns/unit)\n", name, threads, avg / units, avg / units / inst_per_unit,
           ms * 1e6 / units);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int B = 1024;   // one 64-thread workgroup per SIMD
    run<19>("s_nop 0 (clock calibration: 1 issue slot)", B, 64, 16, 1);
    run<0>("v_fma_f64 dependent", B, 64, 16, 1);
    run<1>("v_fma_f64 two chains (per pair)", B, 64, 16, 2);
    run<2>("v_mul_f64 dependent", B, 64, 16, 1);
    run<3>("v_add_f64 dependent", B, 64, 16, 1);
    run<4>("v_rcp_f64 dependent", B, 64, 16, 1);
    run<5>("v_rsq_f64 dependent", B, 64, 16, 1);
    run<6>("v_exp_f32 dependent", B, 64, 16, 1);
    run<7>("v_cvt_f32_f64 + v_cvt_f64_f32 dependent", B, 64, 16, 2);
    run<8>("v_ldexp_f64 dependent", B, 64, 16, 1);
    run<9>("v_rndne_f64 dependent", B, 64, 16, 1);
    run<10>("scan stage: 2 dpp mov + fma dependent", B, 64, 16, 3);
    run<11>("lane_bcast (2 readlane) + fma dependent", B, 64, 16, 3);
    run<12>("v_cmp + ballot + s_cbranch + fma dependent", B, 64, 16, 1);
    run<20>("v_cmp + ballot == all + branch around fma", B, 64, 16, 1);
    run<13>("v_cmp + fma + mul + 2 cndmask dependent", B, 64, 16, 1);
    run<21>("uniform (2 readfirstlane) + fma dependent", B, 64, 16, 1);
    run<14>("ds_read_b64 dependent address (+cvt+shift)", B, 64, 16, 1);
    run<15>("ds_write_b64 + wait + ds_read_b64 + fma", B, 64, 16, 1);
    for (int t : {64, 128, 256}) {
        run<18>("s_barrier + fma", B * 64 / t, t, 16, 1);
        run<16>("exchange: write, barrier, read partner, fma, barrier", B * 64 / t, t, 16, 1);
        run<17>("exchange double-buffered: write, barrier, read, fma", B * 64 / t, t, 16, 1);
    }
    return 0;
}
