// Micro-benchmark: issue cost (cycles per wave-instruction) of the VALU ops the lnprob kernel is made of, for
// one wave per SIMD and for several.  Build: hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int KIND>
__global__ void bench(double *out, int iters, unsigned long long *cyc) {
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = 0.999999, c = 1e-9;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3;
    int i0 = threadIdx.x, i1 = i0 + 1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {  // 8 independent f64 FMA chains
            REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                               "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 1) {  // one dependent f64 FMA chain
            REP16(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                               "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2"
                               : "+v"(a0) : "v"(b), "v"(c));)
        } else if (KIND == 2) {  // 8 independent f64 MUL
            REP16(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                               "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 3) {  // 4 independent f32 FMA chains x2
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                               "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"((float)b), "v"((float)c));)
        } else if (KIND == 4) {  // v_mov_b32 / v_cndmask mix (8 movs)
            REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n"
                               "v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %0" : "+v"(i0), "+v"(i1));)
        } else if (KIND == 5) {  // v_rcp_f64 x8 independent
            REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                               "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 6) {  // DPP movs (8)
            REP16(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "s_nop 1\n v_mov_b32_dpp %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                               "s_nop 1\n v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %1, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
                               "s_nop 1\n v_mov_b32_dpp %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %1, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
                               : "+v"(i0), "+v"(i1));)
        } else if (KIND == 7) {  // s_nop 0 x8
            REP16(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");)
        } else if (KIND == 8) {  // salu x8
            REP16(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                               "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(iters));)
            iters -= 128;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + i0 + i1;
}

template <int KIND>
void run(const char *name, int blocks, int threads) {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double ninst = 128.0 * iters;   // wave-instructions of the measured kind per wave
    printf("%-28s blocks=%5d thr=%4d : %.2f memtime-ticks/inst (s_memtime @100MHz => x%.0f cycles)  wall %.3f ms -> %.2f ns/inst/wave\n",
           name, blocks, threads, avg / ninst, 24.0, ms, ms * 1e6 / ninst);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int blocks = cfg == 0 ? 1024 : cfg == 1 ? 2048 : 1024, threads = cfg == 2 ? 256 : 64;
        printf("---- %d blocks x %d threads (%s)\n", blocks, threads, cfg == 0 ? "1 wave/SIMD" : cfg == 1 ? "2 waves/SIMD" : "4 waves/SIMD");
        run<0>("v_fma_f64 x8 independent", blocks, threads);
        run<1>("v_fma_f64 dependent chain", blocks, threads);
        run<2>("v_mul_f64 x8 independent", blocks, threads);
        run<3>("v_fma_f32 x4 chains", blocks, threads);
        run<4>("v_mov_b32 dependent", blocks, threads);
        run<5>("v_rcp_f64 x8 independent", blocks, threads);
        run<6>("s_nop1 + v_mov_b32_dpp (per pair)", blocks, threads);
        run<7>("s_nop 0", blocks, threads);
        run<8>("s_add_u32 dependent", blocks, threads);
    }
    return 0;
}
