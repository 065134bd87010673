// Micro-benchmark: real shader clock and FP64 issue rate under load on gfx950, as a function of waves per SIMD.
// Each wave runs chains of independent v_fma_f64 (or a mix) and reads s_memtime (shader clock) and
// s_memrealtime (100 MHz constant clock) around the loop: cycles per instruction in REAL shader cycles.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_clock tools/ubench_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int UNROLL = 16;

template <int KIND> __global__ void k_loop(uint64_t *out, double seed, int iters) {
    double a[UNROLL];
    for (int i = 0; i < UNROLL; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    const double b = 1.0000001, c = 1e-9;
    uint32_t x[UNROLL];
    for (int i = 0; i < UNROLL; i++) x[i] = threadIdx.x + i;
    const uint64_t t0 = __builtin_readcyclecounter();      // s_memtime: shader clock
    const uint64_t r0 = wall_clock64();                    // s_memrealtime: 100 MHz
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) % UNROLL]));
            if (KIND == 3) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                             asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) % UNROLL])); }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t r1 = wall_clock64();
    double s = 0; uint32_t sx = 0;
    for (int i = 0; i < UNROLL; i++) { s += a[i]; sx ^= x[i]; }
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = t1 - t0; out[3 * blockIdx.x + 1] = r1 - r0; }
    if (s == 12345.678 && sx == 77) out[3 * blockIdx.x + 2] = 1;
}

typedef void (*kern_t)(uint64_t *, double, int);

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clockRate=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    struct { const char *name; kern_t k; int per_iter; } cases[] = {
        {"v_fma_f64", k_loop<0>, UNROLL}, {"v_add_f64", k_loop<1>, UNROLL}, {"v_add_u32", k_loop<2>, UNROLL},
        {"fma_f64+add_u32 pairs", k_loop<3>, 2 * UNROLL}};
    const int iters = 200000;                               // ~10+ ms per launch: long enough for DVFS to settle
    for (int wps : {1, 2, 3, 4, 8}) {
        // blocks of 64 threads (one wave); wps waves per SIMD -> 4 * wps blocks per CU
        const int blocks = prop.multiProcessorCount * 4 * wps;
        uint64_t *out; CK(hipMalloc(&out, (size_t)blocks * 3 * 8));
        for (auto &c : cases) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, out, 3.0, 1000);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, out, 5.0, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint64_t> h(blocks * 3); CK(hipMemcpy(h.data(), out, blocks * 3 * 8, hipMemcpyDeviceToHost));
            std::vector<double> cyc, mhz;
            for (int b = 0; b < blocks; b++) { cyc.push_back((double)h[3 * b]); mhz.push_back((double)h[3 * b] / ((double)h[3 * b + 1] / 100.0)); }
            std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
            const double instr_per_wave = (double)iters * c.per_iter;
            printf("waves/SIMD %d  %-24s %7.2f ms  shader clock %6.0f MHz (median)  %.2f real cycles / wave-instr / SIMD  (%.2f per wave)\n",
                   wps, c.name, ms, mhz[blocks / 2], cyc[blocks / 2] / instr_per_wave / wps, cyc[blocks / 2] / instr_per_wave);
        }
        CK(hipFree(out));
    }
    return 0;
}
