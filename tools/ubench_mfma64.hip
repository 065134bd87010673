// Micro-benchmark: does v_mfma_f64_16x16x4_f64 on gfx950 run beside the FP64 vector pipe or on it?
//   (1) cycles per MFMA (independent accumulators) at 1..4 waves per SIMD
//   (2) one wave issuing MFMA and v_fma_f64 interleaved: do the FMAs hide under the MFMA?
//   (3) two waves per SIMD, one issuing only MFMA and the other only v_fma_f64: max or sum of the two?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mfma64 tools/ubench_mfma64.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef double __attribute__((ext_vector_type(4))) double4_t;
constexpr int NACC = 8;

// mode 0: MFMA only; 1: FMA only; 2: per MFMA `ratio` FMAs in the same wave; 3: waves < 4 MFMA only, waves >= 4 FMA only
template <int NFMA> __global__ void k_mix(uint64_t *out, double seed, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    double4_t acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = double4_t{seed, seed + 1, seed + 2, seed + 3};
    double f[16];
    for (int i = 0; i < 16; i++) f[i] = seed + threadIdx.x * 1e-3 + i;
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9, c = 1.0000001;
    const bool do_mfma = mode == 0 || mode == 2 || (mode == 3 && wave < 4);
    const bool do_fma = mode == 1 || mode == 2 || (mode == 3 && wave >= 4);
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (do_mfma) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            if (do_fma) {
#pragma unroll
                for (int k = 0; k < NFMA; k++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[k % 16]) : "v"(c), "v"(b));
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 16; i++) s += f[i];
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
    if (s == 12345.678) out[0] = 1;
}

typedef void (*kern_t)(uint64_t *, double, int, int);
static kern_t pick(int ratio) {
    switch (ratio) { case 4: return k_mix<4>; case 8: return k_mix<8>; case 12: return k_mix<12>; case 24: return k_mix<24>; case 32: return k_mix<32>; default: return k_mix<16>; }
}
static float run(int blocks, int threads, size_t lds, uint64_t *out, int iters, int mode, int ratio) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern_t k = pick(ratio);
    if (lds) CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, out, 3.0, 200, mode);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, out, 5.0, iters, mode);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs=%d\n", prop.name, cus);
    uint64_t *out; CK(hipMalloc(&out, (size_t)cus * 64 * 8 * 8));
    std::vector<uint64_t> h((size_t)cus * 64 * 8);
    const int iters = 20000;
    auto cyc = [&](int blocks, int waves) {
        CK(hipMemcpy(h.data(), out, (size_t)blocks * 8 * 8, hipMemcpyDeviceToHost));
        std::vector<uint64_t> v;
        for (int b = 0; b < blocks; b++) for (int w = 0; w < waves; w++) v.push_back(h[(size_t)b * 8 + w]);
        std::sort(v.begin(), v.end());
        return (double)v[v.size() / 2];
    };
    // (1) one-wave blocks, wps waves per SIMD
    for (int wps : {1, 2, 4}) {
        const int blocks = cus * 4 * wps;
        float ms = run(blocks, 64, 0, out, iters, 0, 0);
        printf("[MFMA-only launch %.2f ms for %d MFMA per wave] ", ms, iters * NACC);
        const double c0 = cyc(blocks, 1);
        ms = run(blocks, 64, 0, out, iters, 1, 0);
        const double c1 = cyc(blocks, 1);
        printf("waves/SIMD %d: MFMA only %.1f shader-clock ticks per MFMA per wave (x%d waves: %.1f per SIMD-MFMA) | FMA only %.2f ticks per v_fma_f64 per wave (%.2f per SIMD-FMA)  [%.2f ms]\n",
               wps, c0 / ((double)iters * NACC), wps, c0 / ((double)iters * NACC) / wps, c1 / ((double)iters * NACC * 16), c1 / ((double)iters * NACC * 16) / wps, ms);
    }
    // (2) same wave: 1 MFMA + ratio FMAs
    for (int ratio : {0, 4, 8, 12, 16, 24, 32}) {
        const int blocks = cus * 4 * 2;
        run(blocks, 64, 0, out, iters, ratio ? 2 : 0, ratio);
        const double c0 = cyc(blocks, 1);
        float ms1 = run(blocks, 64, 0, out, iters, 1, ratio ? ratio : 16);
        const double c1 = cyc(blocks, 1);
        printf("2 waves/SIMD, each wave 1 MFMA + %2d FMA: %.1f ticks per group per wave | the FMAs alone: %.1f  [%.2f ms]\n", ratio, c0 / ((double)iters * NACC), c1 / ((double)iters * NACC), ms1);
    }
    // (3) 8-wave workgroups, one per CU (64 KB of LDS each... 160 KB per CU: ask for 96 KB): waves 0-3 MFMA, 4-7 FMA
    {
        const size_t lds = 96 * 1024;
        for (int mode : {0, 1, 3}) {
            run(cus, 512, lds, out, iters, mode, 0);
            CK(hipMemcpy(h.data(), out, (size_t)cus * 8 * 8, hipMemcpyDeviceToHost));
            std::vector<uint64_t> lo, hi;
            for (int b = 0; b < cus; b++) for (int w = 0; w < 8; w++) (w < 4 ? lo : hi).push_back(h[(size_t)b * 8 + w]);
            std::sort(lo.begin(), lo.end()); std::sort(hi.begin(), hi.end());
            printf("8-wave workgroup per CU, mode %d (0 all MFMA, 1 all FMA x16, 3 waves 0-3 MFMA / 4-7 FMA x16): waves 0-3 %.1f ticks per group, waves 4-7 %.1f ticks per group\n",
                   mode, lo[lo.size() / 2] / ((double)iters * NACC), hi[hi.size() / 2] / ((double)iters * NACC));
        }
    }
    return 0;
}
