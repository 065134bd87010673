// Micro-benchmark for VERDICT r2 item 4: can an INTEGER formulation of the exact negacyclic NTT beat the FP64-carried
// two-prime one on gfx950?  Three radix-2 Cooley-Tukey butterflies (a, b) <- (a + w b, a - w b) mod p, each as the inner
// loop an NTT would run (16 independent pairs per lane, lazy ranges where the arithmetic allows):
//   fp64x2 : the product's butterfly (pbs_kernels.hip): residues carried in doubles, p < 2^47, mulmod in 6 FP64 ops,
//            TWO primes cover the 2^93 range of one external product  -> 2 butterflies per coefficient pair
//   mont31 : 32-bit Montgomery, p < 2^31 (SURVEY H1 option b): FOUR primes needed for 2^93 (3 x 31 = 93 bits is not
//            enough with the sign) -> 4 butterflies per coefficient pair
//   gold64 : Goldilocks p = 2^64 - 2^32 + 1 with a 64 x 64 -> 128 product through v_mad_u64_u32 (SURVEY H1 option c:
//            key split into 3 limbs of 22 bits, ONE prime for the transforms of the digits, 3x the pointwise work and
//            3 inverse transforms per output polynomial)
// Output: VALU instructions per butterfly are read off the ISA (tools/ubench_butterfly.isa.txt is produced by
// `make`-less build below), time per butterfly per SIMD at 2 and 4 waves per SIMD is measured here.
// Build: hipcc --offload-arch=gfx950 -O3 --save-temps -o tools/ubench_butterfly tools/ubench_butterfly.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#pragma clang fp contract(off)
constexpr int PAIRS = 8;

// ---- fp64-carried modular butterfly (the product's) -------------------------------------------------------------
__device__ __forceinline__ double mulmod_f(double a, double w, double p, double pinv) {
    const double h = a * w;
    const double l = __builtin_fma(a, w, -h);
    const double q = __builtin_rint(h * pinv);
    return __builtin_fma(-q, p, h) + l;
}
__global__ void k_fp64(double *out, double seed, int iters) {
    const double p = 140737488273409.0, pinv = 1.0 / 140737488273409.0;
    double a[PAIRS], b[PAIRS], w[PAIRS];
    for (int i = 0; i < PAIRS; i++) { a[i] = seed + threadIdx.x + i; b[i] = seed * 3 + i + 0.5 * threadIdx.x; w[i] = 1234567.0 + 2 * i + threadIdx.x; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < PAIRS; i++) {
            const double v = mulmod_f(b[i], w[i], p, pinv);
            b[i] = a[i] - v;
            a[i] = a[i] + v;
        }
        if ((it & 7) == 7) {                                  // lazy range reset every few stages, like the NTT
#pragma unroll
            for (int i = 0; i < PAIRS; i++) { a[i] = __builtin_fma(-__builtin_rint(a[i] * pinv), p, a[i]); b[i] = __builtin_fma(-__builtin_rint(b[i] * pinv), p, b[i]); }
        }
    }
    double s = 0; for (int i = 0; i < PAIRS; i++) s += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- 32-bit Montgomery butterfly, values kept in [0, 2p) (Harvey lazy butterfly) ----------------------------------
__global__ void k_mont31(uint32_t *out, uint32_t seed, int iters) {
    const uint32_t p = 2013265921u, pinv_neg = 2013265919u;   // 15 * 2^27 + 1, -p^-1 mod 2^32
    uint32_t a[PAIRS], b[PAIRS], w[PAIRS];
    for (int i = 0; i < PAIRS; i++) { a[i] = (seed + threadIdx.x + i) % p; b[i] = (seed * 3 + i + 7 * threadIdx.x) % p; w[i] = (1234567u + 2 * i + threadIdx.x) % p; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < PAIRS; i++) {
            const uint64_t t = (uint64_t)b[i] * w[i];                       // v_mad_u64_u32 (or mul_lo + mul_hi)
            const uint32_t m = (uint32_t)t * pinv_neg;                      // v_mul_lo_u32
            const uint32_t v = (uint32_t)((t + (uint64_t)m * p) >> 32);     // v_mad_u64_u32, high half: in [0, 2p)
            uint32_t x = a[i] >= 2 * p ? a[i] - 2 * p : a[i];               // v_subrev + v_min (lazy correction)
            a[i] = x + v;                                                   // [0, 4p)
            b[i] = x + 2 * p - v;                                           // [0, 4p)
        }
    }
    uint32_t s = 0; for (int i = 0; i < PAIRS; i++) s ^= a[i] ^ b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- Goldilocks butterfly: 64 x 64 -> 128 product, reduction by 2^64 = 2^32 - 1 (mod p) ---------------------------
__device__ __forceinline__ uint64_t gold_reduce(unsigned __int128 x) {
    const uint64_t P = 0xFFFFFFFF00000001ull;
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    const uint64_t hh = hi >> 32, hl = hi & 0xFFFFFFFFull;
    uint64_t t = lo - hh; if (lo < hh) t += P;              // - hh * 2^96 = + ... (2^96 = -1 mod p)
    const uint64_t u = hl * 0xFFFFFFFFull;                  // hl * 2^64 = hl * (2^32 - 1)
    uint64_t r = t + u; if (r < u || r >= P) r -= P;
    return r;
}
__global__ void k_gold(uint64_t *out, uint64_t seed, int iters) {
    const uint64_t P = 0xFFFFFFFF00000001ull;
    uint64_t a[PAIRS], b[PAIRS], w[PAIRS];
    for (int i = 0; i < PAIRS; i++) { a[i] = seed + threadIdx.x + i; b[i] = seed * 3 + i + 11 * threadIdx.x; w[i] = 0x123456789abcull + 2 * i + threadIdx.x; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < PAIRS; i++) {
            const uint64_t v = gold_reduce((unsigned __int128)b[i] * w[i]);
            const uint64_t x = a[i];
            uint64_t s = x + v; if (s < v || s >= P) s -= P;
            uint64_t d = x - v; if (x < v) d += P;
            a[i] = s; b[i] = d;
        }
    }
    uint64_t s = 0; for (int i = 0; i < PAIRS; i++) s ^= a[i] ^ b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d\n", prop.name, prop.multiProcessorCount);
    const int iters = 20000;
    void *buf; CK(hipMalloc(&buf, (size_t)prop.multiProcessorCount * 16 * 64 * 8));
    for (int wps : {2, 4}) {
        const int blocks = prop.multiProcessorCount * 4 * wps;
        for (int kind = 0; kind < 3; kind++) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                if (rep) CK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL(k_fp64, dim3(blocks), dim3(64), 0, 0, (double *)buf, 3.0, iters);
                if (kind == 1) hipLaunchKernelGGL(k_mont31, dim3(blocks), dim3(64), 0, 0, (uint32_t *)buf, 3u, iters);
                if (kind == 2) hipLaunchKernelGGL(k_gold, dim3(blocks), dim3(64), 0, 0, (uint64_t *)buf, 3ull, iters);
                CK(hipDeviceSynchronize());
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bf_per_simd = (double)wps * iters * PAIRS;            // wave-butterflies (64 lanes each) per SIMD
            const double ns = ms * 1e6 / bf_per_simd;
            const char *name[3] = {"fp64 (p < 2^47), x2 primes", "mont31 (p < 2^31), x4 primes", "gold64, x1 prime (x3 key limbs)"};
            const double primes[3] = {2, 4, 1};
            printf("waves/SIMD %d  %-34s %7.2f ms  %6.2f ns per wave-butterfly per SIMD  -> x primes: %6.2f ns per coefficient pair\n",
                   wps, name[kind], ms, ns, ns * primes[kind]);
        }
    }
    return 0;
}
