// Micro-benchmark: VALU issue rates on gfx950 for the instructions an exact NTT could be built from.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;   // independent chains per thread

#define KERNEL32(name, asmstr)                                                        \
__global__ void name(uint32_t* out, uint32_t seed) {                                  \
    uint32_t a[UNROLL]; uint32_t b = seed | 1u, c = threadIdx.x * 2654435761u + 12345u;\
    for (int i = 0; i < UNROLL; i++) a[i] = threadIdx.x + i * 77u + seed;             \
    for (int it = 0; it < ITERS; it++) {                                              \
        _Pragma("unroll") for (int i = 0; i < UNROLL; i++)                            \
            asm volatile(asmstr : "+v"(a[i]) : "v"(b), "v"(c));                       \
    }                                                                                 \
    uint32_t s = 0; for (int i = 0; i < UNROLL; i++) s ^= a[i];                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                   \
}
KERNEL32(k_add_u32,   "v_add_u32 %0, %0, %1")
KERNEL32(k_xor,       "v_xor_b32 %0, %0, %1")
KERNEL32(k_mul_lo,    "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi,    "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u24,   "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mulhi_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u24,   "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_mad_u32,   "v_mad_u32_u16 %0, %0, %1, %2")
KERNEL32(k_fma_f32,   "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_alignbit,  "v_alignbit_b32 %0, %0, %1, 7")
KERNEL32(k_cndmask,   "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_addco,     "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_addc,      "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_lshl_add,  "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_add3,      "v_add3_u32 %0, %0, %1, %2")

#define KERNEL64(name, asmstr)                                                        \
__global__ void name(uint32_t* out, uint32_t seed) {                                  \
    uint64_t a[UNROLL]; uint32_t b = seed | 1u, c = threadIdx.x * 2654435761u + 12345u;\
    uint64_t d = ((uint64_t)c << 32) | b;                                             \
    for (int i = 0; i < UNROLL; i++) a[i] = threadIdx.x + i * 77u + seed;             \
    for (int it = 0; it < ITERS; it++) {                                              \
        _Pragma("unroll") for (int i = 0; i < UNROLL; i++)                            \
            asm volatile(asmstr : "+v"(a[i]) : "v"(b), "v"(c), "v"(d));               \
    }                                                                                 \
    uint64_t s = 0; for (int i = 0; i < UNROLL; i++) s ^= a[i];                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));           \
}
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_fma_f64,     "v_fma_f64 %0, %0, %3, %3")
KERNEL64(k_mul_f64,     "v_mul_f64 %0, %0, %3")
KERNEL64(k_add_f64,     "v_add_f64 %0, %0, %3")
KERNEL64(k_lshl_b64,    "v_lshlrev_b64 %0, 5, %0")
KERNEL64(k_lshr_b64,    "v_lshrrev_b64 %0, %1, %0")
KERNEL64(k_pk_fma_f32,  "v_pk_fma_f32 %0, %0, %3, %3")
KERNEL64(k_pk_add_f32,  "v_pk_add_f32 %0, %0, %3")
KERNEL64(k_pk_mul_f32,  "v_pk_mul_f32 %0, %0, %3")
KERNEL64(k_rndne_f64,   "v_rndne_f64 %0, %0")
KERNEL64(k_fract_f64,   "v_fract_f64 %0, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %3")
KERNEL64(k_cvt_f64_i32,  "v_cvt_f64_i32 %0, %1")
KERNEL32(k_bfe_i32,     "v_bfe_i32 %0, %0, 1, 23")
KERNEL32(k_subb,        "v_subb_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_cndmask_s,   "v_cndmask_b32 %0, %0, %1, s[10:11]")
KERNEL32(k_mov_dpp,     "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL32(k_permlane32,  "v_permlane32_swap_b32 %0, %1")

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Case { const char* name; kern_t k; };

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int threads = 256;
    Case cases[] = {
        {"v_add_u32", k_add_u32}, {"v_xor_b32", k_xor}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi},
        {"v_mul_u32_u24", k_mul_u24}, {"v_mul_hi_u32_u24", k_mulhi_u24}, {"v_mad_u32_u24", k_mad_u24},
        {"v_mad_u32_u16", k_mad_u32}, {"v_fma_f32", k_fma_f32}, {"v_alignbit_b32", k_alignbit},
        {"v_cndmask_b32", k_cndmask}, {"v_add_co_u32", k_addco}, {"v_addc_co_u32", k_addc},
        {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3},
        {"v_mad_u64_u32", k_mad_u64_u32}, {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64},
        {"v_lshlrev_b64", k_lshl_b64}, {"v_lshrrev_b64", k_lshr_b64}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_rndne_f64", k_rndne_f64},
        {"v_fract_f64", k_fract_f64}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_bfe_i32", k_bfe_i32}, {"v_subb_co_u32", k_subb}, {"v_cndmask_b32_sgpr", k_cndmask_s},
        {"v_mov_b32_dpp", k_mov_dpp}, {"v_permlane32_swap", k_permlane32},
    };
    for (int wpc : {2}) {          // workgroups (4 waves each) per CU -> waves per SIMD
        const int blocks = prop.multiProcessorCount * wpc;
        uint32_t* out; CK(hipMalloc(&out, (size_t)blocks * threads * 4));
        printf("--- %d waves/SIMD (blocks=%d x %d threads) : cycles per wave-instruction per SIMD\n", wpc, blocks, threads);
        for (auto& c : cases) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, out, 3u);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, out, 5u);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // per SIMD: wpc waves, each ITERS*UNROLL instrs
            double instr_per_simd = (double)wpc * ITERS * UNROLL;
            double cycles = ms * 1e-3 * 2.4e9;   // nominal 2.4 GHz
            printf("%-20s %8.3f ms  %6.2f cyc/instr (nominal 2.4GHz)\n", c.name, ms, cycles / instr_per_simd);
        }
        CK(hipFree(out));
    }
    return 0;
}
