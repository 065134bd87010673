// Keyswitch (big LWE key -> small LWE key) on the matrix cores (gfx950 only).
//
//   ks[ct][j] = b * [j == 742] - sum_{i < 2048, l < 5} d(ct, i, l) * KSK[i][l][j]      (mod 2^64)
//
// is an integer matrix product (B x 10240) x (10240 x 743) whose left entries are balanced base-8 digits
// (d in [-4, 3]).  The 64-bit right entries are split into 8 balanced signed bytes,
// KSK = sum_b s_b * 2^(8b) (mod 2^64, s_b in [-128, 127]), so the product becomes 8 i8 x i8 -> i32 GEMMs
// (|S_b| <= 4 * 128 * 10240 < 2^23: no overflow) that share the digit operand, and
// ks = body - sum_b (S_b << 8b) is exact in wrapping 64-bit arithmetic.
// One wavefront owns a 32-ciphertext x 32-column tile for all 8 byte planes (8 x 16 accumulator registers)
// and walks K in steps of 32 with v_mfma_i32_32x32x32_i8.  Both operands are stored in fragment order
// ([tile][k-step][lane][16 bytes]) so that every operand load is one coalesced 1-KB access and no LDS is
// needed; the 4 wavefronts of a workgroup share the key fragments through L1.
// Replaces tfhe's keyswitch inside shortint::ServerKey::apply_lookup_table.
#include <cstdlib>

#include "pbs_kernels.h"

namespace fhs {

namespace {
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int KS_K = BIG_N * KS_LEVEL;          // 10240
constexpr int KS_STEPS = KS_K / 32;             // 320 k-steps of 32
constexpr int KS_COL_TILES = 24;                // 24 x 32 = 768 >= 743 columns
constexpr int KS_PLANES = 8;

// byte offset of k inside a row-tile's fragment stream: [k-step][lane = 32*(k%32/16) + r][k%16]
__device__ __forceinline__ size_t frag_off(int r, int k) {
    return ((size_t)(k >> 5) * 64 + ((k >> 4) & 1) * 32 + r) * 16 + (k & 15);
}
}  // namespace

// digits of a batch in fragment order: dig[tile][320][64][16]; rows >= B are zero
__global__ __launch_bounds__(256) void ks_digits_kernel(const uint64_t *__restrict__ in, int8_t *__restrict__ dig, int B) {
    const int ct = blockIdx.x;
    int8_t *dst = dig + (size_t)(ct >> 5) * KS_STEPS * 1024;
    const int r = ct & 31;
    for (int i = threadIdx.x; i < BIG_N; i += 256) {
        int d[KS_LEVEL];
#pragma unroll
        for (int l = 0; l < KS_LEVEL; l++) d[l] = 0;
        if (ct < B) {
            const uint64_t a = in[(size_t)ct * BIG_CT + i];
            uint32_t v = (uint32_t)((a + (1ull << 48)) >> 49);   // closest representable on 15 bits
#pragma unroll
            for (int l = KS_LEVEL - 1; l >= 0; l--) {            // least significant level first
                int x = (int)(v & 7u);
                v >>= 3;
                if (x >= 4) { x -= 8; v += 1; }
                d[l] = x;
            }
        }
#pragma unroll
        for (int l = 0; l < KS_LEVEL; l++) dst[frag_off(r, i * KS_LEVEL + l)] = (int8_t)d[l];
    }
}

// The same digits, one workgroup per TILE of 32 ciphertexts (round 4): lane = 32 * (k half) + row assembles the 16 bytes
// of its fragment slot in registers, so a wavefront writes each 1-KB fragment with ONE coalesced 16-byte store per lane
// (the kernel above writes single bytes, 16-byte runs 512 bytes apart: 0.13 ms per 3968 rows where 40 MB of output
// should take 0.03).  A lane's 16 digits of k-step ks, half h are k = 32 ks + 16 h .. + 15, i.e. digits of the
// coefficients (32 ks + 16 h) / 5 .. (32 ks + 16 h + 15) / 5 -- four of them, re-read from L1 by the next k-step.
constexpr int KS_DIG_CHUNKS = 8;
__global__ __launch_bounds__(256) void ks_digits_tile_kernel(const uint64_t *__restrict__ in, int8_t *__restrict__ dig, int B) {
    const int tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int ct = tile * 32 + r;
    const uint64_t *row = in + (size_t)(ct < B ? ct : 0) * BIG_CT;
    v4i *dst = reinterpret_cast<v4i *>(dig + (size_t)tile * KS_STEPS * 1024) + lane;
    // blockIdx.y = one of KS_DIG_CHUNKS slices of the 320 k-steps: ~1 000 workgroups for 3968 rows instead of 124
    const int ks_begin = blockIdx.y * (KS_STEPS / KS_DIG_CHUNKS), ks_end = ks_begin + KS_STEPS / KS_DIG_CHUNKS;
    for (int ks = ks_begin + wave; ks < ks_end; ks += 4) {
        const int k0 = ks * 32 + h * 16;
        const int i0 = k0 / KS_LEVEL, l0 = k0 - i0 * KS_LEVEL;     // first coefficient and its first level in this slot
        uint32_t w[4] = {0, 0, 0, 0};
        int pos = -l0;                                             // byte position of level 0 of coefficient i0 + c
#pragma unroll
        for (int c = 0; c < 4; c++, pos += KS_LEVEL) {
            const int i = i0 + c;
            int d[KS_LEVEL] = {0, 0, 0, 0, 0};
            if (ct < B && i < BIG_N) {
                const uint64_t a = row[i];
                uint32_t v = (uint32_t)((a + (1ull << 48)) >> 49);   // closest representable on 15 bits
#pragma unroll
                for (int l = KS_LEVEL - 1; l >= 0; l--) {            // least significant level first
                    int x = (int)(v & 7u);
                    v >>= 3;
                    if (x >= 4) { x -= 8; v += 1; }
                    d[l] = x;
                }
            }
#pragma unroll
            for (int l = 0; l < KS_LEVEL; l++) {
                const int p = pos + l;
                if (p >= 0 && p < 16) w[p >> 2] |= (uint32_t)(uint8_t)(int8_t)d[l] << (8 * (p & 3));
            }
        }
        v4i o;
        o.x = (int)w[0]; o.y = (int)w[1]; o.z = (int)w[2]; o.w = (int)w[3];
        dst[(size_t)ks * 64] = o;
    }
}

// KSK [10240][743] u64 -> planes[8][24 col tiles][320][64][16] balanced signed bytes (columns >= 743 are zero)
__global__ __launch_bounds__(256) void ksk_to_planes_kernel(const uint64_t *__restrict__ ksk, int8_t *__restrict__ planes) {
    const int k = blockIdx.x;                    // 0..10239
    for (int j = threadIdx.x; j < KS_COL_TILES * 32; j += 256) {
        uint64_t w = j < SMALL_CT ? ksk[(size_t)k * SMALL_CT + j] : 0;
        const size_t off = (size_t)(j >> 5) * KS_STEPS * 1024 + frag_off(j & 31, k);
#pragma unroll
        for (int b = 0; b < KS_PLANES; b++) {
            const int8_t s = (int8_t)(w & 0xff);
            w = (w - (uint64_t)(int64_t)s) >> 8;     // exact: the low byte is cleared before the shift
            planes[(size_t)b * KS_COL_TILES * KS_STEPS * 1024 + off] = s;
        }
    }
}

__global__ __launch_bounds__(256) void keyswitch_mfma_kernel(const int8_t *__restrict__ dig, const int8_t *__restrict__ planes,
                                                             const uint64_t *__restrict__ in, uint64_t *__restrict__ ks_out,
                                                             int B, int steps_per_split) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // the 4 wavefronts of a workgroup take 4 ciphertext tiles of the SAME column tile: the 8 key-byte
    // fragments (8 KB per k-step, the heavy stream) are then shared through L1, the digit fragments are not
    const int tile = blockIdx.x * 4 + wave;                // 32 ciphertexts
    const int cg = blockIdx.y;                             // 32 columns
    const v4i *ap = reinterpret_cast<const v4i *>(dig + (size_t)tile * KS_STEPS * 1024) + lane;
    const v4i *bp = reinterpret_cast<const v4i *>(planes + (size_t)cg * KS_STEPS * 1024) + lane;
    constexpr size_t PLANE_V4 = (size_t)KS_COL_TILES * KS_STEPS * 64;   // v4i elements per plane

    v16i acc[KS_PLANES];
#pragma unroll
    for (int b = 0; b < KS_PLANES; b++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[b][e] = 0;

    // split-K over gridDim.z (small batches): partial sums are combined with 64-bit atomics on a zeroed output
    const int ks0 = blockIdx.z * steps_per_split, ks1 = ks0 + steps_per_split;
    v4i a = ap[(size_t)ks0 * 64];
    v4i bb[KS_PLANES];
#pragma unroll
    for (int b = 0; b < KS_PLANES; b++) bb[b] = bp[b * PLANE_V4 + (size_t)ks0 * 64];
    for (int ks = ks0; ks < ks1; ks++) {
        const int nx = ks + 1 < ks1 ? ks + 1 : ks;         // last step reloads itself (keeps the loop branch-free)
        const v4i a_n = ap[(size_t)nx * 64];
        v4i b_n[KS_PLANES];
#pragma unroll
        for (int b = 0; b < KS_PLANES; b++) b_n[b] = bp[b * PLANE_V4 + (size_t)nx * 64];
#pragma unroll
        for (int b = 0; b < KS_PLANES; b++) acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bb[b], acc[b], 0, 0, 0);
        a = a_n;
#pragma unroll
        for (int b = 0; b < KS_PLANES; b++) bb[b] = b_n[b];
    }

    // C/D layout of the 32x32 shapes: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int col = cg * 32 + (lane & 31);
    if (col >= SMALL_CT) return;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int ct = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (ct >= B) continue;
        uint64_t v = 0;
#pragma unroll
        for (int b = 0; b < KS_PLANES; b++) v += (uint64_t)(int64_t)acc[b][e] << (8 * b);
        uint64_t o = (uint64_t)0 - v;
        if (col == LWE_N && blockIdx.z == 0) o += in[(size_t)ct * BIG_CT + BIG_N];
        if (gridDim.z == 1) ks_out[(size_t)ct * SMALL_CT + col] = o;
        else atomicAdd(reinterpret_cast<unsigned long long *>(ks_out + (size_t)ct * SMALL_CT + col), (unsigned long long)o);
    }
}

// Wide batches (round 4).  The kernel above is LATENCY-bound, not MFMA- or bandwidth-bound: one fragment set (9 KB) in
// flight per wavefront, two wavefronts per SIMD, 8 matrix instructions (256 cycles) between round trips to the L2 --
// 12 % of the i8 matrix peak at 3968 rows.  Here
//  * a wavefront owns 64 ciphertexts x 32 columns x 8 planes: 16 accumulators = 256 registers, the accumulation half of
//    the unified register file (one wavefront per SIMD), and every key fragment feeds two matrix instructions;
//  * the operands reach the matrix cores through an LDS ring of KS2_DEPTH k-steps filled by direct global -> LDS loads
//    (global_load_lds_dwordx4: a wave-wide 1-KB fragment lands in LDS in fragment order, no registers held while it
//    flies): the 8 key fragments of a k-step are loaded ONCE per workgroup and read by its 4 wavefronts, each wavefront
//    adds its own two digit fragments -- 16 KB per k-step and workgroup, KS2_DEPTH - 1 k-steps in flight behind 512 cycles
//    of matrix work per k-step and wavefront;
//  * workgroup -> tile mapping is XCD-aware: workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share an
//    L2), so XCD x takes column tiles 3x .. 3x + 2 for every ciphertext group -- each 4 MB L2 streams 3 of the 24 key
//    column tiles (7.9 MB) to all its workgroups instead of all 24.
constexpr int KS2_DEPTH = 8;
constexpr int KS2_GROUP = 256;                  // ciphertexts per workgroup
constexpr int KS2_MIN_BATCH = 129;              // below: the split-K kernel above (57 against 72 us at 64 rows; 89 against 76 at 256; FHS_KS_OLD_BELOW=n overrides for A/B runs)
constexpr int KS2_STAGE_BYTES = 16 * 1024;      // [8 key fragments][4 wavefronts x 2 digit fragments] of 1 KB
constexpr int KS2_LDS_BYTES = KS2_DEPTH * KS2_STAGE_BYTES;

namespace {
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
// one wave-wide 1-KB fragment: lane l's 16 bytes at g + l * 16 go to lds + l * 16
__device__ __forceinline__ void frag_to_lds(const char *g_uniform, uint32_t lane_off, char *lds_uniform) {
    __builtin_amdgcn_global_load_lds((gptr_t)(g_uniform + lane_off), (lptr_t)lds_uniform, 16, 0, 0);
}
}  // namespace

__global__ __launch_bounds__(256, 1) void keyswitch_mfma2_kernel(const int8_t *__restrict__ dig, const int8_t *__restrict__ planes,
                                                                 const uint64_t *__restrict__ in, uint64_t *__restrict__ ks_out,
                                                                 int B, int n_groups, int full_blocks, int rem_tiles,
                                                                 int rem_slices) {
    extern __shared__ __attribute__((aligned(16))) char ks2_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // 24 n_groups tiles on 256 CUs: the tiles beyond the last whole round (rem_tiles of them) are cut into rem_slices
    // slices of K each, so that they fill ONE partial round instead of costing a whole one (3072 rows: 288 tiles = one
    // round + 32 tiles x 8 slices); the slices add into rows the host zeroed (64-bit atomics, exact: wrapping sums)
    int bid = blockIdx.x, ks_begin = 0, ks_end = KS_STEPS;
    const bool sliced = bid >= full_blocks;
    if (sliced) {
        const int j = bid - full_blocks, slice = j / rem_tiles;
        bid = full_blocks + j % rem_tiles;
        ks_begin = slice * (KS_STEPS / rem_slices);
        ks_end = ks_begin + KS_STEPS / rem_slices;
    }
    const int xcd = bid & 7, idx = bid >> 3;
    const int cg = xcd * 3 + idx % 3;                      // 32 columns
    const int g = idx / 3;                                 // 256 ciphertexts
    if (g >= n_groups) return;
    const int tile0 = (g * 4 + wave) * 2;                  // two tiles of 32 ciphertexts
    // what THIS wavefront brings in per k-step: key planes 2 wave, 2 wave + 1 (shared) and its own two digit fragments
    const char *src[4];
    src[0] = reinterpret_cast<const char *>(planes) + ((size_t)(2 * wave) * KS_COL_TILES + cg) * KS_STEPS * 1024;
    src[1] = reinterpret_cast<const char *>(planes) + ((size_t)(2 * wave + 1) * KS_COL_TILES + cg) * KS_STEPS * 1024;
    src[2] = reinterpret_cast<const char *>(dig) + (size_t)tile0 * KS_STEPS * 1024;
    src[3] = src[2] + (size_t)KS_STEPS * 1024;
    const int dst[4] = {(2 * wave) * 1024, (2 * wave + 1) * 1024, (8 + 2 * wave) * 1024, (9 + 2 * wave) * 1024};
    const uint32_t lane_off = (uint32_t)lane * 16u;
    auto fetch = [&](int ks, int ring) {                   // k-step ks -> ring slot `ring` % KS2_DEPTH
        char *slot = ks2_smem + (ring % KS2_DEPTH) * KS2_STAGE_BYTES;
#pragma unroll
        for (int f = 0; f < 4; f++) frag_to_lds(src[f], lane_off + (uint32_t)ks * 1024u, slot + dst[f]);
    };

    v16i acc[2][KS_PLANES];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int b = 0; b < KS_PLANES; b++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[t][b][e] = 0;

#pragma unroll
    for (int d = 0; d < KS2_DEPTH - 1; d++) fetch(ks_begin + d, ks_begin + d);
    for (int ks = ks_begin; ks < ks_end; ks++) {
        // this wavefront's fragments of k-step ks have landed when at most the 4 x (KS2_DEPTH - 2) younger ones are
        // still in flight; the barrier extends that to the other wavefronts' -- and says every wavefront is done reading
        // the slot of k-step ks - 1, which the fetch below overwrites
        // (a bare s_barrier: __syncthreads() would add a fence that waits for EVERY outstanding load, vmcnt(0))
        // lgkmcnt(0): this wavefront's LDS reads of the previous k-step have RETURNED before it tells the others (by
        // arriving) that the slot may be overwritten
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (KS2_DEPTH - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        const int nx = ks + KS2_DEPTH - 1;
        fetch(nx < ks_end ? nx : ks_end - 1, nx);          // tail: the in-flight count stays constant (the last k-step again, into a dead slot)
        const char *slot = ks2_smem + (ks % KS2_DEPTH) * KS2_STAGE_BYTES + lane_off;
        const v4i a0 = *reinterpret_cast<const v4i *>(slot + (8 + 2 * wave) * 1024);
        const v4i a1 = *reinterpret_cast<const v4i *>(slot + (9 + 2 * wave) * 1024);
#pragma unroll
        for (int b = 0; b < KS_PLANES; b++) {
            const v4i kb = *reinterpret_cast<const v4i *>(slot + b * 1024);
            acc[0][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, kb, acc[0][b], 0, 0, 0);
            acc[1][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, kb, acc[1][b], 0, 0, 0);
        }
    }

    // the tail's dummy fetches are still writing into this workgroup's LDS: let them land before any wavefront can
    // leave (the allocation goes to the next workgroup when the last one does)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int col = cg * 32 + (lane & 31);
    if (col >= SMALL_CT) return;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        // pin the second tile's accumulators to the accumulation registers while the first tile is written out: read
        // into vector registers all at once (what the register allocator does at the loop exit) the 256 of them spill
        // into the main loop
#pragma unroll
        for (int b = 0; b < KS_PLANES; b++) asm volatile("" : "+a"(acc[1][b]));
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int ct = (tile0 + t) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            // straight-line arithmetic, one output at a time (scheduling barriers): hoisted above per-output branches the
            // 256 accumulator reads would all be live at once and spill into the main loop
            uint64_t v = 0;
#pragma unroll
            for (int b = 0; b < KS_PLANES; b++) v += (uint64_t)(int64_t)acc[t][b][e] << (8 * b);
            uint64_t o = (uint64_t)0 - v;
            const int ctc = ct < B ? ct : B - 1;          // rows >= B: computed on zero digits, never stored
            if (col == LWE_N && ks_begin == 0) o += in[(size_t)ctc * BIG_CT + BIG_N];
            __builtin_amdgcn_sched_barrier(0);
            if (ct < B) {
                if (sliced) atomicAdd(reinterpret_cast<unsigned long long *>(ks_out + (size_t)ct * SMALL_CT + col), (unsigned long long)o);
                else ks_out[(size_t)ct * SMALL_CT + col] = o;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

hipError_t prepare_device_for_keyswitch() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(keyswitch_mfma2_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, KS2_LDS_BYTES);
}

size_t ks_planes_bytes() { return (size_t)KS_PLANES * KS_COL_TILES * KS_STEPS * 1024; }
size_t ks_digits_bytes(int B) { return (size_t)((B + 255) / 256) * 8 * KS_STEPS * 1024; }   // whole groups of 8 tiles

hipError_t launch_ksk_to_planes(const uint64_t *d_ksk, int8_t *d_planes, hipStream_t s) {
    hipLaunchKernelGGL(ksk_to_planes_kernel, dim3(KS_K), dim3(256), 0, s, d_ksk, d_planes);
    return hipGetLastError();
}

hipError_t launch_keyswitch_mfma(const uint64_t *d_in, const int8_t *d_planes, int8_t *d_dig, uint64_t *d_ks_out, int B,
                                 hipStream_t s, int n_cus) {
    if (B <= 0) return hipSuccess;
    const int tiles = (B + 31) / 32;
    static const int old_below = [] { const char *v = std::getenv("FHS_KS_OLD_BELOW"); return v ? std::atoi(v) : KS2_MIN_BATCH; }();
    const bool wide = B >= old_below;
    if (wide)
        hipLaunchKernelGGL(ks_digits_tile_kernel, dim3(((B + KS2_GROUP - 1) / KS2_GROUP) * (KS2_GROUP / 32), KS_DIG_CHUNKS),
                           dim3(256), 0, s, d_in, d_dig, B);
    else
        hipLaunchKernelGGL(ks_digits_kernel, dim3(((tiles + 3) / 4) * 128), dim3(256), 0, s, d_in, d_dig, B);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (wide) {
        const int n_groups = (B + KS2_GROUP - 1) / KS2_GROUP;
        const int tiles2 = KS_COL_TILES * n_groups, slots = n_cus > 0 ? n_cus : 256;   // one workgroup per CU
        int full = tiles2 / slots * slots, rem = tiles2 - full, slices = 1;
        if (rem > 0)
            for (int c : {8, 4, 2})
                if (rem * c <= slots) { slices = c; break; }
        if (slices == 1) { full = tiles2; rem = 0; }
        else {
            // rows of the groups that own a sliced tile: zeroed, then overwritten by their unsliced tiles and added to
            // by the slices (stream order: memset, then the one kernel)
            const size_t row0 = (size_t)(full / 8 / 3) * KS2_GROUP;
            if (row0 < (size_t)B) {
                e = hipMemsetAsync(d_ks_out + row0 * SMALL_CT, 0, ((size_t)B - row0) * SMALL_CT * 8, s);
                if (e != hipSuccess) return e;
            }
        }
        hipLaunchKernelGGL(keyswitch_mfma2_kernel, dim3(full + rem * slices), dim3(256), KS2_LDS_BYTES, s, d_dig, d_planes,
                           d_in, d_ks_out, B, n_groups, full, rem > 0 ? rem : 1, slices);
        return hipGetLastError();
    }
    int splits = 1;                                   // fill the 256 CUs when the batch is small
    const int tgroups = (tiles + 3) / 4;
    while (splits < 16 && tgroups * KS_COL_TILES * splits < 256) splits *= 2;
    if (splits > 1) {
        e = hipMemsetAsync(d_ks_out, 0, (size_t)B * SMALL_CT * 8, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(keyswitch_mfma_kernel, dim3(tgroups, KS_COL_TILES, splits), dim3(256), 0, s, d_dig, d_planes,
                       d_in, d_ks_out, B, KS_STEPS / splits);
    return hipGetLastError();
}

}  // namespace fhs
