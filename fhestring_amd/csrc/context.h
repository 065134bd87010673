// Host runtime: device context, key material, batched-PBS dispatch, kernel timing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "dist.h"
#include "fft_tables.h"
#include "ntt_tables.h"
#include "pbs_kernels.h"

namespace fhs {

// grow-only device buffer
struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes);
    void release();
    template <class T> T *as() const { return reinterpret_cast<T *>(ptr); }
};

// HIP-event timing of the two PBS kernels on the stream they are launched on
struct KernelTimer {
    struct Pending { hipEvent_t e0, e1; int kind; uint64_t units; uint32_t launches; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    // 0 = blind rotation (exact NTT kernel, or the 2-wavefront FFT kernel), 1 = keyswitch,
    // 2 = blind rotation on the 4-wavefront FFT kernel (batches <= fft4_max_batch)
    double ms[3] = {0, 0, 0};
    uint64_t n[3] = {0, 0, 0};       // KERNEL launches covered (a blind rotation cut into one-round launches counts each of them:
                                     // the per-launch average then is what `rocprofv3 --kernel-trace --stats` reports per kernel)
    uint64_t units[3] = {0, 0, 0};   // PBS covered by the timed launches
    bool enabled = true;
    hipEvent_t get();
    void begin(int kind, uint64_t units, hipStream_t s);
    void end(hipStream_t s, uint32_t kernel_launches = 1);
    void resolve();   // synchronises pending events and accumulates
    void reset();
    void destroy();
};

class Context {
  public:
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool key_loaded = false;

    // key material on device
    int8_t *d_ksk_planes = nullptr;  // KSK as 8 byte planes in MFMA fragment order (ks_kernels.hip)
    double *d_bsk_ntt = nullptr;
    double *d_tables = nullptr;   // fwd_uni | fwd_lane | inv_uni | inv_lane
    NttTables tw{};
    double crt_c = 0;

    // optional f64-FFT arithmetic (fft_kernels.hip): 0 = exact two-prime NTT (default), 1 = f64 FFT.
    // Select before load_server_key: the Fourier-domain key is only built when the mode asks for it.
    int arith = 0;
    double *d_bsk_fft = nullptr;
    uint64_t *d_bsk_std = nullptr;   // standard-domain key, kept for a later conversion to the Fourier domain (build_fft_key)
    double *d_fft_tables = nullptr;   // lanetab[12][64] | weff[1024][2] | mono[4096][2] | r16[16][2]
    uint32_t *d_work_counter = nullptr;   // persistent-workgroup ciphertext counter of the 2-wavefront FFT kernel
    int wg_slots = 1024;                  // 4 workgroups per CU
    // arith 2 (FHS_ARITH_F64_FFT_MB2): two key bits per external product (fftmb_kernels.hip); needs the pair key
    double *d_bsk_mb = nullptr;       // [371][K1,K2,K3][4][1024] complex
    double *d_bsk_ntt_mb = nullptr;   // arith 3 (FHS_ARITH_EXACT_NTT_MB2): [371][K1,K2,K3][4][2 primes][2048] residues
    const double *d_ntt_mono = nullptr;   // [2][4096] inside d_tables
    int load_multibit_key(const uint64_t *bsk_mb2);   // [371][K1,K2,K3][4][2048] u64 standard domain (fhs_client_bsk_mb2)
    int fft4_max_batch = 512;
    size_t launch_chunk[4] = {0, 0, 0, 0};   // per arithmetic: ciphertexts per blind-rotation launch (0 = whole batch)         // batches up to this size use the 4-wavefront kernel (lower latency)
    int set_arithmetic(int mode);
    // keyswitch of a dense batch into ks_buf (timed as kernel kind 1); ks_buf must hold B rows
    int keyswitch(const uint64_t *d_in, size_t B, hipStream_t s);
    // blind rotation in the selected arithmetic (timed as kernel kind 0)
    int blind_rotate(const uint64_t *d_ks, const uint32_t *d_lut_idx, const uint64_t *d_luts, uint64_t *d_out,
                     uint64_t *const *d_out_ptrs, size_t B, hipStream_t s, uint64_t *const *d_body_ptrs = nullptr);

    // multi-GPU exchange (fhs_dist_init): RCCL communicator of this context, or a host transport
    Dist dist;
    DevBuf xchg_send, xchg_recv;     // char exchange of the sharded string ops / level slices of the level-parallel flush

    // scratch
    DevBuf dig_buf;                  // keyswitch digits of the current batch
    DevBuf ks_buf, ms_buf, in_buf, out_buf, lutidx_buf, luts_buf, tab_buf;
    KernelTimer timer;

    int init(int device_id);
    void shutdown();
    int fail(int code, const std::string &msg) { err = msg; return code; }
    int hip_fail(hipError_t e, const char *what);

    int load_server_key(const uint64_t *bsk, const uint64_t *ksk);
    int build_fft_key();
    // all device pointers; enqueues KS+MS then blind rotation on `s`
    int pbs_batch_device(const uint64_t *d_in, const uint32_t *d_lut_idx, const uint64_t *d_luts,
                         uint64_t *d_out, size_t B, hipStream_t s);
    int pbs_batch_shifted_host(const uint64_t *in, const uint32_t *lut_idx, const uint64_t *luts, size_t n_luts,
                               const uint32_t *shifts, size_t S, uint64_t *out, size_t B);
    int pbs_batch_host(const uint64_t *in, const uint32_t *lut_idx, const uint64_t *luts, size_t n_luts,
                       uint64_t *out, size_t B);
    int ks_ms_batch_host(const uint64_t *in, uint32_t *ms_out, size_t B);
    int blind_rotate_host(const uint64_t *ks, const uint32_t *lut_idx, const uint64_t *luts, size_t n_luts,
                          uint64_t *out, size_t B);
};

}  // namespace fhs
