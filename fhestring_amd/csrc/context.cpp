#include "context.h"

#include <algorithm>
#include <cstdlib>
#include <thread>

namespace fhs {

#define HIP_TRY(expr, what)                                \
    do {                                                   \
        hipError_t e__ = (expr);                           \
        if (e__ != hipSuccess) return hip_fail(e__, what); \
    } while (0)

hipError_t DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    size_t want = std::max(bytes, (size_t)1 << 20);
    hipError_t e = hipMalloc(&ptr, want);
    if (e == hipSuccess) cap = want;
    return e;
}
void DevBuf::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
}

hipEvent_t KernelTimer::get() {
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
void KernelTimer::begin(int kind, uint64_t u, hipStream_t s) {
    if (!enabled) return;
    if (pending.size() > 4096) resolve();
    Pending p{get(), get(), kind, u, 1};
    (void)hipEventRecord(p.e0, s);
    pending.push_back(p);
}
void KernelTimer::end(hipStream_t s, uint32_t kernel_launches) {
    if (!enabled || pending.empty()) return;
    pending.back().launches = kernel_launches ? kernel_launches : 1;
    (void)hipEventRecord(pending.back().e1, s);
}
void KernelTimer::resolve() {
    for (auto &p : pending) {
        (void)hipEventSynchronize(p.e1);
        float t = 0;
        if (hipEventElapsedTime(&t, p.e0, p.e1) == hipSuccess) {
            ms[p.kind] += t;
            n[p.kind] += p.launches;
            units[p.kind] += p.units;
        }
        pool.push_back(p.e0);
        pool.push_back(p.e1);
    }
    pending.clear();
}
void KernelTimer::reset() {
    resolve();
    for (int k = 0; k < 3; k++) { ms[k] = 0; n[k] = 0; units[k] = 0; }
}
void KernelTimer::destroy() {
    resolve();
    for (auto e : pool) (void)hipEventDestroy(e);
    pool.clear();
}

int Context::hip_fail(hipError_t e, const char *what) {
    err = std::string(what) + ": " + hipGetErrorString(e);
    return -2;
}

int Context::init(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(-2, "no HIP device visible: this backend has no CPU fallback");
    if (device_id < 0 || device_id >= n) return fail(-1, "device_id out of range");
    device = device_id;
    HIP_TRY(hipSetDevice(device), "hipSetDevice");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(-2, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    HIP_TRY(prepare_device_for_kernels(), "hipFuncSetAttribute (dynamic LDS)");
    HIP_TRY(prepare_device_for_fft4(), "hipFuncSetAttribute (dynamic LDS, 4-wavefront kernel)");
    HIP_TRY(prepare_device_for_keyswitch(), "hipFuncSetAttribute (dynamic LDS, wide keyswitch kernel)");
    // FHS_STREAM_CU_MASK=<hex word>[,<hex word>...] (32 CUs per word, lowest CUs first): this context's stream only
    // runs on the compute units of the mask (hipExtStreamCreateWithCUMask) and the persistent kernels size their grid for
    // them.  An experiment switch (tools/exp_cumask.py, DESIGN.md section 5d: a narrow dependency level of one request on
    // a slice of the chip beside another request's wide level), not a tuning knob of the product.
    int cus = prop.multiProcessorCount;
    if (const char *m = std::getenv("FHS_STREAM_CU_MASK")) {
        std::vector<uint32_t> mask;
        int bits = 0;
        for (const char *q = m; *q;) {
            char *end = nullptr;
            const unsigned long w = std::strtoul(q, &end, 16);
            if (end == q) break;
            mask.push_back((uint32_t)w);
            bits += __builtin_popcount((uint32_t)w);
            q = *end == ',' ? end + 1 : end;
            if (*end != ',') break;
        }
        if (mask.empty() || bits == 0 || bits > cus) return fail(-1, "FHS_STREAM_CU_MASK: expected hex words selecting 1.." + std::to_string(cus) + " CUs");
        HIP_TRY(hipExtStreamCreateWithCUMask(&stream, (uint32_t)mask.size(), mask.data()), "hipExtStreamCreateWithCUMask");
        cus = bits;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
    }
    wg_slots = 4 * cus;
    // two-bit f64 kernel: one round of resident workgroups per launch (its 73 MB key only stays inside the L2 window
    // while the workgroups walk it together; consecutive launches overlap at the seams): 202 k instead of 183 k PBS/s at
    // 4096 rows, 214 k instead of 152 k at 14 336.  The classic f64 kernel (48 MB key) does not care below 4096 rows.
    launch_chunk[2] = (size_t)wg_slots;
    // classic f64 kernel (round 5): one round per launch as well.  Time-neutral (default bench 134.5 k PBS/s plain, 135.5 k
    // chunked; 3968-wide launches 30.12 ms plain, 30.33 ms chunked: profiles/r05_chunk_ab.txt) but every launch restarts
    // the key walk of all workgroups together: L2 hit rate 95 -> 98.5 %, fabric-side traffic 6.3 -> 2.4 GB per launch
    // group -- traffic that eight processes on one node would otherwise multiply by eight.  A remainder below a quarter of
    // a round rides in the last launch instead of becoming one of its own (blind_rotate()).
    launch_chunk[1] = (size_t)wg_slots;
    HIP_TRY(hipMalloc(&d_work_counter, 64), "hipMalloc counter");
    return 0;
}

void Context::shutdown() {
    if (stream) (void)hipStreamSynchronize(stream);
    dist.shutdown();
    xchg_send.release();
    xchg_recv.release();
    timer.destroy();
    ms_buf.release();
    ks_buf.release();
    in_buf.release();
    out_buf.release();
    lutidx_buf.release();
    tab_buf.release();
    luts_buf.release();
    if (d_ksk_planes) (void)hipFree(d_ksk_planes);
    d_ksk_planes = nullptr;
    dig_buf.release();
    if (d_bsk_ntt) (void)hipFree(d_bsk_ntt);
    if (d_tables) (void)hipFree(d_tables);
    if (d_work_counter) (void)hipFree(d_work_counter);
    d_work_counter = nullptr;
    if (d_bsk_fft) (void)hipFree(d_bsk_fft);
    if (d_bsk_std) (void)hipFree(d_bsk_std);
    if (d_bsk_mb) (void)hipFree(d_bsk_mb);
    d_bsk_mb = nullptr;
    if (d_bsk_ntt_mb) (void)hipFree(d_bsk_ntt_mb);
    d_bsk_ntt_mb = nullptr;
    if (d_fft_tables) (void)hipFree(d_fft_tables);
    d_bsk_fft = nullptr;
    d_bsk_std = nullptr;
    d_fft_tables = nullptr;
    d_bsk_ntt = nullptr;
    d_tables = nullptr;
    if (stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
}

int Context::load_server_key(const uint64_t *bsk, const uint64_t *ksk) {
    if (!bsk || !ksk) return fail(-1, "null key pointer");
    HIP_TRY(hipSetDevice(device), "hipSetDevice");
    // a pair key belongs to the server key it was generated with: a new server key invalidates it
    if (d_bsk_mb) { (void)hipFree(d_bsk_mb); d_bsk_mb = nullptr; }
    if (d_bsk_ntt_mb) { (void)hipFree(d_bsk_ntt_mb); d_bsk_ntt_mb = nullptr; }
    const size_t ksk_bytes = (size_t)BIG_N * KS_LEVEL * SMALL_CT * sizeof(uint64_t);
    const size_t bsk_ntt_doubles = (size_t)LWE_N * 4 * 2 * POLY_N;
    if (!d_bsk_ntt) HIP_TRY(hipMalloc(&d_bsk_ntt, bsk_ntt_doubles * sizeof(double)), "hipMalloc bsk");
    {   // the KSK is only kept as byte planes in MFMA fragment order (ks_kernels.hip)
        uint64_t *d_ksk = nullptr;
        HIP_TRY(hipMalloc(&d_ksk, ksk_bytes), "hipMalloc ksk staging");
        hipError_t e = hipMemcpy(d_ksk, ksk, ksk_bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess && !d_ksk_planes) e = hipMalloc(&d_ksk_planes, ks_planes_bytes());
        if (e == hipSuccess) e = launch_ksk_to_planes(d_ksk, d_ksk_planes, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_ksk);
        HIP_TRY(e, "ksk -> byte planes");
    }
    {
        std::vector<double> host(bsk_ntt_doubles);
        unsigned hc = std::thread::hardware_concurrency();
        convert_bsk_to_ntt(bsk, host.data(), (int)std::min(32u, std::max(1u, hc)));
        HIP_TRY(hipMemcpy(d_bsk_ntt, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice),
                "copy bsk");
    }
    HostNttTables ht;
    build_ntt_tables(ht);
    const size_t n_tab = ht.fwd_uni.size() + ht.fwd_lane.size() + ht.inv_uni.size() + ht.inv_lane.size() + ht.mono.size();
    if (!d_tables) HIP_TRY(hipMalloc(&d_tables, n_tab * sizeof(double)), "hipMalloc tables");
    double *pd = d_tables;
    auto up = [&](const std::vector<double> &v, const double *&slot) -> hipError_t {
        slot = pd;
        hipError_t e = hipMemcpy(pd, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice);
        pd += v.size();
        return e;
    };
    HIP_TRY(up(ht.fwd_uni, tw.fwd_uni), "copy tables");
    HIP_TRY(up(ht.fwd_lane, tw.fwd_lane), "copy tables");
    HIP_TRY(up(ht.inv_uni, tw.inv_uni), "copy tables");
    HIP_TRY(up(ht.inv_lane, tw.inv_lane), "copy tables");
    HIP_TRY(up(ht.mono, d_ntt_mono), "copy tables");
    crt_c = ht.crt_c;
    {   // the kernel's baked-in uniform twiddles must equal the exactly derived ones
        std::vector<double> fu(64), iu(128);
        double c = 0;
        HIP_TRY(read_device_ntt_consts(fu.data(), iu.data(), &c), "read device constants");
        if (fu != ht.fwd_uni || iu != ht.inv_uni || c != ht.crt_c)
            return fail(-3, "ntt_consts.inc does not match the derived twiddle tables (regenerate it)");
    }
    {   // the standard-domain key stays on the device (48.6 MB of 288 GB): the Fourier-domain key is built from it with the
        // device's own forward transform -- now if the f64 arithmetic is selected, otherwise the first time it is
        const size_t n = (size_t)LWE_N * 4 * POLY_N;
        if (!d_bsk_std) HIP_TRY(hipMalloc(&d_bsk_std, n * sizeof(uint64_t)), "hipMalloc bsk (standard domain)");
        HIP_TRY(hipMemcpy(d_bsk_std, bsk, n * sizeof(uint64_t), hipMemcpyHostToDevice), "copy bsk (standard domain)");
    }
    if (d_bsk_fft) {          // a new key invalidates the Fourier-domain form of the old one
        (void)hipFree(d_bsk_fft);
        d_bsk_fft = nullptr;
    }
    if (arith == 1 || arith == 2)
        if (int rc = build_fft_key()) return rc;
    key_loaded = true;
    return 0;
}

// Fourier-domain bootstrapping key of the f64 arithmetics from the retained standard-domain key (load_server_key, or the
// first fhs_set_arithmetic(FHS_ARITH_F64_FFT) after a key was loaded under the exact arithmetic: either order works).
int Context::build_fft_key() {
    if (!d_bsk_std) return fail(-3, "server key not loaded");
    HIP_TRY(hipSetDevice(device), "hipSetDevice");
    HostFftTables ft;
    build_fft_tables(ft);
    {   // the kernel's literal twiddles must equal the libm-derived ones
        double wr[16], wi[16], ur[3], ui[3];
        fft_uniform_consts(wr, wi, ur, ui);
        bool ok = true;
        for (int k = 1; k < 16; k++) ok = ok && wr[k] == ft.w_re[k] && wi[k] == ft.w_im[k];
        for (int k = 0; k < 3; k++) ok = ok && ur[k] == ft.u_re[k] && ui[k] == ft.u_im[k];
        if (!ok) return fail(-3, "fft_consts.inc does not match the libm-derived twiddles (regenerate it)");
    }
    std::vector<double> flat(ft.lanetab);
    flat.insert(flat.end(), ft.weff.begin(), ft.weff.end());
    flat.insert(flat.end(), ft.mono.begin(), ft.mono.end());
    flat.insert(flat.end(), ft.r16.begin(), ft.r16.end());
    if (!d_fft_tables) HIP_TRY(hipMalloc(&d_fft_tables, flat.size() * sizeof(double)), "hipMalloc fft tables");
    HIP_TRY(hipMemcpy(d_fft_tables, flat.data(), flat.size() * sizeof(double), hipMemcpyHostToDevice), "copy fft tables");
    const size_t n = (size_t)LWE_N * 4 * POLY_N;   // 1024 complex (2048 doubles) per polynomial
    if (!d_bsk_fft) HIP_TRY(hipMalloc(&d_bsk_fft, n * sizeof(double)), "hipMalloc bsk fft");
    hipError_t e = launch_bsk_to_fft(d_bsk_std, d_bsk_fft, d_fft_tables, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        (void)hipFree(d_bsk_fft);
        d_bsk_fft = nullptr;
    }
    HIP_TRY(e, "bsk -> Fourier domain");
    return 0;
}

int Context::keyswitch(const uint64_t *d_in, size_t B, hipStream_t s) {
    if (dig_buf.cap < ks_digits_bytes((int)B)) {
        HIP_TRY(hipStreamSynchronize(s), "sync");
        HIP_TRY(dig_buf.reserve(ks_digits_bytes((int)B)), "hipMalloc digits");
    }
    timer.begin(1, B, s);
    hipError_t e = launch_keyswitch_mfma(d_in, d_ksk_planes, dig_buf.as<int8_t>(), ks_buf.as<uint64_t>(), (int)B, s,
                                         wg_slots / 4);
    timer.end(s);
    if (e != hipSuccess) return hip_fail(e, "keyswitch launch");
    return 0;
}

int Context::set_arithmetic(int mode) {
    if (mode < 0 || mode > 3) return fail(-1, "unknown arithmetic mode");
    if ((mode == 1 || mode == 2) && key_loaded && !d_bsk_fft)      // the key was loaded under the exact arithmetic: build
        if (int rc = build_fft_key()) return rc;                   // its Fourier-domain form now (either order works)
    if (mode == 2 && key_loaded && !d_bsk_mb)
        return fail(-3, "the two-bits-per-product arithmetic needs the pair key in the Fourier domain: call "
                        "fhs_load_multibit_key while arithmetic 1 (f64 FFT) is selected, then fhs_set_arithmetic 2");
    if (mode == 3 && key_loaded && !d_bsk_ntt_mb)
        return fail(-3, "the exact two-bits-per-product arithmetic needs the pair key converted to residues: call "
                        "fhs_load_multibit_key while the EXACT arithmetic (fhs_set_arithmetic 0) is selected, then "
                        "fhs_set_arithmetic 3");
    arith = mode;
    return 0;
}

// Pair key of the two-bits-per-product blind rotation: transformed with the device's own forward transform like the
// classic key (fftmb_kernels.hip).
int Context::load_multibit_key(const uint64_t *bsk_mb2) {
    if (!bsk_mb2) return fail(-1, "null key pointer");
    if (!key_loaded) return fail(-3, "load the server key before the pair key");
    HIP_TRY(hipSetDevice(device), "hipSetDevice");
    if (arith == 0 || arith == 3) {
        // exact arithmetic: residues of the pair key modulo the two NTT primes, on the 57-bit torus grid (nttmb_kernels.hip)
        if (!ntt_slot_roots_are_bitreversed()) return fail(-3, "internal: NTT slot order is not bit-reversed");
        const size_t n_d = (size_t)(LWE_N / 2) * 3 * 4 * 2 * POLY_N;
        std::vector<double> host(n_d);
        unsigned hc = std::thread::hardware_concurrency();
        convert_bsk_to_ntt(bsk_mb2, host.data(), (int)std::min(32u, std::max(1u, hc)), (LWE_N / 2) * 3, 7);
        if (!d_bsk_ntt_mb) HIP_TRY(hipMalloc(&d_bsk_ntt_mb, n_d * sizeof(double)), "hipMalloc pair key (NTT)");
        HIP_TRY(hipMemcpy(d_bsk_ntt_mb, host.data(), n_d * sizeof(double), hipMemcpyHostToDevice), "copy pair key");
        HIP_TRY(prepare_device_for_ntt_mb2(), "kernel attributes");
        return 0;
    }
    if (!d_fft_tables)
        return fail(-3, "load the server key in an f64-FFT arithmetic (fhs_set_arithmetic 1 or 2) before the pair key");
    const int n_polys = (LWE_N / 2) * 3 * 4;
    const size_t n = (size_t)n_polys * POLY_N;            // u64 in, doubles out (1024 complex per polynomial)
    uint64_t *d_std = nullptr;
    HIP_TRY(hipMalloc(&d_std, n * sizeof(uint64_t)), "hipMalloc pair key staging");
    hipError_t e = hipSuccess;
    if (!d_bsk_mb) e = hipMalloc(&d_bsk_mb, n * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d_std, bsk_mb2, n * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_bsk_to_fft(d_std, d_bsk_mb, d_fft_tables, stream, n_polys);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_std);
    HIP_TRY(e, "pair key -> Fourier domain");
    return 0;
}

int Context::blind_rotate(const uint64_t *d_ks, const uint32_t *d_lut_idx, const uint64_t *d_luts, uint64_t *d_out,
                          uint64_t *const *d_out_ptrs, size_t B, hipStream_t s, uint64_t *const *d_body_ptrs) {
    hipError_t e = hipSuccess;
    const bool four = arith == 1 && B <= (size_t)fft4_max_batch;
    if (arith == 2 && !d_bsk_mb) return fail(-3, "pair key not loaded (fhs_load_multibit_key)");
    if (arith == 3 && !d_bsk_ntt_mb) return fail(-3, "pair key not loaded in the exact arithmetic (fhs_load_multibit_key)");
    if (arith == 1 && !d_bsk_fft) return fail(-3, "Fourier-domain key not loaded");
    timer.begin(four ? 2 : 0, B, s);
    // A launch is cut into chunks of launch_chunk[arith] ciphertexts (0 = whole batch): every chunk starts all workgroups on
    // the first key element together again.  The kernels whose key does not fit the L2 window of a drifting launch need
    // that (two-bit f64 kernel: 185 k PBS/s in launches of 3 072 - 4 096 rows, 144 k in one launch of 13 400).
    // A remainder of less than a quarter of a chunk does not become a launch of its own (8194 rows = 7 launches of 1024 and
    // one of 1026, not eight and a launch of 2 that costs a whole bootstrap alone).  The narrow-level kernel is never cut.
    const size_t chunk = (launch_chunk[arith] && !four) ? launch_chunk[arith] : B;
    uint32_t n_launches = 0;
    for (size_t off = 0, n = 0; off < B && e == hipSuccess; off += n) {
        n = std::min(chunk, B - off);
        if (B - off - n < chunk / 4) n = B - off;
        n_launches++;
        const uint64_t *ks = d_ks + off * SMALL_CT;
        const uint32_t *li = d_lut_idx + off;
        uint64_t *out = d_out ? d_out + off * BIG_CT : nullptr;
        uint64_t *const *outp = d_out_ptrs ? d_out_ptrs + off : nullptr;
        uint64_t *const *bodyp = d_body_ptrs ? d_body_ptrs + off : nullptr;   // rotation sharing (engine.cpp)
        if (arith == 3) {
            BlindRotateNttMb2Params p{};
            p.ks = ks; p.lut_idx = li; p.luts = d_luts;
            p.bsk_ntt_mb = d_bsk_ntt_mb; p.tw = tw; p.crt_c = crt_c; p.mono = d_ntt_mono;
            p.out = out; p.out_ptrs = outp; p.body_ptrs = bodyp; p.B = (int)n;
            e = launch_blind_rotate_ntt_mb2(p, s);
        } else if (arith == 2) {
            BlindRotateMb2Params p{};
            p.ks = ks; p.lut_idx = li; p.luts = d_luts;
            p.bsk_mb = d_bsk_mb;
            p.lanetab = d_fft_tables;
            p.mono = d_fft_tables + 12 * 64 + 2 * 1024;
            p.r16 = p.mono + 2 * 4096;
            p.work_counter = d_work_counter;
            p.slots = wg_slots;
            p.out = out; p.out_ptrs = outp; p.body_ptrs = bodyp; p.B = (int)n;
            e = launch_blind_rotate_mb2(p, s);
        } else if (arith == 1) {
            BlindRotateFftParams p{};
            p.ks = ks; p.lut_idx = li; p.luts = d_luts;
            p.bsk_fft = d_bsk_fft;
            p.lanetab = d_fft_tables;
            p.weff = d_fft_tables + 12 * 64;
            p.work_counter = d_work_counter;
            p.slots = wg_slots;
            p.out = out; p.out_ptrs = outp; p.body_ptrs = bodyp; p.B = (int)n;
            e = four ? launch_blind_rotate_fft4(p, s) : launch_blind_rotate_fft(p, s);
        } else {
            BlindRotateParams p{};
            p.ks = ks; p.lut_idx = li; p.luts = d_luts;
            p.bsk_ntt = d_bsk_ntt; p.tw = tw; p.crt_c = crt_c;
            p.out = out; p.out_ptrs = outp; p.body_ptrs = bodyp; p.B = (int)n;
            e = launch_blind_rotate(p, s);
        }
    }
    timer.end(s, n_launches);
    if (e != hipSuccess) return hip_fail(e, "blind_rotate launch");
    return 0;
}

int Context::pbs_batch_device(const uint64_t *d_in, const uint32_t *d_lut_idx, const uint64_t *d_luts,
                              uint64_t *d_out, size_t B, hipStream_t s) {
    if (!key_loaded) return fail(-3, "server key not loaded");
    if (B == 0) return 0;
    if (B > (size_t)1 << 24) return fail(-1, "batch too large");
    HIP_TRY(ks_buf.reserve(B * SMALL_CT * sizeof(uint64_t)), "hipMalloc ks");
    if (int rc = keyswitch(d_in, B, s)) return rc;
    if (int rc = blind_rotate(ks_buf.as<uint64_t>(), d_lut_idx, d_luts, d_out, nullptr, B, s)) return rc;
    return 0;
}

int Context::pbs_batch_host(const uint64_t *in, const uint32_t *lut_idx, const uint64_t *luts, size_t n_luts,
                            uint64_t *out, size_t B) {
    if (!in || !lut_idx || !luts || !out) return fail(-1, "null pointer");
    if (B == 0) return 0;
    for (size_t b = 0; b < B; b++)
        if (lut_idx[b] >= n_luts) return fail(-1, "lut_idx out of range");
    HIP_TRY(hipSetDevice(device), "hipSetDevice");
    HIP_TRY(in_buf.reserve(B * BIG_CT * 8), "hipMalloc");
    HIP_TRY(out_buf.reserve(B * BIG_CT * 8), "hipMalloc");
    HIP_TRY(lutidx_buf.reserve(B * 4), "hipMalloc");
    HIP_TRY(luts_buf.reserve(n_luts * POLY_N * 8), "hipMalloc");
    HIP_TRY(hipMemcpyAsync(in_buf.ptr, in, B * BIG_CT * 8, hipMemcpyHostToDevice, stream), "H2D");
    HIP_TRY(hipMemcpyAsync(lutidx_buf.ptr, lut_idx, B * 4, hipMemcpyHostToDevice, stream), "H2D");
    HIP_TRY(hipMemcpyAsync(luts_buf.ptr, luts, n_luts * POLY_N * 8, hipMemcpyHostToDevice, stream), "H2D");
    int rc = pbs_batch_device(in_buf.as<uint64_t>(), lutidx_buf.as<uint32_t>(), luts_buf.as<uint64_t>(),
                              out_buf.as<uint64_t>(), B, stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, out_buf.ptr, B * BIG_CT * 8, hipMemcpyDeviceToHost, stream), "D2H");
    HIP_TRY(hipStreamSynchronize(stream), "sync");
    return 0;
}

// Rotation sharing at the raw boundary: ONE keyswitch + blind rotation per input, S sample extractions each --
// out[b][s] = what a bootstrap of (in[b] + shifts[b][s] * Delta) with LUT lut_idx[b] yields (shifts in message units,
// 0..31).  The engine does the same for rows of a level that differ only in a trivial constant (engine.cpp plan_job).
int Context::pbs_batch_shifted_host(const uint64_t *in, const uint32_t *lut_idx, const uint64_t *luts, size_t n_luts,
                                    const uint32_t *shifts, size_t S, uint64_t *out, size_t B) {
    if (!key_loaded) return fail(-3, "server key not loaded");
    if (!in || !lut_idx || !luts || !shifts || !out) return fail(-1, "null pointer");
    if (B == 0 || S == 0) return 0;
    if (B * S > (size_t)1 << 22) return fail(-1, "batch too large");
    for (size_t b = 0; b < B; b++)
        if (lut_idx[b] >= n_luts) return fail(-1, "lut_idx out of range");
    HIP_TRY(hipSetDevice(device), "hipSetDevice");
    // device layout: [B rotation outputs | B body polynomials (2049-word rows) | B * S extractions]
    HIP_TRY(in_buf.reserve(B * BIG_CT * 8), "hipMalloc");
    HIP_TRY(out_buf.reserve((2 * B + B * S) * BIG_CT * 8), "hipMalloc");
    HIP_TRY(lutidx_buf.reserve(B * 4), "hipMalloc");
    HIP_TRY(luts_buf.reserve(n_luts * POLY_N * 8), "hipMalloc");
    HIP_TRY(ks_buf.reserve(B * SMALL_CT * sizeof(uint64_t)), "hipMalloc ks");
    uint64_t *d_rot = out_buf.as<uint64_t>(), *d_body = d_rot + B * BIG_CT, *d_ext = d_body + B * BIG_CT;
    std::vector<uint64_t *> ptrs(2 * B);
    std::vector<ExtractDesc> descs(B * S);
    for (size_t b = 0; b < B; b++) {
        ptrs[b] = d_rot + b * BIG_CT;
        ptrs[B + b] = d_body + b * BIG_CT;
        for (size_t k = 0; k < S; k++)
            descs[b * S + k] = ExtractDesc{ptrs[b], ptrs[B + b], d_ext + (b * S + k) * BIG_CT, 128u * (shifts[b * S + k] & 31u), 0};
    }
    DevBuf &tab = tab_buf;
    const size_t tab_bytes = ptrs.size() * sizeof(uint64_t *) + descs.size() * sizeof(ExtractDesc);
    HIP_TRY(tab.reserve(tab_bytes), "hipMalloc");
    HIP_TRY(hipMemcpyAsync(tab.ptr, ptrs.data(), ptrs.size() * sizeof(uint64_t *), hipMemcpyHostToDevice, stream), "H2D");
    HIP_TRY(hipMemcpyAsync(tab.as<uint8_t>() + ptrs.size() * sizeof(uint64_t *), descs.data(), descs.size() * sizeof(ExtractDesc),
                           hipMemcpyHostToDevice, stream), "H2D");
    HIP_TRY(hipMemcpyAsync(in_buf.ptr, in, B * BIG_CT * 8, hipMemcpyHostToDevice, stream), "H2D");
    HIP_TRY(hipMemcpyAsync(lutidx_buf.ptr, lut_idx, B * 4, hipMemcpyHostToDevice, stream), "H2D");
    HIP_TRY(hipMemcpyAsync(luts_buf.ptr, luts, n_luts * POLY_N * 8, hipMemcpyHostToDevice, stream), "H2D");
    HIP_TRY(hipStreamSynchronize(stream), "sync");                       // the host vectors above are pageable
    if (int rc = keyswitch(in_buf.as<uint64_t>(), B, stream)) return rc;
    uint64_t *const *d_ptrs = tab.as<uint64_t *>();
    if (int rc = blind_rotate(ks_buf.as<uint64_t>(), lutidx_buf.as<uint32_t>(), luts_buf.as<uint64_t>(), nullptr, d_ptrs, B, stream,
                              d_ptrs + B))
        return rc;
    HIP_TRY(launch_extract_shift(reinterpret_cast<const ExtractDesc *>(tab.as<uint8_t>() + ptrs.size() * sizeof(uint64_t *)),
                                 (int)(B * S), stream), "extract launch");
    HIP_TRY(hipMemcpyAsync(out, d_ext, B * S * BIG_CT * 8, hipMemcpyDeviceToHost, stream), "D2H");
    HIP_TRY(hipStreamSynchronize(stream), "sync");
    return 0;
}

// kernel-level tests: blind rotation + sample extraction only, from given keyswitched LWEs (u64 torus, mod-switched in
// the kernel like always), in the selected arithmetic
int Context::blind_rotate_host(const uint64_t *ks, const uint32_t *lut_idx, const uint64_t *luts, size_t n_luts,
                               uint64_t *out, size_t B) {
    if (!key_loaded) return fail(-3, "server key not loaded");
    if (!ks || !lut_idx || !luts || !out) return fail(-1, "null pointer");
    if (B == 0) return 0;
    for (size_t b = 0; b < B; b++)
        if (lut_idx[b] >= n_luts) return fail(-1, "lut_idx out of range");
    HIP_TRY(hipSetDevice(device), "hipSetDevice");
    HIP_TRY(ks_buf.reserve(B * SMALL_CT * 8), "hipMalloc");
    HIP_TRY(out_buf.reserve(B * BIG_CT * 8), "hipMalloc");
    HIP_TRY(lutidx_buf.reserve(B * 4), "hipMalloc");
    HIP_TRY(luts_buf.reserve(n_luts * POLY_N * 8), "hipMalloc");
    HIP_TRY(hipMemcpyAsync(ks_buf.ptr, ks, B * SMALL_CT * 8, hipMemcpyHostToDevice, stream), "H2D");
    HIP_TRY(hipMemcpyAsync(lutidx_buf.ptr, lut_idx, B * 4, hipMemcpyHostToDevice, stream), "H2D");
    HIP_TRY(hipMemcpyAsync(luts_buf.ptr, luts, n_luts * POLY_N * 8, hipMemcpyHostToDevice, stream), "H2D");
    if (int rc = blind_rotate(ks_buf.as<uint64_t>(), lutidx_buf.as<uint32_t>(), luts_buf.as<uint64_t>(),
                              out_buf.as<uint64_t>(), nullptr, B, stream))
        return rc;
    HIP_TRY(hipMemcpyAsync(out, out_buf.ptr, B * BIG_CT * 8, hipMemcpyDeviceToHost, stream), "D2H");
    HIP_TRY(hipStreamSynchronize(stream), "sync");
    return 0;
}

int Context::ks_ms_batch_host(const uint64_t *in, uint32_t *ms_out, size_t B) {
    if (!key_loaded) return fail(-3, "server key not loaded");
    if (!in || !ms_out) return fail(-1, "null pointer");
    if (B == 0) return 0;
    HIP_TRY(hipSetDevice(device), "hipSetDevice");
    HIP_TRY(in_buf.reserve(B * BIG_CT * 8), "hipMalloc");
    HIP_TRY(ms_buf.reserve(B * SMALL_CT * 4), "hipMalloc");
    HIP_TRY(ks_buf.reserve(B * SMALL_CT * 8), "hipMalloc");
    HIP_TRY(hipMemcpyAsync(in_buf.ptr, in, B * BIG_CT * 8, hipMemcpyHostToDevice, stream), "H2D");
    if (int rc = keyswitch(in_buf.as<uint64_t>(), B, stream)) return rc;
    HIP_TRY(launch_modswitch(ks_buf.as<uint64_t>(), ms_buf.as<uint32_t>(), (int)B, stream), "modswitch launch");
    HIP_TRY(hipMemcpyAsync(ms_out, ms_buf.ptr, B * SMALL_CT * 4, hipMemcpyDeviceToHost, stream), "D2H");
    HIP_TRY(hipStreamSynchronize(stream), "sync");
    return 0;
}

}  // namespace fhs
