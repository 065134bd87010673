/*
 * fhestring_hip.h -- C ABI of the MI355X-native TFHE backend that sits under
 * the FheString server-key operations of MakisChristou/fhestring.
 *
 * The reference has no FFI: its seam is the set of `tfhe::integer` calls made
 * from src/ciphertext/fheasciichar.rs and src/client_key.rs.
 * Every entry point below names the reference interface it replaces.  Plain
 * pointers and sizes only; every function returns 0 on success or a negative
 * error code (text via fhs_last_error); nothing unwinds across the boundary.
 *
 * Ciphertext layout (unchanged from the reference's types):
 *   block   = big LWE ciphertext, 2049 x u64 (2048 mask + body), 16 392 B
 *   char    = FheAsciiChar = 4 blocks, little-endian 2-bit digits, 65 568 B
 *             (src/ciphertext/fheasciichar.rs:8-10)
 *   string  = contiguous array of chars (src/ciphertext/fhestring.rs:6-9)
 * Keys: PARAM_MESSAGE_2_CARRY_2_KS_PBS (src/main.rs:3,43)
 *   bsk[742][2][2][2048] u64  standard-domain GGSW rows (row 0 mask, row 1 body)
 *   ksk[2048][5][743]    u64
 */
#ifndef FHESTRING_HIP_H
#define FHESTRING_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FHS_LWE_N 742
#define FHS_POLY_N 2048
#define FHS_BIG_CT 2049          /* u64 words per block */
#define FHS_SMALL_CT 743
#define FHS_BLOCKS_PER_CHAR 4    /* MAX_BLOCKS, src/main.rs:23 */
#define FHS_CHAR_WORDS (4 * 2049)
#define FHS_BSK_WORDS ((size_t)742 * 4 * 2048)
#define FHS_KSK_WORDS ((size_t)2048 * 5 * 743)
#define FHS_MAX_FIND_LENGTH 255  /* src/main.rs:20 */
#define FHS_MAX_REPETITIONS 16   /* src/main.rs:17 */

#define FHS_OK 0
#define FHS_ERR_ARG (-1)
#define FHS_ERR_HIP (-2)
#define FHS_ERR_STATE (-3)
#define FHS_ERR_LIMIT (-4)       /* the reference's panic!("Maximum supported size for find reached") */

typedef struct fhs_ctx fhs_ctx;
typedef struct fhs_client fhs_client;
typedef uint64_t fhs_char_t;     /* opaque handle of one lazily evaluated FheAsciiChar */

/* ---- context / server key -------------------------------------------------
 * replaces: tfhe::integer::ServerKey held by MyServerKey (src/server_key/mod.rs:13-16) */
int fhs_ctx_create(int device_id, fhs_ctx **out);
/* Planner context: no device, no key.  Every fhs_* op records its DAG exactly as on a real context and fhs_flush
 * levelises it, so the statistics (PBS count, dependency levels and their widths, noise bookkeeping) are those of the
 * real run -- but NOTHING is computed: fhs_download / exports / raw PBS entry points fail with FHS_ERR_STATE.  Host logic
 * only (capacity planning, the CPU test-suite's checks of DAG shape and noise budget); not a CPU fallback. */
int fhs_ctx_create_planner(fhs_ctx **out);
void fhs_ctx_destroy(fhs_ctx *ctx);
const char *fhs_last_error(const fhs_ctx *ctx);
/* Copies the key to the device; the BSK is rounded to the 58-bit torus grid and
 * transformed to the device NTT representation (DESIGN.md "BSK precision"). */
int fhs_load_server_key(fhs_ctx *ctx, const uint64_t *bsk, const uint64_t *ksk);
/* Arithmetic of the negacyclic products inside blind rotation.  EXACT_NTT (default): exact integer
 * arithmetic over two 47-bit NTT primes.  F64_FFT: folded f64 complex FFT, the algorithm class of the
 * reference's CPU engine (tfhe 0.5.2 + concrete-fft 0.4.0, Cargo.lock:168-179) - ~3x faster, approximate
 * at the 2^-53 relative level (far below the scheme's noise) and deterministic.  fhs_set_arithmetic and
 * fhs_load_server_key may come in either order: the standard-domain key stays on the device (48.6 MB) and its
 * Fourier-domain form is built at the key load when F64_FFT is already selected, otherwise the first time it is
 * selected (round 5; before that a key loaded first left a drop-in host on the 3.7x slower exact path).  Switching
 * back and forth is always allowed. */
#define FHS_ARITH_EXACT_NTT 0
#define FHS_ARITH_F64_FFT 1
/* F64_FFT_MB2: the f64 FFT arithmetic with TWO LWE key bits per GGSW x GLWE external product (371 products per bootstrap
 * instead of 742; the "multi-bit" blind rotation of Joye-Paillier / tfhe-rs' GPU backend at group size 2, here on the
 * reference's own parameter set: dimensions, bases, noise distributions and keyswitch unchanged).  Needs the pair key
 * (three GGSWs per pair of key bits, fhs_client_bsk_mb2) loaded with fhs_load_multibit_key AFTER fhs_load_server_key in
 * arithmetic 1 or 2.  Same inputs, same outputs up to noise (bootstrap output sigma 2^49.62 instead of 2^48.87; the
 * string layer's noise margins hold in this arithmetic too, tests/test_gpu_noise.py), bit-exact against mode 4 of the
 * CPU oracle.  csrc/fftmb_kernels.hip. */
#define FHS_ARITH_F64_FFT_MB2 2
/* EXACT_NTT_MB2: the same two-key-bits-per-product blind rotation in EXACT integer arithmetic (two-prime NTT, like
 * EXACT_NTT): no f64 rounding anywhere, bootstrap output sigma 2^48.8 (lower than the classic f64 FFT's), 1.6x the rate
 * of EXACT_NTT.  Working order: fhs_load_server_key, then fhs_load_multibit_key WHILE arithmetic 0 (EXACT_NTT) is
 * selected -- the pair key is converted for the arithmetic selected at that moment: residues modulo the two NTT primes
 * on the 57-bit torus grid for 0 / 3, the Fourier domain for 1 / 2 -- then fhs_set_arithmetic(3), which is refused
 * until the converted key exists (likewise: load under arithmetic 1, then select 2).
 * Bit-exact against mode 5 of the CPU oracle (an independent exact algorithm).  csrc/nttmb_kernels.hip. */
#define FHS_ARITH_EXACT_NTT_MB2 3
#define FHS_BSK_MB2_WORDS ((size_t)371 * 3 * 4 * 2048)
int fhs_load_multibit_key(fhs_ctx *ctx, const uint64_t *bsk_mb2 /*[371][3][2 rows][2 cols][2048]*/);
int fhs_set_arithmetic(fhs_ctx *ctx, int arith);
int fhs_get_arithmetic(const fhs_ctx *ctx);
/* Tuning knob of the F64_FFT arithmetic: batches of at most `max_batch` ciphertexts run on the 4-wavefront kernel
 * (lower latency), larger ones on the 2-wavefront kernel (higher throughput).  Default 512; 0 = never, a huge value
 * = always.  Both kernels produce identical bits. */
int fhs_set_fft4_max_batch(fhs_ctx *ctx, int max_batch);
/* Tuning knob: a blind-rotation launch of arithmetic `arith` is cut into chunks of n_ciphertexts (0 = the whole batch in
 * one launch).  Every chunk starts all workgroups on the first key element together again, which keeps the key stream
 * inside the L2 for the kernels whose key does not fit it otherwise.  Results are identical. */
int fhs_set_launch_chunk(fhs_ctx *ctx, int arith, size_t n_ciphertexts);
/* Diagnostic: the host-derived twiddle tables of the F64_FFT mode (W[1024] re/im; U[16] re/im, 3 used). */
void fhs_fft_tables(double *w_re, double *w_im, double *u_re, double *u_im);
/* ... and the monomial evaluation table of the F64_FFT_MB2 mode: exp(i*pi*k/2048), k < 4096, (re, im) pairs. */
void fhs_fft_mono_table(double *mono /*[4096][2]*/);

/* ---- raw batched PBS (the hot path; kernel-level parity tests use these) ----
 * replaces: tfhe::shortint::ServerKey::apply_lookup_table on B blocks.
 * in[B][2049], lut_idx[B], luts[L][2048] (LUT body polynomials), out[B][2049]: host memory. */
int fhs_pbs_batch(fhs_ctx *ctx, const uint64_t *in, const uint32_t *lut_idx, const uint64_t *luts,
                  size_t n_luts, uint64_t *out, size_t B);
/* keyswitch + modulus switch only: ms_out[B][743] values in [0,4096) */
int fhs_keyswitch_modswitch_batch(fhs_ctx *ctx, const uint64_t *in, uint32_t *ms_out, size_t B);
/* Rotation sharing at the raw boundary: ONE keyswitch + blind rotation per input and n_shifts sample extractions of its
 * accumulator: out[b][s] ([B][n_shifts][2049]) is what a bootstrap of (in[b] + shifts[b][s] * 2^59) with the same table
 * yields (shifts in message units, 0..31; 0 = fhs_pbs_batch's output): adding c * 2^59 to a body moves the modulus-switched
 * body by exactly 128 c, i.e. rotates the accumulator by X^(128 c), so coefficient 128 c of ONE accumulator is the other
 * bootstrap's coefficient 0.  The string layer gets this automatically: rows of one dependency level that apply the same
 * table to the same linear combination up to its trivial constant -- the nibble of a character tested against the nibbles
 * of a clear pattern, is0(x - c) -- share a rotation (fhs_set_rotation_sharing, on by default in fused mode;
 * fhs_stats.pbs_extracted counts them).  shortint::ServerKey::apply_lookup_table per shifted input, batched. */
int fhs_pbs_batch_shifted(fhs_ctx *ctx, const uint64_t *in, const uint32_t *lut_idx, const uint64_t *luts, size_t n_luts,
                          const uint32_t *shifts /*[B][n_shifts]*/, size_t n_shifts, uint64_t *out, size_t B);
int fhs_set_rotation_sharing(fhs_ctx *ctx, int on);
/* Kernel-level tests: blind rotation + sample extraction only, in the selected arithmetic, from GIVEN keyswitched LWEs
 * ks[B][743] (u64 torus; the kernels apply the modulus switch to 2N = 4096 themselves) -> out[B][2049]. */
int fhs_debug_blind_rotate_batch(fhs_ctx *ctx, const uint64_t *ks, const uint32_t *lut_idx, const uint64_t *luts,
                                 size_t n_luts, uint64_t *out, size_t B);
/* Device-resident variant: all pointers are device pointers (e.g. torch tensors' data_ptr); work is enqueued
 * on `hip_stream` (0 = default).  A context owns one set of scratch buffers and one work counter: calls on it
 * must be ordered (one stream at a time); use several contexts for concurrent streams. */
int fhs_pbs_batch_device(fhs_ctx *ctx, const uint64_t *d_in, const uint32_t *d_lut_idx,
                         const uint64_t *d_luts, uint64_t *d_out, size_t B, void *hip_stream);
/* Average duration (ms) of the blind-rotation / keyswitch KERNEL launches since the last reset, measured with HIP events
 * on the launch stream, and the number of kernel launches timed (a batch cut into one-round launches --
 * fhs_set_launch_chunk, the default of the f64 kernels -- counts every launch: the figure is what `rocprofv3
 * --kernel-trace --stats` reports per kernel; x launches = the time of the batches). */
int fhs_kernel_timing(fhs_ctx *ctx, int reset, double *blind_rotate_ms, double *keyswitch_ms,
                      uint64_t *n_blind_rotate, uint64_t *n_keyswitch, uint64_t *pbs_in_launches);
/* Per kernel class: kind 0 = blind rotation on the exact-NTT kernel or the 2-wavefront FFT kernel (fhs_kernel_timing
 * reports this one), 1 = keyswitch, 2 = blind rotation on the 4-wavefront FFT kernel (batches <= fft4_max_batch).
 * Does not reset. */
int fhs_kernel_timing_kind(fhs_ctx *ctx, int kind, double *avg_ms, uint64_t *launches, uint64_t *pbs_in_launches);

/* ---- FheAsciiChar boundary ops (lazy DAG nodes) ------------------------------
 * Each constructor mirrors one method of src/ciphertext/fheasciichar.rs and returns a
 * handle; nothing runs until fhs_flush / fhs_download. 0 is never a valid handle. */
fhs_char_t fhs_trivial(fhs_ctx *ctx, uint8_t value);                    /* encrypt_trivial :17-25 */
fhs_char_t fhs_upload(fhs_ctx *ctx, const uint64_t *blocks /*[4][2049]*/); /* FheAsciiChar::new :13 */
/* n characters back to back ([n][4][2049] words, FheString.bytes of fhestring.rs:6-9 as the client produced them): one
 * staged copy and one scatter launch instead of one pageable copy per block (a 64-character string: 0.5 ms
 * instead of 2.5 ms).  out[n] receives the handles. */
int fhs_upload_string(fhs_ctx *ctx, const uint64_t *blocks, size_t n, fhs_char_t *out);
fhs_char_t fhs_eq(fhs_ctx *ctx, fhs_char_t a, fhs_char_t b);            /* eq  :35-38 */
fhs_char_t fhs_ne(fhs_ctx *ctx, fhs_char_t a, fhs_char_t b);            /* ne  :40-43 */
fhs_char_t fhs_le(fhs_ctx *ctx, fhs_char_t a, fhs_char_t b);            /* le  :45-48 */
fhs_char_t fhs_lt(fhs_ctx *ctx, fhs_char_t a, fhs_char_t b);            /* lt  :50-53 */
fhs_char_t fhs_ge(fhs_ctx *ctx, fhs_char_t a, fhs_char_t b);            /* ge  :55-58 */
fhs_char_t fhs_gt(fhs_ctx *ctx, fhs_char_t a, fhs_char_t b);            /* gt  :60-63 */
fhs_char_t fhs_bitand(fhs_ctx *ctx, fhs_char_t a, fhs_char_t b);        /* bitand :65-72 */
fhs_char_t fhs_bitor(fhs_ctx *ctx, fhs_char_t a, fhs_char_t b);         /* bitor  :74-81 */
fhs_char_t fhs_sub(fhs_ctx *ctx, fhs_char_t a, fhs_char_t b);           /* sub :83-86 */
fhs_char_t fhs_add(fhs_ctx *ctx, fhs_char_t a, fhs_char_t b);           /* add :88-91 */
fhs_char_t fhs_if_then_else(fhs_ctx *ctx, fhs_char_t cond, fhs_char_t t, fhs_char_t f); /* :93-104 */
fhs_char_t fhs_flip(fhs_ctx *ctx, fhs_char_t a);                        /* flip :161-168 */
fhs_char_t fhs_is_whitespace(fhs_ctx *ctx, fhs_char_t a);               /* :106-130 */
fhs_char_t fhs_is_uppercase(fhs_ctx *ctx, fhs_char_t a);                /* :132-144 */
fhs_char_t fhs_is_lowercase(fhs_ctx *ctx, fhs_char_t a);                /* :146-158 */
fhs_char_t fhs_clone(fhs_ctx *ctx, fhs_char_t a);                       /* #[derive(Clone)] :7 */
int fhs_release(fhs_ctx *ctx, fhs_char_t a);
int fhs_flush(fhs_ctx *ctx);                                             /* run every pending level */
/* Same, but only enqueues the launches on the context's HIP stream and returns (fhs_stream_sync, a download
 * or the next fhs_flush waits).  Lets several contexts on one GPU overlap: the narrow tail levels of one
 * operation run beside the wide first level of the next (bench.py --pipelines). */
int fhs_flush_async(fhs_ctx *ctx);
/* Level-skewed batching of independent requests (a server's steady state).  fhs_submit plans everything recorded since
 * the last submit / flush as one JOB and schedules its dependency levels on consecutive TICKS; fhs_pump(ctx, n) enqueues
 * the next n ticks, each as ONE launch group over the union of every job's level scheduled for it.  With one submit + one
 * pump per request, the narrow tail levels of request k ride in the wide launch of request k + 1 (a 64-char contains has
 * levels 496 / 62 / 5 / 1 wide: alone, the last three pay one bootstrap latency each on an almost empty GPU).  A job
 * that consumes results of an unfinished job is scheduled behind it.  Results are complete after fhs_flush (which
 * drains every tick) or a download. */
int fhs_submit(fhs_ctx *ctx);
/* Automatic partial flush: once `n_pending` bootstraps whose inputs are all available (dependency depth 1) are recorded,
 * that level is planned and enqueued while the caller keeps recording the rest of the operation; the other pending
 * bootstraps stay pending, one level shallower (default 8192; 0 = off).  On a device the level is also peeled as soon as
 * a grid's worth (1024) is ready and the previous launch group has finished.  Hides the host time of building large
 * DAGs (a 1024-character replace records 256 k bootstraps) behind GPU work.  Contexts driven with fhs_submit never
 * flush on their own. */
int fhs_set_auto_flush(fhs_ctx *ctx, size_t n_pending);
int fhs_pump(fhs_ctx *ctx, size_t n_ticks);
/* Round alignment of the launch groups (fhs_submit scheduling): the persistent blind-rotation kernel works on `slots`
 * ciphertexts at a time (fhs_resident_slots: 1024 on an MI355X for the f64-FFT kernels), so a launch group of 3.5 x slots
 * rows leaves half the chip idle during its last round.  With balancing on, a job level that would leave its tick's
 * group with a partly filled last round keeps only the rows that fill whole rounds; the excess (less than one round)
 * runs one tick later together with everything that consumes it.  Results are unchanged; a few results of a request are
 * complete one tick later.  0 = off (default). */
int fhs_set_tick_balance(fhs_ctx *ctx, size_t slots);
int fhs_resident_slots(const fhs_ctx *ctx);
int fhs_download(fhs_ctx *ctx, fhs_char_t a, uint64_t *blocks /*[4][2049]*/);
/* device-to-device import/export of one char (multi-GPU gather of partial results) */
/* n characters at once into [n][4][2049] words (what MyClientKey::decrypt, src/client_key.rs:89-106, walks): one gather
 * launch and one copy per 2048 blocks instead of a synchronous 16 KB copy per block (a 1025-character replace result:
 * ~10 ms instead of ~120 ms). */
int fhs_download_string(fhs_ctx *ctx, const fhs_char_t *chars, size_t n, uint64_t *blocks);
int fhs_export_device(fhs_ctx *ctx, fhs_char_t a, uint64_t *d_blocks /*[4][2049] device*/);
/* Stream-ordered variant: flushes asynchronously and enqueues the copies on the context's stream, no host wait.
 * fhs_stream_handle returns that hipStream_t so a caller can order its own work (e.g. an RCCL all-gather issued
 * under torch.cuda.ExternalStream) after the export and before a following fhs_import_device. */
int fhs_export_device_async(fhs_ctx *ctx, fhs_char_t a, uint64_t *d_blocks /*[4][2049] device*/);
void *fhs_stream_handle(fhs_ctx *ctx);
fhs_char_t fhs_import_device(fhs_ctx *ctx, const uint64_t *d_blocks);

/* ---- MyServerKey string methods (src/server_key/mod.rs, trim.rs) --------------
 * Strings are arrays of handles (FheString.bytes). mode: 0 = as written in the reference
 * (same op sequence), 1 = re-associated (log-depth trees, single-block flags; decrypts
 * identically).  Outputs are fresh handles owned by the caller. */
#define FHS_MODE_AS_WRITTEN 0
#define FHS_MODE_FUSED 1
int fhs_set_mode(fhs_ctx *ctx, int mode);
int fhs_str_contains(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *pat, size_t m, fhs_char_t *out);      /* mod.rs:151 */
int fhs_str_contains_clear(fhs_ctx *c, const fhs_char_t *s, size_t n, const char *pat, size_t m, fhs_char_t *out);      /* mod.rs:198 */
int fhs_str_starts_with(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *pat, size_t m, fhs_char_t *out);   /* mod.rs:344 */
int fhs_str_ends_with(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *pat, size_t m, fhs_char_t *out);     /* mod.rs:241 */
int fhs_str_find(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *pat, size_t m, fhs_char_t *out);          /* mod.rs:1010 */
int fhs_str_find_clear(fhs_ctx *c, const fhs_char_t *s, size_t n, const char *pat, size_t m, fhs_char_t *out);          /* mod.rs:1075 */
int fhs_str_rfind(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *pat, size_t m, fhs_char_t *out);         /* mod.rs:727 */
int fhs_str_is_empty(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t *out);                                       /* mod.rs:431 */
int fhs_str_len(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t *out);                                            /* mod.rs:478 */
int fhs_str_eq(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, fhs_char_t *out);            /* mod.rs:1122 */
int fhs_str_ne(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, fhs_char_t *out);            /* mod.rs:1178 */
int fhs_str_eq_ignore_case(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, fhs_char_t *out);/* mod.rs:1221 */
/* cmp: 0 lt, 1 le, 2 gt, 3 ge (enum Comparison, fhestring.rs:11-16) */
int fhs_str_compare(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, int cmp, fhs_char_t *out); /* mod.rs:1470 */
/* Positional half of the comparison (mod.rs:1497-1518) on two slices of equal length n: *any_diff = some position
 * differs, *verdict = a[i] < b[i] (cmp 0,1) resp. a[i] > b[i] (cmp 2,3) at the FIRST differing position, 0 if none.
 * Per-GPU partial of a position-sharded comparison: the first range that differs decides. */
int fhs_str_compare_partial(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, int cmp,
                            fhs_char_t *any_diff, fhs_char_t *verdict);
int fhs_str_to_upper(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t *out /*[n]*/);                               /* mod.rs:65 */
int fhs_str_to_lower(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t *out /*[n]*/);                               /* mod.rs:110 */
/* out_cap >= fhs_str_replace_len(n, m_from, m_to); *out_len receives the produced length */
size_t fhs_str_replace_len(size_t n, size_t m_from, size_t m_to);
int fhs_str_replace(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *from, size_t mf,
                    const fhs_char_t *to, size_t mt, fhs_char_t *out, size_t out_cap, size_t *out_len);                 /* mod.rs:624 */
int fhs_str_replacen(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *from, size_t mf,
                     const fhs_char_t *to, size_t mt, fhs_char_t count, fhs_char_t *out, size_t out_cap, size_t *out_len); /* mod.rs:1729 */
int fhs_str_repeat(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t count, fhs_char_t *out /*[16*n]*/);             /* mod.rs:567 */
int fhs_str_repeat_clear(fhs_ctx *c, const fhs_char_t *s, size_t n, size_t count, fhs_char_t *out /*[count*n]*/);        /* mod.rs:517 */
int fhs_str_concatenate(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, fhs_char_t *out /*[na+nb]*/); /* mod.rs:1864 */
int fhs_str_strip_prefix(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *pat, size_t m, fhs_char_t *out /*[n]*/, fhs_char_t *found); /* mod.rs:1261 */
int fhs_str_strip_suffix(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *pat, size_t m, fhs_char_t *out /*[n]*/, fhs_char_t *found); /* mod.rs:1335 */
int fhs_str_trim_end(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t *out /*[n]*/);                               /* trim.rs:36 */
int fhs_str_trim_start(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t *out /*[n]*/);                             /* trim.rs:86 */
int fhs_str_trim(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t *out /*[n]*/);                                   /* trim.rs:146 */
int fhs_bubble_zeroes_right(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t *out /*[n]*/);                        /* utils.rs:28 */
/* split family (src/server_key/split.rs, FheSplit of src/ciphertext/fhesplit.rs:5-8).
 * kind: 0 split :989, 1 split_inclusive :1020, 2 split_terminator :1051, 3 splitn :1448, 4 rsplit :394,
 * 5 rsplit_terminator :504, 6 rsplitn :421, 7 rsplit_once :462, 8 split_ascii_whitespace :1377.
 * `count` is the encrypted (or trivial) n of splitn/rsplitn, 0 otherwise.  The reference returns
 * d buffers of d chars with d = n + 1 (d = n for kind 8): out[d*d] row-major, *dim = d. */
size_t fhs_str_split_dim(int kind, size_t n);
int fhs_str_split(fhs_ctx *c, int kind, const fhs_char_t *s, size_t n, const fhs_char_t *pat, size_t m,
                  fhs_char_t count, fhs_char_t *out, size_t out_cap, size_t *dim, fhs_char_t *found);
/* OR / AND of n 0/1 flag chars in log_15(n) levels (chains of bitor/bitand :65-81 re-associated);
 * used to combine per-GPU partial results after the gather. */
int fhs_flags_or(fhs_ctx *c, const fhs_char_t *flags, size_t n, fhs_char_t *out);
int fhs_flags_and(fhs_ctx *c, const fhs_char_t *flags, size_t n, fhs_char_t *out);
/* Combines the partials of fhs_str_compare_partial over n consecutive ranges (string order): the first range that
 * differs decides; `tie` (0 or 1) if none differs. */
int fhs_flags_first_decides(fhs_ctx *c, const fhs_char_t *any_diff, const fhs_char_t *verdict, size_t n, int tie,
                            fhs_char_t *out);

/* ---- multi-GPU inside the library (one process per GPU, RCCL over xGMI) ---------------------------------
 * north_star: "characters of an FheString are independent PBS batches, so long strings shard across the GPUs of one
 * node with an RCCL gather".  fhs_dist_init creates this context's communicator (librccl.so.1 is loaded at run time;
 * rank 0 obtains the 128-byte id with fhs_dist_unique_id and hands it to the other ranks by any means).  All
 * collectives are ncclAllGather calls enqueued on the context's own HIP stream: nothing waits on the host. */
#define FHS_DIST_ID_BYTES 128
int fhs_dist_unique_id(void *id /*[128]*/);
int fhs_dist_init(fhs_ctx *ctx, int rank, int world, const void *nccl_unique_id /*[128]*/);
/* Transport for ranks that SHARE one GPU (RCCL refuses two ranks on one device; tests on a one-GPU box): the library
 * stages through pinned host memory and calls `fn(user, send, recv, bytes_per_rank)` (0 = ok), e.g. a gloo all-gather. */
typedef int (*fhs_allgather_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank);
int fhs_dist_init_host_transport(fhs_ctx *ctx, int rank, int world, fhs_allgather_fn fn, void *user);
int fhs_dist_shutdown(fhs_ctx *ctx);
/* Teardown after a bring-up that did not succeed on EVERY rank (fhs_dist_init returned 0 here and an error elsewhere):
 * the half-formed communicator is aborted (ncclCommAbort: no exchange with the peers) instead of destroyed, nothing is
 * flushed or waited for.  fhs_dist_shutdown on such a communicator may block. */
int fhs_dist_abort(fhs_ctx *ctx);
/* 1 if librccl.so.1 can be loaded with every entry point used here (dlopen + dlsym only: no communicator is made, no
 * GPU is touched).  ncclCommInitRank is collective: agree on this among ALL ranks before any of them calls
 * fhs_dist_init, or the ranks that could load it block forever waiting for the one that could not. */
int fhs_dist_available(void);
/* Exchange counters of this context since fhs_dist_init*: all-gathers issued (ncclAllGather calls, or host-transport
 * callbacks), bytes THIS rank contributed, and which transport carries them. */
#define FHS_TRANSPORT_NONE 0
#define FHS_TRANSPORT_RCCL 1
#define FHS_TRANSPORT_HOST 2
int fhs_dist_stats(const fhs_ctx *ctx, uint64_t *n_allgather, uint64_t *bytes_sent, int *transport);
int fhs_dist_rank(const fhs_ctx *ctx);
int fhs_dist_world(const fhs_ctx *ctx);
/* Partition helpers (pure host logic).  Windows 0..n_chars-m of contains/find split into `world` contiguous ranges:
 * rank evaluates windows [*w0, *w1) and holds characters [*c0, *c1) (its slice plus an (m-1)-character halo).
 * Positions 0..n_chars split into `world` contiguous ranges [*c0, *c1) (eq / eq_ignore_case / comparisons). */
void fhs_dist_plan_windows(size_t n_chars, size_t m, int world, int rank, size_t *w0, size_t *w1, size_t *c0, size_t *c1);
void fhs_dist_plan_positions(size_t n_chars, int world, int rank, size_t *c0, size_t *c1);
/* Generic exchange: n chars (resp. n single-block flags) of every rank -> out[r * n + i] on every rank. */
int fhs_dist_allgather_chars(fhs_ctx *ctx, const fhs_char_t *local, size_t n, fhs_char_t *out /*[world * n]*/);
int fhs_dist_allgather_flags(fhs_ctx *ctx, const fhs_char_t *local, size_t n, fhs_char_t *out /*[world * n]*/);
/* Sharded string methods: `shard` is THIS rank's slice (fhs_dist_plan_*), patterns are replicated; every rank returns
 * the full result.  contains: mod.rs:151-182 (1 block exchanged per rank).  find: mod.rs:1010-1053, every rank
 * bootstraps the match flags of its windows (the two wide levels), ONE all-gather of ceil(W / world) blocks per rank hands
 * every rank all W flags, the narrow rest (prefix OR, index of the first flag) runs replicated: six dependency levels
 * like the single-GPU find; `shard` / `first_window` must be this rank's slice of fhs_dist_plan_windows(total_chars, m);
 * FHS_ERR_LIMIT when total_chars >= 255 + m like the reference's panic.  eq / eq_ignore_case:
 * mod.rs:1122-1149, :1221-1231 on equally long padded buffers (1 block).  compare: mod.rs:1470-1541, cmp 0 lt, 1 le,
 * 2 gt, 3 ge (2 blocks). */
int fhs_dist_str_contains(fhs_ctx *c, const fhs_char_t *shard, size_t n, const fhs_char_t *pat, size_t m, fhs_char_t *out);
int fhs_dist_str_contains_clear(fhs_ctx *c, const fhs_char_t *shard, size_t n, const char *pat, size_t m, fhs_char_t *out);
int fhs_dist_str_find(fhs_ctx *c, const fhs_char_t *shard, size_t n, const fhs_char_t *pat, size_t m, size_t first_window,
                      size_t total_chars, fhs_char_t *out);
int fhs_dist_str_find_clear(fhs_ctx *c, const fhs_char_t *shard, size_t n, const char *pat, size_t m, size_t first_window,
                            size_t total_chars, fhs_char_t *out);
int fhs_dist_str_eq(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, int ignore_case,
                    fhs_char_t *out);
int fhs_dist_str_compare(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, int cmp, fhs_char_t *out);
/* Level-parallel mode for ANY op (replace with its compaction, ...): every rank holds the same ciphertexts and records
 * the same DAG; while on, fhs_flush / fhs_flush_async run slice [rank*cap, (rank+1)*cap) of every dependency level,
 * all-gather the level (cap x 16 392 B per rank) on the stream and install it -- all levels enqueued back to back. */
int fhs_dist_level_parallel(fhs_ctx *ctx, int on);

/* ---- level-parallel execution with a caller-provided collective (low-level protocol; SURVEY 5 "per-level") ----
 * Every rank holds the same ciphertexts and records the same DAG.  fhs_flush_plan levelises it;
 * for each level k every rank runs its slice [rank*cap, (rank+1)*cap) of the level's PBS
 * (cap = ceil(width / world)) into rows of `d_slice` (device, cap x 2049 u64), the caller all-gathers
 * the slices (RCCL) into `d_all` (world x cap x 2049) and fhs_flush_level_commit installs the results.
 * fhs_flush() refuses to run while world > 1 has pending work. */
int fhs_dist_config(fhs_ctx *ctx, int rank, int world);
int fhs_flush_plan(fhs_ctx *ctx, uint64_t *n_levels, uint64_t *max_level_width);
int fhs_flush_level_exec(fhs_ctx *ctx, uint64_t level, uint64_t *d_slice, uint64_t *width, uint64_t *cap);
int fhs_flush_level_commit(fhs_ctx *ctx, uint64_t level, const uint64_t *d_all);
int fhs_stream_sync(fhs_ctx *ctx);

/* ---- debug: capture of PBS inputs (noise-margin measurement; tests/test_gpu_noise.py) -------------
 * While enabled, every executed level copies a strided sample (at most max_rows_per_level rows) of its PBS inputs --
 * the linear-combination results that enter keyswitch -- to host memory, with the record below.  The caller
 * decrypts their phase with the client key; the library never sees a secret.  0 disables and clears. */
typedef struct {
    uint32_t level, index;     /* dependency level within its flush, position within the level */
    uint32_t lut;              /* LUT catalogue id of the PBS (csrc/luts.h) */
    uint32_t n_terms;          /* ciphertext terms of the linear combination */
    int64_t sum_c2;            /* sum of squared coefficients (noise amplification of the construct) */
    int32_t konst;             /* trivial constant added (mod 32) */
    uint32_t width;            /* PBS in this level */
} fhs_capture_rec;
int fhs_debug_capture_pbs_inputs(fhs_ctx *ctx, size_t max_rows_per_level);
/* on != 0: sample inside the ordinary execution path -- rotation sharing, round alignment and tick scheduling exactly as
 * in production (tests/test_gpu_margins.py); 0 (default): through an all-at-once plan without rotation sharing.  `level`
 * of a record then counts the job levels executed since the last fhs_reset_stats. */
int fhs_debug_capture_live(fhs_ctx *ctx, int on);
/* Copies up to `cap` captured rows ([2049] u64 each) and records, returns the number in *n and clears the capture;
 * rows == NULL only reports the count (nothing is cleared). */
int fhs_debug_capture_read(fhs_ctx *ctx, uint64_t *rows, fhs_capture_rec *recs, size_t cap, size_t *n);

/* ---- debug: plan trace of a planner context (tests/test_plan_exec.py; bench.py's CPU-baseline leg) --------------
 * While on, a planner context (fhs_ctx_create_planner: no device) writes what it WOULD run as 64-bit words; block tokens
 * are the planner's unique fake pointers:
 *   1 token                                    an uploaded block, in upload order
 *   2 out lut konst n (token coef) x n         one bootstrap: LUT `lut` of sum coef * block + konst * 2^59 -> block `out`
 *   3 leader_out out K                         a shared extraction: coefficient K of the accumulator of `leader_out`'s row
 *                                              (= that row's bootstrap with konst + K / 128)
 *   4 width                                    end of a launch group (its rows are independent of one another)
 * A host executor replays the list with any bootstrap implementation: the CPU oracle runs the product's fused DAGs on
 * real ciphertexts this way (oracle/plan_exec.py).  fhs_debug_char_terms describes a result handle after the flush, per
 * block: kind (0 plaintext, 1 block, 2 linear combination), value or konst, n, (token coef) x n.  fhs_debug_lut_poly:
 * the catalogue's body polynomial [2048] of a LUT id (csrc/luts.h). */
int fhs_debug_plan_trace(fhs_ctx *ctx, int on);
int fhs_debug_plan_read(fhs_ctx *ctx, uint64_t *out, size_t cap, size_t *n);
int fhs_debug_char_terms(fhs_ctx *ctx, fhs_char_t h, uint64_t *out, size_t cap, size_t *n);
int fhs_debug_lut_poly(int lut_id, uint64_t *out);

/* ---- statistics ----------------------------------------------------------------
 * fhs_stats grows at its END when a counter is added (round 5: pbs_extracted): a host must be compiled against the
 * header of the library it loads -- fhs_get_stats writes sizeof(fhs_stats) bytes of THAT build.  (build() recompiles
 * examples/c_host for this reason; the generated Rust binding carries the same layout.) */
typedef struct {
    uint64_t pbs_executed;     /* PBS actually run on the GPU (constant-folded ones excluded) */
    uint64_t pbs_folded;       /* PBS on all-trivial inputs folded at DAG construction */
    uint64_t levels;           /* dependency levels launched */
    uint64_t max_level_width;
    uint64_t blocks_live;      /* device ciphertext blocks currently allocated */
    uint64_t max_input_sum_c2; /* largest sum of squared coefficients of a bootstrap's input (flattened linear
                                * combination of bootstrap outputs / uploads): its noise variance in units of one
                                * bootstrap output's.  The string layer keeps it <= FHS_NOISE_BUDGET_SUM_C2. */
    uint64_t pbs_shared;       /* fused mode: PBS not run because an identical one (same LUT on the same linear combination
                                * of the same blocks) already exists -- e.g. the high-nibble tests of one character
                                * against pattern characters that share their high nibble */
    uint64_t pbs_extracted;    /* results obtained as a further sample extraction of ANOTHER row's blind rotation (rotation
                                * sharing: same table, same combination up to its trivial constant); not in pbs_executed */
} fhs_stats;
/* Design rule of the fused DAGs: with a measured bootstrap-output sigma of 2^48.9 a sum with
 * sum c^2 <= 64 adds sigma <= 2^51.9 to the 2^55.2 of keyswitch + modulus switch (+0.8 %): the reference parameter
 * set's 2^-40 failure probability is kept (tests/test_gpu_noise.py measures it). */
#define FHS_NOISE_BUDGET_SUM_C2 64
int fhs_get_stats(fhs_ctx *ctx, fhs_stats *out);
/* Noise of a handle in the same unit: the largest sum of squared coefficients over its four blocks (0 trivial, 1 a
 * bootstrap output or an upload).  Results are handed back at <= 4 except the index of find / find_clear, whose digits
 * are sums of up to 57 bootstrap outputs (inside the decryption margin; saves one dependency level).  The library
 * refreshes a block above 4 by itself when its handle is used as an operand, also after a download (a linear
 * combination materialised for fhs_download / fhs_export_device keeps its figure).  A ciphertext that LEAVES the library
 * has to carry the figure with it, like the noise_level of a tfhe-rs shortint ciphertext travels in its serialised form
 * (tfhe 0.5.2, Cargo.lock:416-433): query it here before the download ... */
int fhs_char_sum_c2(fhs_ctx *ctx, fhs_char_t h, uint64_t *out);
/* ... and declare it after fhs_upload / fhs_upload_string of such a ciphertext (uploads count as 1 otherwise, the
 * figure of a fresh client encryption -- fheasciichar.rs:27-29).  Handles of uploaded (or trivial) blocks only; the
 * figure can be raised, never lowered below what the library tracks. */
int fhs_char_set_noise(fhs_ctx *ctx, fhs_char_t h, uint64_t sum_c2);
/* Constant folding made visible: *is_trivial = 1 and *value = the byte if all four blocks of the handle are trivial
 * (plaintext) ciphertexts -- what an operation on trivially encrypted inputs folds to, without a GPU (planner contexts
 * included).  The CPU tests evaluate the re-associated DAGs on every byte pair this way. */
int fhs_trivial_value(fhs_ctx *ctx, fhs_char_t h, int *is_trivial, uint8_t *value);
/* Width (PBS count) of every dependency level executed since the last fhs_reset_stats, in execution order (the shape
 * of the levelized batches: what a CPU baseline has to run to do the same work).  *n = number of levels; out may be
 * NULL to query it. */
int fhs_level_widths(fhs_ctx *ctx, uint32_t *out, size_t cap, size_t *n);
/* Rows THIS rank ran in every launch group (one lincomb -> keyswitch -> blind-rotation sequence) since the last
 * fhs_reset_stats, in order: several dependency levels may share a group (level-skewed batching, round alignment) and a
 * level-parallel rank runs its slice of each.  With a planner context that has a host transport attached
 * (fhs_dist_init_host_transport with any callback: it is never called) the sharded entry points can be RECORDED for any
 * (rank, world): groups, PBS, all-gathers and bytes of fhs_dist_stats are what a real run of that rank would issue --
 * the basis of tools/project_multi_gpu.py. */
int fhs_launch_groups(fhs_ctx *ctx, uint32_t *out, size_t cap, size_t *n);
int fhs_reset_stats(fhs_ctx *ctx);

/* ---- client side (MyClientKey, src/client_key.rs) -- host CPU, like the reference -- */
/* Keys, masks and noise come from ChaCha20 keyed with 256 bits of getrandom(2) entropy (the reference: tfhe-rs's
 * OS-seeded concrete-csprng); separate streams for secret keys, public masks and noise. */
int fhs_client_create(fhs_client **out);                                 /* from_params :30-35 */
/* TEST / BENCHMARK ONLY: the same generator keyed from a 64-bit seed -> reproducible keys and ciphertexts (identical
 * keys on every rank of a multi-GPU run, fixed test vectors).  At most 64 bits of entropy: never for real data. */
int fhs_client_create_insecure_seeded(uint64_t seed, fhs_client **out);
/* Diagnostic: one block of the generator's ChaCha20 (RFC 8439 2.3.2 known-answer test in tests/test_cabi.py). */
void fhs_chacha20_block(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint32_t out[16]);
/* Diagnostic: n 64-bit draws from that state, i.e. the keystream of consecutive blocks in order (the generator makes
 * eight blocks at a time with AVX2 where the CPU has it: same stream as the scalar block function). */
void fhs_chacha20_stream(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint64_t *out, size_t n);
void fhs_client_destroy(fhs_client *ck);
const uint64_t *fhs_client_bsk(const fhs_client *ck);                    /* get_server_key :37-39 */
const uint64_t *fhs_client_ksk(const fhs_client *ck);
/* pair key of FHS_ARITH_F64_FFT_MB2 (generated on first call; FHS_BSK_MB2_WORDS words): for each pair of LWE key bits
 * (s, s') GGSW encryptions of s(1-s'), (1-s)s' and s s' */
const uint64_t *fhs_client_bsk_mb2(fhs_client *ck);
int fhs_client_encrypt_char(fhs_client *ck, uint8_t v, uint64_t *blocks /*[4][2049]*/); /* encrypt_char :85-87 */
int fhs_client_decrypt_char(const fhs_client *ck, const uint64_t *blocks, uint8_t *out); /* decrypt_char :81-83 */
/* encrypt :45-65 (ASCII, no NUL, `padding` NULs appended): out[(len+padding)][4][2049] */
int fhs_client_encrypt_str(fhs_client *ck, const char *s, size_t len, size_t padding, uint64_t *out);
/* decrypt :89-106 (truncates at the first NUL); returns the number of bytes written */
int fhs_client_decrypt_str(const fhs_client *ck, const uint64_t *chars, size_t n, char *out, size_t *out_len);
int fhs_client_secret_keys(const fhs_client *ck, uint64_t *lwe_sk /*[742]*/, uint64_t *glwe_sk /*[2048]*/);

/* ---- key files (the reference derives serde traits at client_key.rs:9 and
 * server_key/mod.rs:13 but never calls them).  Little-endian: 64-byte header {magic "FHSKEY01", kind,
 * lwe_n, poly_n, ks_levels, ks_base_log, pbs_base_log, bsk_quant_bits}, then raw u64 arrays.
 * kind 1 = client key (secret keys + server key), kind 2 = server key only (bsk, ksk). */
int fhs_client_save(const fhs_client *ck, const char *path, int server_key_only);
int fhs_client_load(const char *path, fhs_client **out);             /* kind 1 files only */
int fhs_load_server_key_file(fhs_ctx *ctx, const char *path);        /* kind 1 or 2 */
/* kind 3 = the pair key of FHS_ARITH_F64_FFT_MB2 / FHS_ARITH_EXACT_NTT_MB2 alone (generated on first use): it travels
 * beside a kind 1 / 2 file and is loaded after it, converted for the arithmetic selected at that moment */
int fhs_client_save_multibit_key(fhs_client *ck, const char *path);
int fhs_load_multibit_key_file(fhs_ctx *ctx, const char *path);

#ifdef __cplusplus
}
#endif
#endif
