"""Thin Python mirror of the C ABI (include/fhestring_hip.h)."""
import ctypes as C

import numpy as np

from ._lib import FhsError, lib, check_single_hip_runtime

BIG_CT = 2049
SMALL_CT = 743
POLY_N = 2048


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """fhs_ctx: one per GPU.  Replaces the tfhe::integer::ServerKey held by
    MyServerKey (src/server_key/mod.rs:13-16)."""

    def __init__(self, device_id=0, planner=False):
        self._L = lib()
        if not planner:
            check_single_hip_runtime()     # a second HIP runtime mapped since the library was loaded: refuse, do not crash
        h = C.c_void_p()
        if planner:        # fhs_ctx_create_planner: records and levelises DAGs (statistics only), computes nothing
            rc = self._L.fhs_ctx_create_planner(C.byref(h))
        else:
            rc = self._L.fhs_ctx_create(int(device_id), C.byref(h))
        self._h = h
        if rc != 0:
            msg = self._L.fhs_last_error(h).decode() if h else "allocation failed"
            if h:
                self._L.fhs_ctx_destroy(h)
            self._h = None
            raise FhsError("fhs_ctx_create failed (%d): %s" % (rc, msg))

    def close(self):
        if getattr(self, "_h", None):
            self._L.fhs_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise FhsError("fhs error %d: %s" % (rc, self._L.fhs_last_error(self._h).decode()))

    ARITH_EXACT_NTT = 0
    ARITH_F64_FFT = 1
    ARITH_F64_FFT_MB2 = 2
    ARITH_EXACT_NTT_MB2 = 3

    def set_arithmetic(self, arith):
        """fhs_set_arithmetic: 0 ARITH_EXACT_NTT (default), 1 ARITH_F64_FFT, 2 ARITH_F64_FFT_MB2 (select before
        load_server_key; mode 2 also needs load_multibit_key)."""
        self._check(self._L.fhs_set_arithmetic(self._h, int(arith)))

    def set_launch_chunk(self, arith, n):
        """fhs_set_launch_chunk: ciphertexts per blind-rotation launch of that arithmetic (0 = whole batch)."""
        self._check(self._L.fhs_set_launch_chunk(self._h, int(arith), int(n)))

    def set_fft4_max_batch(self, n):
        """fhs_set_fft4_max_batch: batches <= n use the 4-wavefront FFT kernel (default 512)."""
        self._check(self._L.fhs_set_fft4_max_batch(self._h, int(n)))

    @property
    def arithmetic(self):
        return int(self._L.fhs_get_arithmetic(self._h))

    def load_server_key(self, bsk, ksk):
        bsk = np.ascontiguousarray(bsk, np.uint64)
        ksk = np.ascontiguousarray(ksk, np.uint64)
        assert bsk.size == 742 * 4 * 2048 and ksk.size == 2048 * 5 * 743
        self._check(self._L.fhs_load_server_key(self._h, _ptr(bsk), _ptr(ksk)))

    def blind_rotate_batch(self, ks, lut_idx, luts):
        """fhs_debug_blind_rotate_batch: blind rotation + sample extraction from given keyswitched LWEs [B, 743]."""
        ks = np.ascontiguousarray(ks, np.uint64).reshape(-1, 743)
        luts = np.ascontiguousarray(luts, np.uint64).reshape(-1, POLY_N)
        lut_idx = np.ascontiguousarray(lut_idx, np.uint32)
        out = np.zeros((ks.shape[0], BIG_CT), np.uint64)
        self._check(self._L.fhs_debug_blind_rotate_batch(self._h, _ptr(ks), _ptr(lut_idx), _ptr(luts), luts.shape[0],
                                                         _ptr(out), ks.shape[0]))
        return out

    def load_multibit_key(self, bsk_mb2):
        """fhs_load_multibit_key: pair key of ARITH_F64_FFT_MB2, after load_server_key in arithmetic 1 or 2."""
        bsk_mb2 = np.ascontiguousarray(bsk_mb2, np.uint64)
        assert bsk_mb2.size == 371 * 3 * 4 * 2048
        self._check(self._L.fhs_load_multibit_key(self._h, _ptr(bsk_mb2)))

    def pbs_batch(self, cts, lut_idx, luts):
        cts = np.ascontiguousarray(cts, np.uint64).reshape(-1, BIG_CT)
        luts = np.ascontiguousarray(luts, np.uint64).reshape(-1, POLY_N)
        lut_idx = np.ascontiguousarray(lut_idx, np.uint32)
        B = cts.shape[0]
        assert lut_idx.size == B
        out = np.zeros((B, BIG_CT), np.uint64)
        self._check(self._L.fhs_pbs_batch(self._h, _ptr(cts), _ptr(lut_idx), _ptr(luts), luts.shape[0],
                                          _ptr(out), B))
        return out

    def pbs_batch_shifted(self, cts, lut_idx, luts, shifts):
        """fhs_pbs_batch_shifted: ONE blind rotation per input, one sample extraction per shift -> [B, S, 2049]; row
        (b, s) = what a bootstrap of (cts[b] + shifts[b][s] * Delta) yields (message units 0..31)."""
        cts = np.ascontiguousarray(cts, np.uint64).reshape(-1, BIG_CT)
        luts = np.ascontiguousarray(luts, np.uint64).reshape(-1, POLY_N)
        lut_idx = np.ascontiguousarray(lut_idx, np.uint32)
        B = cts.shape[0]
        shifts = np.ascontiguousarray(shifts, np.uint32).reshape(B, -1)
        out = np.zeros((B, shifts.shape[1], BIG_CT), np.uint64)
        self._check(self._L.fhs_pbs_batch_shifted(self._h, _ptr(cts), _ptr(lut_idx), _ptr(luts), luts.shape[0],
                                                  _ptr(shifts), shifts.shape[1], _ptr(out), B))
        return out

    def set_rotation_sharing(self, on=True):
        self._check(self._L.fhs_set_rotation_sharing(self._h, int(bool(on))))

    def keyswitch_modswitch_batch(self, cts):
        cts = np.ascontiguousarray(cts, np.uint64).reshape(-1, BIG_CT)
        B = cts.shape[0]
        out = np.zeros((B, SMALL_CT), np.uint32)
        self._check(self._L.fhs_keyswitch_modswitch_batch(self._h, _ptr(cts), _ptr(out), B))
        return out

    def pbs_batch_device(self, d_in, d_lut_idx, d_luts, d_out, B, stream=0):
        """All arguments are raw device pointers (ints), e.g. torch tensor.data_ptr()."""
        self._check(self._L.fhs_pbs_batch_device(self._h, C.c_void_p(d_in), C.c_void_p(d_lut_idx),
                                                 C.c_void_p(d_luts), C.c_void_p(d_out), int(B),
                                                 C.c_void_p(stream)))

    def kernel_timing(self, reset=False):
        br, ks = C.c_double(), C.c_double()
        nb, nk, units = C.c_uint64(), C.c_uint64(), C.c_uint64()
        narrow = self.kernel_timing_kind(2)                     # read before a reset
        self._check(self._L.fhs_kernel_timing(self._h, int(reset), C.byref(br), C.byref(ks), C.byref(nb),
                                              C.byref(nk), C.byref(units)))
        return {"blind_rotate_ms": br.value, "keyswitch_ms": ks.value, "n_blind_rotate": nb.value,
                "n_keyswitch": nk.value, "pbs_in_launches": units.value,
                "fft4_ms": narrow[0], "n_fft4": narrow[1], "pbs_in_fft4": narrow[2]}

    def kernel_timing_kind(self, kind):
        """(average ms, launches, PBS) of one kernel class: 0 wide blind rotation, 1 keyswitch, 2 4-wavefront FFT."""
        ms, n, u = C.c_double(), C.c_uint64(), C.c_uint64()
        self._check(self._L.fhs_kernel_timing_kind(self._h, int(kind), C.byref(ms), C.byref(n), C.byref(u)))
        return ms.value, n.value, u.value


# ---------------------------------------------------------------------------------------------
# Mirror of the reference's types: MyClientKey (src/client_key.rs), FheAsciiChar / FheString
# (src/ciphertext), MyServerKey (src/server_key/mod.rs).  Same method names and argument meaning;
# `public_parameters` is accepted and ignored like the reference's dead parameter.
# ---------------------------------------------------------------------------------------------
CHAR_WORDS = 4 * BIG_CT
MAX_FIND_LENGTH = 255
MAX_REPETITIONS = 16
STRING_PADDING = 1
MODE_AS_WRITTEN = 0
MODE_FUSED = 1


class MyClientKey:
    """Host-CPU client key (keygen / encrypt / decrypt), like the reference's.

    MyClientKey() draws everything from a ChaCha20 generator keyed with OS entropy (fhs_client_create), like the
    reference's OS-seeded tfhe-rs CSPRNG.  MyClientKey(seed) is the TEST / BENCHMARK constructor
    (fhs_client_create_insecure_seeded): reproducible keys, identical on every rank, never for real data."""

    def __init__(self, seed=None, _handle=None):
        import threading
        self._L = lib()
        self._scratch, self._scratch_lock = None, threading.Lock()
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        if seed is None:
            rc = self._L.fhs_client_create(C.byref(h))
        else:
            rc = self._L.fhs_client_create_insecure_seeded(int(seed), C.byref(h))
        if rc != 0:
            raise FhsError("fhs_client_create failed (%d)" % rc)
        self._h = h

    def save(self, path, server_key_only=False):
        """Key file (include/fhestring_hip.h "key files"): raw little-endian u64 arrays + 64-byte header."""
        if self._L.fhs_client_save(self._h, str(path).encode(), int(server_key_only)) != 0:
            raise FhsError("cannot write key file %s" % path)

    def save_multibit_key(self, path):
        """fhs_client_save_multibit_key: kind 3 file with the pair key of the two-key-bits-per-product arithmetics."""
        if self._L.fhs_client_save_multibit_key(self._h, str(path).encode()) != 0:
            raise FhsError("cannot write pair key file %s" % path)

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        if lib().fhs_client_load(str(path).encode(), C.byref(h)) != 0:
            raise FhsError("cannot read client key file %s" % path)
        return cls(_handle=h)

    @classmethod
    def from_params(cls, params=None, num_blocks=4, seed=None):   # client_key.rs:30-35 (seed: tests only)
        assert num_blocks == 4
        return cls(seed)

    def close(self):
        if getattr(self, "_h", None):
            self._L.fhs_client_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bsk(self):
        return np.ctypeslib.as_array(self._L.fhs_client_bsk(self._h), shape=(742 * 4 * 2048,))

    def ksk(self):
        return np.ctypeslib.as_array(self._L.fhs_client_ksk(self._h), shape=(2048 * 5 * 743,))

    def bsk_mb2(self):
        """fhs_client_bsk_mb2: pair key of ARITH_F64_FFT_MB2 (generated on first use)."""
        return np.ctypeslib.as_array(self._L.fhs_client_bsk_mb2(self._h), shape=(371 * 3 * 4 * 2048,))

    def secret_keys(self):
        lwe = np.zeros(742, np.uint64)
        glwe = np.zeros(2048, np.uint64)
        self._L.fhs_client_secret_keys(self._h, _ptr(lwe), _ptr(glwe))
        return lwe, glwe

    def get_server_key(self, device_id=0, arith=0):                      # client_key.rs:37-39
        return MyServerKey.from_client_key(self, device_id, arith)

    def encrypt_char_raw(self, v):
        out = np.zeros((4, BIG_CT), np.uint64)
        self._L.fhs_client_encrypt_char(self._h, int(v) & 255, _ptr(out))
        return out

    def encrypt_str_raw(self, text, padding, out=None):
        data = text.encode("ascii") if isinstance(text, str) else bytes(text)
        if out is None:
            out = np.empty((len(data) + padding, 4, BIG_CT), np.uint64)  # every word is written by the call
        rc = self._L.fhs_client_encrypt_str(self._h, data, len(data), int(padding), _ptr(out))
        if rc != 0:
            raise AssertionError("The input string must only contain ascii letters and not include null characters")
        return out

    def decrypt_char_raw(self, blocks):
        v = C.c_uint8()
        blocks = np.ascontiguousarray(blocks, np.uint64)
        self._L.fhs_client_decrypt_char(self._h, _ptr(blocks), C.byref(v))
        return v.value

    def decrypt_str_raw(self, chars):
        chars = np.ascontiguousarray(chars, np.uint64).reshape(-1, 4, BIG_CT)
        buf = C.create_string_buffer(chars.shape[0] + 1)
        n = C.c_size_t()
        self._L.fhs_client_decrypt_str(self._h, _ptr(chars), chars.shape[0], buf, C.byref(n))
        return buf.raw[:n.value].decode("ascii")

    # reference-shaped API (server key needed to place ciphertexts on the device)
    def encrypt(self, string, padding, public_parameters=None, server_key=None):   # :45-65
        # the ciphertext only lives until fhs_upload_string has staged it: one scratch buffer per client, grown on demand
        # (a fresh 268 MB array per 4097-character string is 65 000 page faults before the first byte is encrypted)
        # ctypes releases the GIL inside fhs_client_encrypt_str / fhs_upload_string (the C side is thread-safe, per-call
        # streams): the scratch buffer is therefore guarded by a per-client lock -- two threads encrypting with one client
        # key take turns -- and a buffer that grew beyond 320 MB (more than BASELINE's largest string, 4097
        # characters = 268 MB) is dropped after use instead of staying pinned for the client's lifetime
        n = len(string) + padding
        with self._scratch_lock:
            if self._scratch is None or self._scratch.shape[0] < n:
                self._scratch = np.empty((max(n, 64), 4, BIG_CT), np.uint64)
            try:
                return server_key.upload_string(self.encrypt_str_raw(string, padding, out=self._scratch[:n]))
            finally:
                if self._scratch.nbytes > (320 << 20):
                    self._scratch = None

    def encrypt_no_padding(self, string, server_key=None):                         # :67-79
        return server_key.upload_string(self.encrypt_str_raw(string, 0)).chars

    def encrypt_char(self, v, server_key=None):                                    # :85-87
        return server_key.upload_char(self.encrypt_char_raw(v))

    def decrypt_char(self, ch):                                                    # :81-83
        return self.decrypt_char_raw(ch.download())

    def decrypt(self, fhe_string):                                                 # :89-106
        return self.decrypt_str_raw(fhe_string.download())


class FheAsciiChar:
    """Handle of one lazily evaluated encrypted char (fheasciichar.rs:8-10)."""
    __slots__ = ("sk", "h")

    def __init__(self, sk, h):
        if not h:
            raise FhsError("null char handle: " + sk.ctx._L.fhs_last_error(sk.ctx._h).decode())
        self.sk = sk
        self.h = h

    def __del__(self):
        try:
            if self.h and self.sk.ctx._h:
                self.sk.ctx._L.fhs_release(self.sk.ctx._h, self.h)
        except Exception:
            pass

    @staticmethod
    def encrypt_trivial(value, public_parameters, server_key):           # :17-25
        return server_key.trivial(value)

    def _bin(self, name, other):
        L, c = self.sk.ctx._L, self.sk.ctx._h
        return FheAsciiChar(self.sk, getattr(L, "fhs_" + name)(c, self.h, other.h))

    def eq(self, o): return self._bin("eq", o)
    def ne(self, o): return self._bin("ne", o)
    def le(self, o): return self._bin("le", o)
    def lt(self, o): return self._bin("lt", o)
    def ge(self, o): return self._bin("ge", o)
    def gt(self, o): return self._bin("gt", o)
    def bitand(self, o): return self._bin("bitand", o)
    def bitor(self, o): return self._bin("bitor", o)
    def sub(self, o): return self._bin("sub", o)
    def add(self, o): return self._bin("add", o)

    def if_then_else(self, t, f):
        return FheAsciiChar(self.sk, self.sk.ctx._L.fhs_if_then_else(self.sk.ctx._h, self.h, t.h, f.h))

    def _un(self, name):
        return FheAsciiChar(self.sk, getattr(self.sk.ctx._L, "fhs_" + name)(self.sk.ctx._h, self.h))

    def flip(self): return self._un("flip")
    def is_whitespace(self): return self._un("is_whitespace")
    def is_uppercase(self): return self._un("is_uppercase")
    def is_lowercase(self): return self._un("is_lowercase")
    def clone(self): return self._un("clone")

    def sum_c2(self):
        """Noise of the handle in bootstrap-output variances (fhs_char_sum_c2): results come back at <= 4 except find's
        index (<= 57).  The figure stays with the handle across a download; a ciphertext that leaves the library takes
        it along (`download()` + `sum_c2()`), and `set_noise()` declares it after the upload on the other side."""
        v = C.c_uint64(0)
        self.sk.ctx._check(self.sk.ctx._L.fhs_char_sum_c2(self.sk.ctx._h, self.h, C.byref(v)))
        return int(v.value)

    def set_noise(self, sum_c2):
        """fhs_char_set_noise: this uploaded ciphertext is a result that was handed out at `sum_c2` (uploads count as 1)."""
        self.sk.ctx._check(self.sk.ctx._L.fhs_char_set_noise(self.sk.ctx._h, self.h, int(sum_c2)))
        return self

    def trivial_value(self):
        """The byte if every block of the handle is a trivial ciphertext (what constant folding left), else None."""
        t, v = C.c_int(0), C.c_uint8(0)
        self.sk.ctx._check(self.sk.ctx._L.fhs_trivial_value(self.sk.ctx._h, self.h, C.byref(t), C.byref(v)))
        return int(v.value) if t.value else None

    def download(self):
        out = np.zeros((4, BIG_CT), np.uint64)
        self.sk.flush()                      # (collective in level-parallel mode)
        self.sk.ctx._check(self.sk.ctx._L.fhs_download(self.sk.ctx._h, self.h, _ptr(out)))
        return out


class FheString:
    """Vec<FheAsciiChar> (fhestring.rs:6-9)."""

    def __init__(self, chars):
        self.chars = list(chars)

    def __len__(self):
        return len(self.chars)

    def __getitem__(self, i):
        return self.chars[i]

    def download(self):
        """[n][4][2049] words: the whole string with one gather + one copy per 2048 blocks (fhs_download_string)."""
        if not self.chars:
            return np.zeros((0, 4, BIG_CT), np.uint64)
        sk = self.chars[0].sk
        out = np.empty((len(self.chars), 4, BIG_CT), np.uint64)
        sk.ctx._check(sk.ctx._L.fhs_download_string(sk.ctx._h, _harr(self.chars), len(self.chars), _ptr(out)))
        return out


class FheSplit:
    """Vec<FheString> + pattern_found (src/ciphertext/fhesplit.rs:5-8)."""

    def __init__(self, buffers, pattern_found):
        self.buffers = buffers
        self.pattern_found = pattern_found

    @staticmethod
    def decrypt(fhe_split, my_client_key):                               # fhesplit.rs:29-40
        return ([my_client_key.decrypt(b) for b in fhe_split.buffers],
                my_client_key.decrypt_char(fhe_split.pattern_found))


def trim_vector(vec):
    """src/utils.rs:59-92."""
    vec = list(vec)
    while vec and vec[0] == "":
        vec.pop(0)
    while vec and vec[-1] == "":
        vec.pop()
    return vec


def _harr(chars):
    arr = (C.c_uint64 * max(1, len(chars)))()
    for i, ch in enumerate(chars):
        arr[i] = ch.h
    return arr


class MyServerKey:
    """MyServerKey (src/server_key/mod.rs:13-16) on one MI355X."""

    device_resident = True   # ciphertexts live in HBM; export/import take device pointers

    def __init__(self, ctx):
        self.ctx = ctx
        self._stats = None

    @classmethod
    def from_client_key(cls, client_key, device_id=0, arith=0):
        ctx = Context(device_id)
        ctx.set_arithmetic(arith)
        ctx.load_server_key(client_key.bsk(), client_key.ksk())
        if arith in (2, 3):
            ctx.load_multibit_key(client_key.bsk_mb2())
        return cls(ctx)

    @classmethod
    def planner(cls):
        """Planner context (no GPU): DAG statistics of any op -- PBS count, levels, level widths, noise bookkeeping."""
        return cls(Context(planner=True))

    def dummy_string(self, n):
        """n placeholder chars for a planner context (contents are never read)."""
        z = np.zeros((4, BIG_CT), np.uint64)
        return FheString([self.upload_char(z) for _ in range(n)])

    @classmethod
    def from_key_file(cls, path, device_id=0, arith=0, multibit_key_path=None):
        ctx = Context(device_id)
        ctx.set_arithmetic(arith)
        ctx._check(ctx._L.fhs_load_server_key_file(ctx._h, str(path).encode()))
        if multibit_key_path is not None:                      # kind 3 file, needed by arithmetics 2 and 3
            ctx._check(ctx._L.fhs_load_multibit_key_file(ctx._h, str(multibit_key_path).encode()))
        return cls(ctx)

    @classmethod
    def from_raw_keys(cls, bsk, ksk, device_id=0, arith=0, bsk_mb2=None):
        ctx = Context(device_id)
        ctx.set_arithmetic(arith)
        ctx.load_server_key(bsk, ksk)
        if bsk_mb2 is not None:
            ctx.load_multibit_key(bsk_mb2)
        return cls(ctx)

    def close(self):
        self.ctx.close()

    def set_mode(self, mode):
        self.ctx._check(self.ctx._L.fhs_set_mode(self.ctx._h, int(mode)))

    # ---- ciphertext placement ------------------------------------------------
    def trivial(self, v):
        return FheAsciiChar(self, self.ctx._L.fhs_trivial(self.ctx._h, int(v) & 255))

    def upload_char(self, blocks):
        blocks = np.ascontiguousarray(blocks, np.uint64)
        return FheAsciiChar(self, self.ctx._L.fhs_upload(self.ctx._h, _ptr(blocks)))

    def upload_string(self, chars):
        chars = np.ascontiguousarray(chars, np.uint64).reshape(-1, 4, BIG_CT)
        n = chars.shape[0]
        hs = (C.c_uint64 * max(1, n))()
        self.ctx._check(self.ctx._L.fhs_upload_string(self.ctx._h, _ptr(chars), n, hs))
        return FheString([FheAsciiChar(self, hs[i]) for i in range(n)])

    def import_device(self, d_ptr):
        return FheAsciiChar(self, self.ctx._L.fhs_import_device(self.ctx._h, C.c_void_p(d_ptr)))

    def export_device(self, ch, d_ptr):
        self.flush()
        self.ctx._check(self.ctx._L.fhs_export_device(self.ctx._h, ch.h, C.c_void_p(d_ptr)))

    def export_device_async(self, ch, d_ptr):
        """Stream-ordered export (no host wait); pair with stream_handle() to order foreign work after it."""
        self.ctx._check(self.ctx._L.fhs_export_device_async(self.ctx._h, ch.h, C.c_void_p(d_ptr)))

    def stream_handle(self):
        return int(self.ctx._L.fhs_stream_handle(self.ctx._h) or 0)

    def flush(self, wait=True):
        """Run every pending PBS level.  wait=False only enqueues the launches (fhs_flush_async)."""
        if wait:
            self.ctx._check(self.ctx._L.fhs_flush(self.ctx._h))
        else:
            self.ctx._check(self.ctx._L.fhs_flush_async(self.ctx._h))

    def submit(self):
        """fhs_submit: plan what was recorded since the last submit / flush as one job (nothing runs yet)."""
        self.ctx._check(self.ctx._L.fhs_submit(self.ctx._h))

    def pump(self, n_ticks=1):
        """fhs_pump: enqueue the next n ticks (one launch group each over every job's level scheduled for it)."""
        self.ctx._check(self.ctx._L.fhs_pump(self.ctx._h, int(n_ticks)))

    def set_tick_balance(self, slots=None):
        """fhs_set_tick_balance: round-align the launch groups of fhs_submit scheduling (None = the selected kernel's
        resident slots, 0 = off)."""
        if slots is None:
            slots = self.ctx._L.fhs_resident_slots(self.ctx._h)
        self.ctx._check(self.ctx._L.fhs_set_tick_balance(self.ctx._h, int(slots)))
        return int(slots)

    def set_auto_flush(self, n_depth1):
        """fhs_set_auto_flush: peel the ready level once n_depth1 bootstraps of it are recorded (0 = off)."""
        self.ctx._check(self.ctx._L.fhs_set_auto_flush(self.ctx._h, int(n_depth1)))

    def stream_sync(self):
        self.ctx._check(self.ctx._L.fhs_stream_sync(self.ctx._h))

    def enable_level_parallel(self, rank, world, dist, torch):
        """Level-parallel multi-GPU mode inside the library (fhs_dist_level_parallel): every rank records the same
        DAG on the same ciphertexts and runs 1/world of every PBS level; one stream-ordered all-gather per level."""
        from .parallel import Dist
        if world > 1:
            self.dist = Dist.from_torch(self, dist, torch, rank, world)
            self.dist.level_parallel(True)

    def _flags(self, name, flags):
        flags = self._chars(flags)
        out = C.c_uint64()
        self.ctx._check(getattr(self.ctx._L, name)(self.ctx._h, _harr(flags), len(flags), C.byref(out)))
        return FheAsciiChar(self, out.value)

    def flags_or(self, flags): return self._flags("fhs_flags_or", flags)
    def flags_and(self, flags): return self._flags("fhs_flags_and", flags)

    def flags_first_decides(self, any_diff, verdict, tie):
        """Combine compare_partial() results of consecutive ranges: the first range that differs decides."""
        out = C.c_uint64()
        self.ctx._check(self.ctx._L.fhs_flags_first_decides(self.ctx._h, _harr(any_diff), _harr(verdict),
                                                            len(any_diff), int(tie), C.byref(out)))
        return FheAsciiChar(self, out.value)

    def capture_pbs_inputs(self, max_rows_per_level, live=False):
        """fhs_debug_capture_pbs_inputs: sample the PBS inputs of every executed level (0 = off).  live: inside the
        ordinary execution path (rotation sharing and scheduling as in production: fhs_debug_capture_live)."""
        self.ctx._check(self.ctx._L.fhs_debug_capture_live(self.ctx._h, int(bool(live))))
        self.ctx._check(self.ctx._L.fhs_debug_capture_pbs_inputs(self.ctx._h, int(max_rows_per_level)))

    def read_capture(self):
        """-> (rows [n, 2049] u64, records: structured array level/index/lut/n_terms/sum_c2/konst/width)."""
        from ._lib import CaptureRec
        L, h = self.ctx._L, self.ctx._h
        n = C.c_size_t()
        self.ctx._check(L.fhs_debug_capture_read(h, None, None, 0, C.byref(n)))
        rows = np.zeros((n.value, BIG_CT), np.uint64)
        recs = (CaptureRec * max(1, n.value))()
        got = C.c_size_t()
        self.ctx._check(L.fhs_debug_capture_read(h, _ptr(rows), C.cast(recs, C.c_void_p), n.value, C.byref(got)))
        dt = np.dtype([("level", "u4"), ("index", "u4"), ("lut", "u4"), ("n_terms", "u4"), ("sum_c2", "i8"),
                       ("konst", "i4"), ("width", "u4")])
        return rows[:got.value], np.frombuffer(recs, dtype=dt, count=got.value).copy()

    def level_widths(self):
        """Widths of the dependency levels executed since the last stats reset (fhs_level_widths)."""
        n = C.c_size_t()
        self.ctx._check(self.ctx._L.fhs_level_widths(self.ctx._h, None, 0, C.byref(n)))
        out = np.zeros(max(1, n.value), np.uint32)
        self.ctx._check(self.ctx._L.fhs_level_widths(self.ctx._h, _ptr(out), n.value, C.byref(n)))
        return out[:n.value].tolist()

    def launch_groups(self):
        """Rows this rank ran in every launch group since the last stats reset (fhs_launch_groups)."""
        n = C.c_size_t()
        self.ctx._check(self.ctx._L.fhs_launch_groups(self.ctx._h, None, 0, C.byref(n)))
        out = np.zeros(max(1, n.value), np.uint32)
        self.ctx._check(self.ctx._L.fhs_launch_groups(self.ctx._h, _ptr(out), n.value, C.byref(n)))
        return out[:n.value].tolist()

    def stats(self, reset=False):
        from ._lib import Stats
        st = Stats()
        self.ctx._check(self.ctx._L.fhs_get_stats(self.ctx._h, C.byref(st)))
        out = {k: getattr(st, k) for k, _ in Stats._fields_}
        if reset:
            self.ctx._L.fhs_reset_stats(self.ctx._h)
        return out

    # ---- helpers -------------------------------------------------------------
    @staticmethod
    def _chars(x):
        return x.chars if isinstance(x, FheString) else list(x)

    def _flag_op(self, name, s, pat):
        s, pat = self._chars(s), self._chars(pat)
        out = C.c_uint64()
        rc = getattr(self.ctx._L, "fhs_str_" + name)(self.ctx._h, _harr(s), len(s), _harr(pat), len(pat),
                                                      C.byref(out))
        if rc == -4:
            raise OverflowError(self.ctx._L.fhs_last_error(self.ctx._h).decode())
        self.ctx._check(rc)
        return FheAsciiChar(self, out.value)

    def _map_op(self, name, s, n_out=None):
        s = self._chars(s)
        n_out = len(s) if n_out is None else n_out
        out = (C.c_uint64 * max(1, n_out))()
        self.ctx._check(getattr(self.ctx._L, name)(self.ctx._h, _harr(s), len(s), out))
        return FheString([FheAsciiChar(self, out[i]) for i in range(n_out)])

    # ---- the reference's methods (public_parameters accepted and ignored) -----
    def contains(self, string, needle, public_parameters=None):          # mod.rs:151
        return self._flag_op("contains", string, needle)

    def contains_clear(self, string, clear_needle, public_parameters=None):   # mod.rs:198
        s = self._chars(string)
        data = clear_needle.encode("ascii")
        out = C.c_uint64()
        self.ctx._check(self.ctx._L.fhs_str_contains_clear(self.ctx._h, _harr(s), len(s), data, len(data),
                                                           C.byref(out)))
        return FheAsciiChar(self, out.value)

    def starts_with(self, string, pattern, public_parameters=None):      # mod.rs:344
        return self._flag_op("starts_with", string, pattern)

    def ends_with(self, string, needle, public_parameters=None):         # mod.rs:241
        return self._flag_op("ends_with", string, needle)

    def find(self, string, pattern, public_parameters=None):             # mod.rs:1010
        return self._flag_op("find", string, pattern)

    def find_clear(self, string, clear_pattern, public_parameters=None):  # mod.rs:1075
        s = self._chars(string)
        data = clear_pattern.encode("ascii")
        out = C.c_uint64()
        rc = self.ctx._L.fhs_str_find_clear(self.ctx._h, _harr(s), len(s), data, len(data), C.byref(out))
        if rc == -4:
            raise OverflowError(self.ctx._L.fhs_last_error(self.ctx._h).decode())
        self.ctx._check(rc)
        return FheAsciiChar(self, out.value)

    def rfind(self, string, pattern, public_parameters=None):            # mod.rs:727
        return self._flag_op("rfind", string, pattern)

    def is_empty(self, string, public_parameters=None):                  # mod.rs:431
        s = self._chars(string)
        out = C.c_uint64()
        self.ctx._check(self.ctx._L.fhs_str_is_empty(self.ctx._h, _harr(s), len(s), C.byref(out)))
        return FheAsciiChar(self, out.value)

    def len(self, string, public_parameters=None):                       # mod.rs:478
        s = self._chars(string)
        out = C.c_uint64()
        self.ctx._check(self.ctx._L.fhs_str_len(self.ctx._h, _harr(s), len(s), C.byref(out)))
        return FheAsciiChar(self, out.value)

    def eq(self, string, other, public_parameters=None):                 # mod.rs:1122
        return self._flag_op("eq", string, other)

    def ne(self, string, other, public_parameters=None):                 # mod.rs:1178
        return self._flag_op("ne", string, other)

    def eq_ignore_case(self, string, other, public_parameters=None):     # mod.rs:1221
        return self._flag_op("eq_ignore_case", string, other)

    def _compare(self, a, b, cmp):
        a, b = self._chars(a), self._chars(b)
        out = C.c_uint64()
        self.ctx._check(self.ctx._L.fhs_str_compare(self.ctx._h, _harr(a), len(a), _harr(b), len(b), cmp,
                                                    C.byref(out)))
        return FheAsciiChar(self, out.value)

    def compare_partial(self, a, b, cmp):
        """(any position differs, verdict at the first differing position) of two equally long slices:
        the per-GPU partial of a position-sharded lt/le/gt/ge (cmp = 0/1/2/3)."""
        a, b = self._chars(a), self._chars(b)
        d, v = C.c_uint64(), C.c_uint64()
        self.ctx._check(self.ctx._L.fhs_str_compare_partial(self.ctx._h, _harr(a), len(a), _harr(b), len(b), cmp,
                                                            C.byref(d), C.byref(v)))
        return FheAsciiChar(self, d.value), FheAsciiChar(self, v.value)

    def lt(self, a, b, public_parameters=None): return self._compare(a, b, 0)   # mod.rs:1577
    def le(self, a, b, public_parameters=None): return self._compare(a, b, 1)   # mod.rs:1613
    def gt(self, a, b, public_parameters=None): return self._compare(a, b, 2)   # mod.rs:1649
    def ge(self, a, b, public_parameters=None): return self._compare(a, b, 3)   # mod.rs:1685

    def to_upper(self, s, public_parameters=None): return self._map_op("fhs_str_to_upper", s)   # mod.rs:65
    def to_lower(self, s, public_parameters=None): return self._map_op("fhs_str_to_lower", s)   # mod.rs:110
    def trim_end(self, s, public_parameters=None): return self._map_op("fhs_str_trim_end", s)   # trim.rs:36
    def trim_start(self, s, public_parameters=None): return self._map_op("fhs_str_trim_start", s)  # trim.rs:86
    def trim(self, s, public_parameters=None): return self._map_op("fhs_str_trim", s)            # trim.rs:146
    def bubble_zeroes_right(self, s): return self._map_op("fhs_bubble_zeroes_right", s)           # utils.rs:28

    def _replace(self, s, frm, to, n=None):
        s, frm, to = self._chars(s), self._chars(frm), self._chars(to)
        cap = self.ctx._L.fhs_str_replace_len(len(s), len(frm), len(to))
        out = (C.c_uint64 * max(1, cap))()
        n_out = C.c_size_t()
        if n is None:
            rc = self.ctx._L.fhs_str_replace(self.ctx._h, _harr(s), len(s), _harr(frm), len(frm), _harr(to),
                                             len(to), out, cap, C.byref(n_out))
        else:
            rc = self.ctx._L.fhs_str_replacen(self.ctx._h, _harr(s), len(s), _harr(frm), len(frm), _harr(to),
                                              len(to), n.h, out, cap, C.byref(n_out))
        self.ctx._check(rc)
        return FheString([FheAsciiChar(self, out[i]) for i in range(n_out.value)])

    def replace(self, s, frm, to, public_parameters=None): return self._replace(s, frm, to)          # mod.rs:624
    def replacen(self, s, frm, to, n, public_parameters=None): return self._replace(s, frm, to, n)  # mod.rs:1729

    def replace_clear(self, s, clear_from, clear_to, public_parameters=None):                        # mod.rs:679
        return self._replace(s, [self.trivial(b) for b in clear_from.encode("ascii")],
                             [self.trivial(b) for b in clear_to.encode("ascii")])

    def repeat(self, s, n, public_parameters=None):                       # mod.rs:567
        s = self._chars(s)
        n_out = MAX_REPETITIONS * len(s)
        out = (C.c_uint64 * max(1, n_out))()
        self.ctx._check(self.ctx._L.fhs_str_repeat(self.ctx._h, _harr(s), len(s), n.h, out))
        return FheString([FheAsciiChar(self, out[i]) for i in range(n_out)])

    def repeat_clear(self, s, n, public_parameters=None):                 # mod.rs:517
        s = self._chars(s)
        n_out = n * len(s)
        out = (C.c_uint64 * max(1, n_out))()
        self.ctx._check(self.ctx._L.fhs_str_repeat_clear(self.ctx._h, _harr(s), len(s), n, out))
        return FheString([FheAsciiChar(self, out[i]) for i in range(n_out)])

    def concatenate(self, a, b, public_parameters=None):                  # mod.rs:1864
        a, b = self._chars(a), self._chars(b)
        out = (C.c_uint64 * max(1, len(a) + len(b)))()
        self.ctx._check(self.ctx._L.fhs_str_concatenate(self.ctx._h, _harr(a), len(a), _harr(b), len(b), out))
        return FheString([FheAsciiChar(self, out[i]) for i in range(len(a) + len(b))])

    def _strip(self, name, s, pat):
        s, pat = self._chars(s), self._chars(pat)
        out = (C.c_uint64 * max(1, len(s)))()
        found = C.c_uint64()
        self.ctx._check(getattr(self.ctx._L, name)(self.ctx._h, _harr(s), len(s), _harr(pat), len(pat), out,
                                                   C.byref(found)))
        return FheString([FheAsciiChar(self, out[i]) for i in range(len(s))]), FheAsciiChar(self, found.value)

    # ---- split family (src/server_key/split.rs) ---------------------------------
    def _split(self, kind, s, pat=(), n=None):
        s, pat = self._chars(s), self._chars(pat)
        d = self.ctx._L.fhs_str_split_dim(kind, len(s))
        out = (C.c_uint64 * max(1, d * d))()
        dim, found = C.c_size_t(), C.c_uint64()
        self.ctx._check(self.ctx._L.fhs_str_split(self.ctx._h, kind, _harr(s), len(s), _harr(pat), len(pat),
                                                  n.h if n is not None else 0, out, d * d, C.byref(dim),
                                                  C.byref(found)))
        d = dim.value
        bufs = [FheString([FheAsciiChar(self, out[i * d + j]) for j in range(d)]) for i in range(d)]
        return FheSplit(bufs, FheAsciiChar(self, found.value))

    def _clear(self, text):
        return [self.trivial(b) for b in text.encode("ascii")]

    def split(self, s, p, public_parameters=None): return self._split(0, s, p)                         # :989
    def split_clear(self, s, p, public_parameters=None): return self._split(0, s, self._clear(p))
    def split_inclusive(self, s, p, public_parameters=None): return self._split(1, s, p)               # :1020
    def split_inclusive_clear(self, s, p, public_parameters=None): return self._split(1, s, self._clear(p))
    def split_terminator(self, s, p, public_parameters=None): return self._split(2, s, p)              # :1051
    def split_terminator_clear(self, s, p, public_parameters=None): return self._split(2, s, self._clear(p))
    def splitn(self, s, p, n, public_parameters=None): return self._split(3, s, p, n)                  # :1448
    def splitn_clear(self, s, p, n, public_parameters=None): return self._split(3, s, self._clear(p), self.trivial(n))
    def rsplit(self, s, p, public_parameters=None): return self._split(4, s, p)                        # :394
    def rsplit_clear(self, s, p, public_parameters=None): return self._split(4, s, self._clear(p))
    def rsplit_terminator(self, s, p, public_parameters=None): return self._split(5, s, p)             # :504
    def rsplit_terminator_clear(self, s, p, public_parameters=None): return self._split(5, s, self._clear(p))
    def rsplitn(self, s, p, n, public_parameters=None): return self._split(6, s, p, n)                 # :421
    def rsplitn_clear(self, s, p, n, public_parameters=None): return self._split(6, s, self._clear(p), self.trivial(n))
    def rsplit_once(self, s, p, public_parameters=None): return self._split(7, s, p)                   # :462
    def rsplit_once_clear(self, s, p, public_parameters=None): return self._split(7, s, self._clear(p))
    def split_ascii_whitespace(self, s, public_parameters=None): return self._split(8, s)              # :1377

    def strip_prefix(self, s, pat, public_parameters=None): return self._strip("fhs_str_strip_prefix", s, pat)   # mod.rs:1261
    def strip_suffix(self, s, pat, public_parameters=None): return self._strip("fhs_str_strip_suffix", s, pat)   # mod.rs:1335
