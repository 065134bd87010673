"""ctypes binding of oracle/tfhe_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (fhestring_amd/) never does.  See tfhe_oracle.c for
the parity status ("ciphertext-level parity with tfhe-rs unpinned").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

LWE_N = 742
POLY_N = 2048
BIG_N = 2048
BIG_CT = BIG_N + 1
SMALL_CT = LWE_N + 1
KS_LEVEL = 5
DELTA_LOG = 59
BSK_WORDS = LWE_N * 4 * POLY_N
KSK_WORDS = BIG_N * KS_LEVEL * SMALL_CT

_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so with gcc (no GPU needed)."""
    src = os.path.join(_HERE, "tfhe_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_keygen.argtypes = [C.c_uint64, _u64p, _u64p, _u64p, _u64p]
        L.orc_keygen.restype = None
        L.orc_bsk_quantize.argtypes = [_u64p, C.c_uint64]
        L.orc_bsk_quantize.restype = None
        L.orc_server_key_new.argtypes = [_u64p, _u64p]
        L.orc_server_key_new.restype = C.c_void_p
        L.orc_keygen_mb2.argtypes = [C.c_uint64, _u64p, _u64p, _u64p]
        L.orc_keygen_mb2.restype = None
        L.orc_server_key_set_mb2.argtypes = [C.c_void_p, _u64p]
        L.orc_server_key_set_mb2.restype = None
        L.orc_server_key_free.argtypes = [C.c_void_p]
        L.orc_server_key_free.restype = None
        L.orc_encrypt_block.argtypes = [_u64p, C.c_uint64, C.POINTER(C.c_uint64), _u64p]
        L.orc_encrypt_block.restype = None
        L.orc_phase.argtypes = [_u64p, _u64p]
        L.orc_phase.restype = C.c_uint64
        L.orc_decrypt_block.argtypes = [_u64p, _u64p]
        L.orc_decrypt_block.restype = C.c_uint64
        L.orc_make_lut.argtypes = [_u64p, _u64p]
        L.orc_make_lut.restype = None
        L.orc_keyswitch_modswitch.argtypes = [C.c_void_p, _u64p, _u32p]
        L.orc_keyswitch_modswitch.restype = None
        L.orc_keyswitch.argtypes = [C.c_void_p, _u64p, _u64p]
        L.orc_keyswitch.restype = None
        L.orc_pbs.argtypes = [C.c_void_p, _u64p, _u64p, _u64p, C.c_int]
        L.orc_pbs.restype = None
        L.orc_pbs_shifted.argtypes = [C.c_void_p, _u64p, _u64p, _u32p, C.c_uint64, _u64p, C.c_int]
        L.orc_pbs_shifted.restype = None
        L.orc_blind_rotate.argtypes = [C.c_void_p, _u32p, _u64p, _u64p, C.c_int]
        L.orc_blind_rotate.restype = None
        L.orc_pbs_batch.argtypes = [C.c_void_p, _u64p, _u32p, _u64p, _u64p, C.c_uint64, C.c_int, C.c_int]
        L.orc_pbs_batch.restype = None
        L.orc_negacyclic_schoolbook.argtypes = [_i64p, _u64p, _u64p]
        L.orc_negacyclic_schoolbook.restype = None
        L.orc_negacyclic_ntt.argtypes = [_i64p, _u64p, _u64p]
        L.orc_negacyclic_ntt.restype = None
        _f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        L.orc_fft_tables.argtypes = [_f64p, _f64p, _f64p, _f64p]
        L.orc_fft_tables.restype = None
        _lib = L
    return _lib


class Keys:
    """Secret keys + server key material as flat numpy u64 arrays."""

    def __init__(self, seed):
        self.seed = int(seed)
        self.lwe_sk = np.zeros(LWE_N, np.uint64)
        self.glwe_sk = np.zeros(BIG_N, np.uint64)
        self.bsk = np.zeros(BSK_WORDS, np.uint64)
        self.ksk = np.zeros(KSK_WORDS, np.uint64)
        lib().orc_keygen(self.seed, self.lwe_sk, self.glwe_sk, self.bsk, self.ksk)
        self._rng = C.c_uint64((self.seed * 0x9E3779B97F4A7C15 + 0x1234567) & (2**64 - 1))

    @property
    def bsk_mb2(self):
        """Pair key of mode 4 (two key bits per external product), generated on first use: [371 * 3 * 4 * 2048]."""
        if getattr(self, "_bsk_mb2", None) is None:
            self._bsk_mb2 = np.zeros((LWE_N // 2) * 3 * 4 * BIG_N, np.uint64)
            lib().orc_keygen_mb2(self.seed, self.lwe_sk, self.glwe_sk, self._bsk_mb2)
        return self._bsk_mb2

    # client side -----------------------------------------------------------
    def encrypt_block(self, m):
        ct = np.zeros(BIG_CT, np.uint64)
        lib().orc_encrypt_block(self.glwe_sk, int(m) & 31, C.byref(self._rng), ct)
        return ct

    def decrypt_block(self, ct):
        return int(lib().orc_decrypt_block(self.glwe_sk, np.ascontiguousarray(ct, np.uint64)))

    def phase(self, ct):
        return int(lib().orc_phase(self.glwe_sk, np.ascontiguousarray(ct, np.uint64)))

    def encrypt_char(self, v):
        """u8 -> [4, 2049] (little-endian 2-bit blocks; fheasciichar.rs:27-29)."""
        return np.stack([self.encrypt_block((int(v) >> (2 * b)) & 3) for b in range(4)])

    def decrypt_char(self, ct4):
        v = 0
        for b in range(4):
            v += (self.decrypt_block(ct4[b]) & 15) << (2 * b)
        return v & 255


class ServerKey:
    def __init__(self, keys_or_bsk, ksk=None):
        if ksk is None:
            bsk, ksk = keys_or_bsk.bsk, keys_or_bsk.ksk
        else:
            bsk = keys_or_bsk
        self._h = lib().orc_server_key_new(np.ascontiguousarray(bsk, np.uint64),
                                           np.ascontiguousarray(ksk, np.uint64))

    def set_mb2(self, bsk_mb2):
        """orc_server_key_set_mb2: pair key for mode 4."""
        lib().orc_server_key_set_mb2(self._h, np.ascontiguousarray(bsk_mb2, np.uint64))
        return self

    def __del__(self):
        try:
            if self._h:
                lib().orc_server_key_free(self._h)
                self._h = None
        except Exception:
            pass

    def keyswitch_modswitch(self, ct):
        out = np.zeros(SMALL_CT, np.uint32)
        lib().orc_keyswitch_modswitch(self._h, np.ascontiguousarray(ct, np.uint64), out)
        return out

    def keyswitch(self, ct):
        out = np.zeros(SMALL_CT, np.uint64)
        lib().orc_keyswitch(self._h, np.ascontiguousarray(ct, np.uint64), out)
        return out

    def blind_rotate(self, ms, lut, mode=0):
        acc = np.zeros(2 * POLY_N, np.uint64)
        lib().orc_blind_rotate(self._h, np.ascontiguousarray(ms, np.uint32),
                               np.ascontiguousarray(lut, np.uint64), acc, mode)
        return acc

    def pbs(self, ct, lut, mode=0):
        out = np.zeros(BIG_CT, np.uint64)
        lib().orc_pbs(self._h, np.ascontiguousarray(ct, np.uint64),
                      np.ascontiguousarray(lut, np.uint64), out, mode)
        return out

    def pbs_shifted(self, ct, lut, shifts, mode=0):
        """orc_pbs_shifted: ONE blind rotation, one sample extraction per shift: [len(shifts), 2049], row s = what a
        bootstrap of (ct + shifts[s] * Delta) would give (message units, 0..31)."""
        shifts = np.ascontiguousarray(shifts, np.uint32)
        out = np.zeros((len(shifts), BIG_CT), np.uint64)
        lib().orc_pbs_shifted(self._h, np.ascontiguousarray(ct, np.uint64), np.ascontiguousarray(lut, np.uint64),
                              shifts, len(shifts), out, mode)
        return out

    def pbs_batch(self, cts, lut_idx, luts, nthreads=None, mode=0):
        cts = np.ascontiguousarray(cts, np.uint64).reshape(-1, BIG_CT)
        B = cts.shape[0]
        out = np.zeros((B, BIG_CT), np.uint64)
        if B == 0:
            return out
        if nthreads is None:
            nthreads = os.cpu_count() or 1
        lib().orc_pbs_batch(self._h, cts, np.ascontiguousarray(lut_idx, np.uint32),
                            np.ascontiguousarray(luts, np.uint64).reshape(-1, POLY_N), out, B,
                            int(nthreads), mode)
        return out


def fft_tables():
    """Twiddle tables of mode 3 (the mirror of the product's f64-FFT kernel): W[1024] re/im, U[16] re/im."""
    w_re, w_im, u_re, u_im = np.zeros(1024), np.zeros(1024), np.zeros(16), np.zeros(16)
    lib().orc_fft_tables(w_re, w_im, u_re, u_im)
    return w_re, w_im, u_re, u_im


def make_lut(f):
    """LUT polynomial for f: Z16 -> block value (callable or 16-sequence)."""
    tab = np.array([(f(x) if callable(f) else f[x]) & 31 for x in range(16)], np.uint64)
    out = np.zeros(POLY_N, np.uint64)
    lib().orc_make_lut(tab, out)
    return out


def trivial_block(m):
    ct = np.zeros(BIG_CT, np.uint64)
    ct[BIG_N] = np.uint64((int(m) & 31) << DELTA_LOG)
    return ct


def negacyclic_schoolbook(d, b):
    res = np.zeros(POLY_N, np.uint64)
    lib().orc_negacyclic_schoolbook(np.ascontiguousarray(d, np.int64), np.ascontiguousarray(b, np.uint64), res)
    return res


def negacyclic_ntt(d, b):
    res = np.zeros(POLY_N, np.uint64)
    lib().orc_negacyclic_ntt(np.ascontiguousarray(d, np.int64), np.ascontiguousarray(b, np.uint64), res)
    return res
