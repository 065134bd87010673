"""Thin Python mirror of the C ABI (include/fhestring_hip.h)."""
import ctypes as C

import numpy as np

from ._lib import FhsError, lib

BIG_CT = 2049
SMALL_CT = 743
POLY_N = 2048


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """fhs_ctx: one per GPU.  Replaces the tfhe::integer::ServerKey held by
    MyServerKey (src/server_key/mod.rs:13-16)."""

    def __init__(self, device_id=0):
        self._L = lib()
        h = C.c_void_p()
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

    def load_server_key(self, bsk, ksk):
        bsk = np.ascontiguousarray(bsk, np.uint64)
        ksk = np.ascontiguousarray(ksk, np.uint64)
        assert bsk.size == 742 * 4 * 2048 and ksk.size == 2048 * 5 * 743
        self._check(self._L.fhs_load_server_key(self._h, _ptr(bsk), _ptr(ksk)))

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
        self._check(self._L.fhs_kernel_timing(self._h, int(reset), C.byref(br), C.byref(ks), C.byref(nb),
                                              C.byref(nk), C.byref(units)))
        return {"blind_rotate_ms": br.value, "keyswitch_ms": ks.value, "n_blind_rotate": nb.value,
                "n_keyswitch": nk.value, "pbs_in_launches": units.value}
