"""ctypes loader of the in-tree HIP shared library (fhestring_amd/libfhestring_hip.so).

There is no CPU fallback: if the library is missing this raises, and every
entry point fails loudly when no MI355X is visible (fhs_ctx_create returns an
error that FhsError carries).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfhestring_hip.so")

_lib = None


class FhsError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FhsError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


def _declare(L):
    vp, sz, i, u64, u8 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.c_uint8
    L.fhs_ctx_create.argtypes = [i, C.POINTER(vp)]
    L.fhs_ctx_create.restype = i
    L.fhs_ctx_destroy.argtypes = [vp]
    L.fhs_ctx_destroy.restype = None
    L.fhs_last_error.argtypes = [vp]
    L.fhs_last_error.restype = C.c_char_p
    L.fhs_load_server_key.argtypes = [vp, vp, vp]
    L.fhs_load_server_key.restype = i
    L.fhs_pbs_batch.argtypes = [vp, vp, vp, vp, sz, vp, sz]
    L.fhs_pbs_batch.restype = i
    L.fhs_keyswitch_modswitch_batch.argtypes = [vp, vp, vp, sz]
    L.fhs_keyswitch_modswitch_batch.restype = i
    L.fhs_pbs_batch_device.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    L.fhs_pbs_batch_device.restype = i
    L.fhs_kernel_timing.argtypes = [vp, i, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.fhs_kernel_timing.restype = i
    for name, args, res in _OPTIONAL:
        if hasattr(L, name):
            f = getattr(L, name)
            f.argtypes = args
            f.restype = res


_OPTIONAL = []
