"""Which source files make each blind-rotation kernel, and their git blob hashes.

`tools/pmc_to_json.py` records the hashes into profiles/r0N_counters.json when a kernel is profiled; `bench.py` compares
them with the tree it runs from and refuses to quote hardware-counted figures of a kernel whose source changed since
(`roofline.counters_stale`).  The hash is git's blob id (sha1 of "blob <len>\\0" + content), computed
here without git: the GPU box has no repository."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
FLAGS = "Makefile:flags"     # not a file: the compiler-flag lines of the Makefile (CXXFLAGS, ARCH, per-file SCHED_ strategy)
_FFT = ["fft_transform.h", "fft_device.h", "fft_consts.inc", FLAGS]
_NTT = ["ntt_transform.h", "ntt_consts.inc", FLAGS]
KERNEL_SOURCES = {
    "blind_rotate_fft_kernel": ["fft_kernels.hip"] + _FFT,
    "blind_rotate_fft4_kernel": ["fft4_kernels.hip"] + _FFT[1:],
    "blind_rotate_mb2_kernel": ["fftmb_kernels.hip"] + _FFT,
    "blind_rotate_kernel": ["pbs_kernels.hip"] + _NTT,
    "blind_rotate_ntt_mb2_kernel": ["nttmb_kernels.hip"] + _NTT,
    "keyswitch_mfma_kernel": ["ks_kernels.hip", FLAGS],
    "keyswitch_mfma2_kernel": ["ks_kernels.hip", FLAGS],
    "ks_digits_tile_kernel": ["ks_kernels.hip", FLAGS],
}
UNKNOWN = "(kernel not listed in fhestring_amd/kernel_sources.py)"


def blob_hash(data):
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_blobs(kernel, read=None):
    """{file: git blob hash} of the sources of `kernel`; `read(name) -> bytes` defaults to the working tree."""
    if read is None:
        read = lambda name: open(os.path.join(CSRC, name), "rb").read()

    def content(name):
        if name != FLAGS:
            return read(name)
        keep = (b"CXXFLAGS", b"ARCH", b"SCHED_", b"HIPCC")
        return b"\n".join(l for l in read("Makefile").splitlines() if l.startswith(keep))
    return {name: blob_hash(content(name)) for name in KERNEL_SOURCES.get(kernel, [])}


def stale_sources(kernel, recorded):
    """Files of `kernel` whose hash in the tree differs from the recorded one (all of them if nothing was recorded).
    A kernel this module does not know is stale by definition: nothing ties its counters to a source."""
    if kernel not in KERNEL_SOURCES:
        return [UNKNOWN]
    now = source_blobs(kernel)
    if not recorded:
        return sorted(now)
    return sorted(name for name, h in now.items() if recorded.get(name) != h)
