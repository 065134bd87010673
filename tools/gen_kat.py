#!/usr/bin/env python3
"""Known-answer digests of the PBS layer (last row) -> tests/golden/pbs_kat.json.  CPU only.

Ground truth = oracle mode 1: the SCHOOLBOOK negacyclic product modulo 2^64 (no transform at all) inside the full
bootstrap (keyswitch -> modulus switch -> 742 CMUX -> sample extract), under the fixed seed
0xF5E57121, for all 32 block values (the padding bit set included) x {msg, eq_biv, sign}.  Per output the file keeps the
SHA-256 of its 2049 little-endian u64 words, the first and last word, and the decrypted value; plus digests of the
keys and of the 32 input ciphertexts, so that a mismatch can be attributed (keygen / encryption / bootstrap).

The f64-FFT arithmetic (the kernel `value` is measured on) is deterministic but not exact: its digests come from
oracle mode 3 (the lane-for-lane C mirror of the kernel) AS IT WAS WHEN THIS FILE WAS GENERATED -- a later simultaneous
edit of kernel and mirror that changes any bit no longer passes silently.  Likewise modes 4 / 5 (two key bits per
external product).  Regenerate only when an arithmetic change is intended, and say so in the commit:

    python tools/gen_kat.py            # ~2 minutes on 8 cores (the schoolbook bootstrap is 12 G multiply-adds per PBS)
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED = 0xF5E57121
LUT_NAMES = ("msg", "eq_biv", "sign")
OUT = os.path.join(ROOT, "tests", "golden", "pbs_kat.json")
SHIFT_INPUTS = [0, 5, 9, 15, 22, 31]          # block values whose ciphertexts (kat_inputs) are bootstrapped once each ...
SHIFT_LIST = [0, 1, 7, 16, 27, 31]            # ... and extracted at these shifts (message units)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, "<u8").tobytes()).hexdigest()


def kat_inputs():
    """keys, the 32 input ciphertexts (block values 0..31 encrypted in this order by a fresh Keys(SEED)), LUT polynomials"""
    from oracle import core, radix
    K = core.Keys(SEED)
    cts = np.stack([K.encrypt_block(m) for m in range(32)])
    luts = np.stack([radix.lut_poly(n) for n in LUT_NAMES])
    # batch row r = 32 * lut + m
    rows = np.concatenate([cts] * len(LUT_NAMES))
    idx = np.repeat(np.arange(len(LUT_NAMES), dtype=np.uint32), 32)
    return K, cts, luts, rows, idx


def digests(out):
    return [{"sha256": sha(o), "first": int(o[0]), "last": int(o[-1])} for o in out]


def main():
    from oracle import core, radix
    t0 = time.time()
    K, cts, luts, rows, idx = kat_inputs()
    S = core.ServerKey(K)
    S.set_mb2(K.bsk_mb2)
    nt = os.cpu_count() or 1
    rec = {"seed": "0x%X" % SEED, "luts": list(LUT_NAMES), "row_order": "row = 32 * lut_index + block_value",
           "generator": "tools/gen_kat.py", "params": "PARAM_MESSAGE_2_CARRY_2_KS_PBS (SURVEY.md Appendix A)",
           "keys": {"lwe_sk": sha(K.lwe_sk), "glwe_sk": sha(K.glwe_sk), "bsk": sha(K.bsk), "ksk": sha(K.ksk),
                    "bsk_mb2": sha(K.bsk_mb2)},
           "lut_polys": {n: sha(luts[i]) for i, n in enumerate(LUT_NAMES)},
           "inputs": digests(cts)}
    ks = np.stack([S.keyswitch_modswitch(c) for c in cts])
    rec["keyswitch_modswitch"] = [hashlib.sha256(np.ascontiguousarray(k, "<u4").tobytes()).hexdigest() for k in ks]
    truth = S.pbs_batch(rows, idx, luts, nt, mode=1)
    print("schoolbook ground truth: %.0f s" % (time.time() - t0), file=sys.stderr)
    want = [radix.lut_eval(LUT_NAMES[int(idx[r])], r % 32) for r in range(len(rows))]
    got = [K.decrypt_block(o) for o in truth]
    assert got == want, "the schoolbook bootstrap does not decrypt to the look-up values"
    rec["exact"] = {"source": "oracle mode 1 (schoolbook product modulo 2^64)", "decrypts_to": want, "outputs": digests(truth)}
    assert np.array_equal(S.pbs_batch(rows, idx, luts, nt, mode=0), truth), "Goldilocks NTT != schoolbook"
    for mode, name, what in ((3, "f64_fft_mirror", "oracle mode 3: C mirror of blind_rotate_fft_kernel / blind_rotate_fft4_kernel"),
                             (4, "f64_fft_mb2_mirror", "oracle mode 4: C mirror of blind_rotate_mb2_kernel (pair key)"),
                             (5, "exact_mb2", "oracle mode 5: exact two-bit bootstrap on the 57-bit key grid (pair key)")):
        out = S.pbs_batch(rows, idx, luts, nt, mode=mode)
        assert [K.decrypt_block(o) for o in out] == want, name
        rec[name] = {"source": what, "outputs": digests(out)}
    # rotation sharing: ONE blind rotation, several sample extractions (orc_pbs_shifted; the product's
    # fhs_pbs_batch_shifted and the engine's shared rows) -- schoolbook ground truth, and the f64 mirror's digests
    sh_rows, sh_shifts = SHIFT_INPUTS, SHIFT_LIST
    for mode, name in ((1, "shifted_exact"), (3, "shifted_f64_fft_mirror")):
        outs = [S.pbs_shifted(cts[m], luts[0], sh_shifts, mode=mode) for m in sh_rows]
        for m, o in zip(sh_rows, outs):
            assert [K.decrypt_block(x) for x in o] == [radix.lut_eval("msg", (m + t) & 31) for t in sh_shifts], (name, m)
        if mode == 1:
            assert all(np.array_equal(o, S.pbs_shifted(cts[m], luts[0], sh_shifts, mode=0)) for m, o in zip(sh_rows, outs))
        rec[name] = {"lut": "msg", "inputs": sh_rows, "shifts": sh_shifts,
                     "outputs": [digests(o) for o in outs]}
    with open(OUT, "w") as f:
        json.dump(rec, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote %s (%d bytes) in %.0f s" % (OUT, os.path.getsize(OUT), time.time() - t0), file=sys.stderr)


if __name__ == "__main__":
    main()
