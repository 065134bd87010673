"""TEST INFRASTRUCTURE (like everything under oracle/): replays a plan trace of the product's planner on the CPU oracle.

A planner context (fhs_ctx_create_planner: host logic only, no device) records a string operation exactly as a device
context would -- fused DAG, rotation-sharing groups, launch groups -- and with fhs_debug_plan_trace writes what it WOULD
run (include/fhestring_hip.h "plan trace").  `PlanRun` feeds real ciphertexts through that list with the oracle's
bootstrap (oracle/tfhe_oracle.c: any mode; 6 = the AVX2 + FMA f64-FFT port that bench.py reports as cpu_baseline) on host
threads, linear combinations in numpy (wrapping u64).  Two uses:

  * tests/test_plan_exec.py: the fused DAGs, executed WITHOUT any HIP kernel on real ciphertexts, decrypt like Python --
    the string layer's re-association is checked independently of the GPU arithmetic;
  * bench.py cpu_baseline: BASELINE config 3 (find, encrypted pattern, 256 characters: 2 574 bootstraps in 6 levels) run to
    completion on the host cores, the time a CPU takes for the SAME DAG the GPU runs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this; the product never imports oracle/.
"""
import ctypes as C
import time

import numpy as np

BIG_CT = 2049
DELTA_LOG = 59
TR_UPLOAD, TR_ROW, TR_EXT, TR_GROUP_END = 1, 2, 3, 4


class PlanRun:
    def __init__(self, oracle_server_key, threads, mode=6):
        from fhestring_amd.api import MyServerKey
        self.S, self.threads, self.mode = oracle_server_key, threads, mode
        self.sk = MyServerKey.planner()              # the product's own host logic
        self.sk.set_mode(1)
        self.L, self.h = self.sk.ctx._L, self.sk.ctx._h
        self.sk.ctx._check(self.L.fhs_debug_plan_trace(self.h, 1))
        self.uploads = []                            # rows in upload order
        self.store = {}                              # block token -> [2049] u64
        self.n_up = 0
        self._luts = {}
        self.pbs = self.groups = 0
        self.pbs_seconds = 0.0

    def close(self):
        self.sk.close()

    # ---- inputs: real ciphertexts, handed to the planner as placeholders of the same shape -------------------------
    def upload_string(self, chars):
        chars = np.ascontiguousarray(chars, np.uint64).reshape(-1, 4, BIG_CT)
        self.uploads.extend(chars.reshape(-1, BIG_CT))
        return self.sk.upload_string(chars)

    def lut_poly(self, lut_id):
        if lut_id not in self._luts:
            out = np.zeros(2048, np.uint64)
            assert self.L.fhs_debug_lut_poly(int(lut_id), out.ctypes.data) == 0
            self._luts[lut_id] = out
        return self._luts[lut_id]

    # ---- replay --------------------------------------------------------------------------------------------------
    def _read_trace(self):
        n = C.c_size_t()
        self.sk.ctx._check(self.L.fhs_debug_plan_read(self.h, None, 0, C.byref(n)))
        buf = np.zeros(max(1, n.value), np.uint64)
        self.sk.ctx._check(self.L.fhs_debug_plan_read(self.h, buf.ctypes.data, n.value, C.byref(n)))
        return buf[:n.value]

    def _lin(self, konst, terms):
        acc = np.zeros(BIG_CT, np.uint64)
        for tok, coef in terms:
            acc += self.store[tok] * np.uint64(coef & 0xFFFFFFFFFFFFFFFF)
        acc[BIG_CT - 1:] += np.uint64((konst & 31) << DELTA_LOG)             # (array add: wraps silently)
        return acc

    def run(self):
        """execute everything recorded since the last run() (the planner flushes inside fhs_debug_plan_read)"""
        t = self._read_trace().tolist()
        i, rows, by_out = 0, [], {}
        while i < len(t):
            tag = t[i]
            if tag == TR_UPLOAD:
                self.store[t[i + 1]] = self.uploads[self.n_up]
                self.n_up += 1
                i += 2
            elif tag == TR_ROW:
                out, lut, konst, n = t[i + 1:i + 5]
                terms = [(t[i + 5 + 2 * k], t[i + 6 + 2 * k]) for k in range(n)]
                by_out[out] = len(rows)
                rows.append((out, lut, self._lin(konst, terms)))
                i += 5 + 2 * n
            elif tag == TR_EXT:                       # a shared extraction = its leader's bootstrap with konst + K / 128
                lead, out, K = t[i + 1:i + 4]
                _, lut, x = rows[by_out[lead]]
                y = x.copy()
                y[BIG_CT - 1:] += np.uint64(((K // 128) & 31) << DELTA_LOG)
                rows.append((out, lut, y))
                i += 4
            elif tag == TR_GROUP_END:
                if rows:
                    ids = sorted({r[1] for r in rows})
                    luts = np.stack([self.lut_poly(k) for k in ids])
                    idx = np.array([ids.index(r[1]) for r in rows], np.uint32)
                    t0 = time.perf_counter()
                    res = self.S.pbs_batch(np.stack([r[2] for r in rows]), idx, luts, self.threads, mode=self.mode)
                    self.pbs_seconds += time.perf_counter() - t0
                    for r, ct in zip(rows, res):
                        self.store[r[0]] = ct
                    self.pbs += len(rows)
                    self.groups += 1
                rows, by_out = [], {}
                i += 2
            else:
                raise ValueError("bad plan trace word %d at %d" % (tag, i))
        assert not rows

    def result_char(self, ch):
        """[4, 2049] ciphertext of a result handle (after run())"""
        n = C.c_size_t()
        self.sk.ctx._check(self.L.fhs_debug_char_terms(self.h, ch.h, None, 0, C.byref(n)))
        buf = np.zeros(n.value, np.uint64)
        self.sk.ctx._check(self.L.fhs_debug_char_terms(self.h, ch.h, buf.ctypes.data, n.value, C.byref(n)))
        t, i, out = buf.tolist(), 0, np.zeros((4, BIG_CT), np.uint64)
        for blk in range(4):
            kind, val, n_t = t[i:i + 3]
            terms = [(t[i + 3 + 2 * k], t[i + 4 + 2 * k]) for k in range(n_t)]
            i += 3 + 2 * n_t
            if kind == 0:
                out[blk, BIG_CT - 1] = np.uint64((val & 31) << DELTA_LOG)
            elif kind == 1:
                out[blk] = self.store[terms[0][0]]
            elif kind == 2:
                out[blk] = self._lin(val, terms)
            else:
                raise ValueError("result block still pending")
        return out
