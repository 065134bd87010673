"""Multi-GPU sharding of one string operation (one process per GPU, RCCL via torch.distributed).

contains(): the match windows of a string are independent (src/server_key/mod.rs:170-177), so
they are split into `world` contiguous ranges; a rank only ever holds the characters its windows
touch (its slice plus an (m-1)-character halo).  The only exchange is one all-gather of one
FheAsciiChar (the partial flag, 65 568 B) per rank, followed by one OR level that every rank
evaluates redundantly (so every rank ends with the result, like the reference's return value).
"""
import numpy as np

CHAR_WORDS = 4 * 2049


def plan_windows(n_chars, m, world):
    """Split windows 0..n_chars-m over `world` ranks.

    Returns a list of (w0, w1, c0, c1): rank r evaluates windows [w0, w1) and needs characters
    [c0, c1).  Ranks beyond the number of windows get an empty range.  n_chars includes padding.
    """
    n_win = max(0, n_chars - m + 1) if m <= n_chars else 0
    base, extra = divmod(n_win, world)
    out, w = [], 0
    for r in range(world):
        cnt = base + (1 if r < extra else 0)
        w0, w1 = w, w + cnt
        c0, c1 = (w0, w1 + m - 1) if cnt else (0, 0)
        out.append((w0, w1, c0, min(c1, n_chars)))
        w = w1
    return out


class _FlagExchange:
    """Gathers one partial FheAsciiChar flag per item from every rank: parts[item][rank]."""

    def _stream_ordered(self):
        import os
        t = self.torch
        return (t.cuda.is_available() and self.sk.device_resident and self.dist.get_backend() == "nccl"
                and not os.environ.get("FHS_SYNC_EXCHANGE"))

    def _gather(self, local):
        sk, torch, n = self.sk, self.torch, len(local)
        if self._stream_ordered():
            # one all-gather of n chars per rank on the context's own HIP stream: no host synchronisation anywhere,
            # so consecutive batches on different contexts keep overlapping
            if getattr(self, "_ext", None) is None:
                self._ext = torch.cuda.ExternalStream(sk.stream_handle())
            with torch.cuda.stream(self._ext):
                mine = torch.empty(n * CHAR_WORDS, dtype=torch.int64, device="cuda")
                allp = torch.empty(self.world * n * CHAR_WORDS, dtype=torch.int64, device="cuda")
                for k, l in enumerate(local):
                    sk.export_device_async(l, mine.data_ptr() + 8 * CHAR_WORDS * k)   # first call flushes the DAG
                self.dist.all_gather_into_tensor(allp, mine)    # RCCL over xGMI: world x n x 65 568 B
            self._keep = (mine, allp)       # device buffers stay referenced until the next exchange replaces them
        else:
            # host-synchronised path: CPU stand-in, several ranks sharing one GPU through gloo (tests), fallback
            dev = "cuda" if torch.cuda.is_available() and sk.device_resident else "cpu"
            mine = torch.empty(n * CHAR_WORDS, dtype=torch.int64, device=dev)
            for k, l in enumerate(local):
                sk.export_device(l, mine.data_ptr() + 8 * CHAR_WORDS * k)             # flushes this rank's DAG
            allp = torch.empty(self.world * n * CHAR_WORDS, dtype=torch.int64, device=dev)
            if dev == "cuda" and self.dist.get_backend() != "nccl":
                parts_cpu = [torch.empty(n * CHAR_WORDS, dtype=torch.int64) for _ in range(self.world)]
                self.dist.all_gather(parts_cpu, mine.cpu())
                allp.copy_(torch.cat(parts_cpu))
            else:
                self.dist.all_gather_into_tensor(allp, mine)
            if dev == "cuda":
                torch.cuda.synchronize()
            self._keep = (mine, allp)
        return [[sk.import_device(allp.data_ptr() + 8 * CHAR_WORDS * (r * n + k)) for r in range(self.world)]
                for k in range(n)]


class ShardedContains(_FlagExchange):
    def __init__(self, sk, rank, world, dist, torch):
        self.sk, self.rank, self.world, self.dist, self.torch = sk, rank, world, dist, torch

    def upload_shard(self, ck, full_string, chars_per_rank, m, padding=1):
        """Encrypt and upload only this rank's slice (+halo) of `full_string` + padding NULs."""
        n_chars = len(full_string) + padding
        w0, w1, c0, c1 = plan_windows(n_chars, m, self.world)[self.rank]
        text = full_string[c0:min(c1, len(full_string))]
        pad_here = max(0, c1 - max(c0, len(full_string)))
        return ck.encrypt(text, pad_here, None, self.sk)

    def _local(self, shard, clear_pattern):
        if len(shard) >= len(clear_pattern):
            return self.sk.contains_clear(shard, clear_pattern)
        return self.sk.trivial(0)          # this rank owns no window

    def run(self, shard, clear_pattern, op="contains"):
        return self.run_batch([shard], clear_pattern, op=op)[0]

    def run_batch(self, shards, clear_pattern, op="contains", force_exchange=False):
        """contains() on several independent strings with ONE exchange: the local flags of all strings are
        evaluated in one DAG flush, gathered with one all-gather of len(shards) chars per rank and OR-ed in one
        level (every rank evaluates it, so every rank ends with the result, like the reference's return value)."""
        sk = self.sk
        if op == "find":
            if self.world != 1:
                raise NotImplementedError("find has no window-sharded form: use LevelParallel")
            return [sk.find_clear(sh, clear_pattern) for sh in shards]
        local = [self._local(sh, clear_pattern) for sh in shards]
        if self.world == 1 and not force_exchange:          # force_exchange: 1-rank RCCL test
            return local
        return [sk.flags_or(parts) for parts in self._gather(local)]


class ShardedEq(_FlagExchange):
    """eq / eq_ignore_case of two padded strings of the SAME buffer length (BASELINE config 5): the character
    positions are split into `world` contiguous ranges and a rank holds only its slice of both strings.  On
    well-formed padded strings (NULs only at the end) the reference's semantics (src/server_key/mod.rs:1122-1149:
    every position equal or both NUL, and equal lengths) is the conjunction of the same predicate over the slices,
    so each rank evaluates the op on its slices and the partial flags are AND-ed after one all-gather of one
    FheAsciiChar per rank."""

    def __init__(self, sk, rank, world, dist, torch):
        self.sk, self.rank, self.world, self.dist, self.torch = sk, rank, world, dist, torch

    @staticmethod
    def plan(n_chars, world):
        base, extra = divmod(n_chars, world)
        out, c = [], 0
        for r in range(world):
            cnt = base + (1 if r < extra else 0)
            out.append((c, c + cnt))
            c += cnt
        return out

    def upload_shard(self, ck, full_string, n_chars):
        """This rank's slice of `full_string` padded with NULs to n_chars positions."""
        c0, c1 = self.plan(n_chars, self.world)[self.rank]
        text = full_string[c0:min(c1, len(full_string))]
        return ck.encrypt(text, (c1 - c0) - len(text), None, self.sk)

    def run(self, a_shard, b_shard, op="eq", force_exchange=False):
        sk = self.sk
        fn = {"eq": sk.eq, "eq_ignore_case": sk.eq_ignore_case}[op]
        local = fn(a_shard, b_shard) if len(a_shard) else sk.trivial(1)
        if self.world == 1 and not force_exchange:
            return local
        return sk.flags_and(self._gather([local])[0])


class ShardedCmp(ShardedEq):
    """lt / le / gt / ge of two padded strings of the SAME buffer length with the character positions split over
    the ranks (BASELINE config 5, `<=`).  Each rank reduces its slices to (some position differs, verdict at the
    first differing position) -- the positional half of src/server_key/mod.rs:1497-1518 -- the two flags of every
    rank are gathered with one all-gather, and the first range that differs decides; when nothing differs the
    buffers, hence the strings, are equal (le / ge -> 1, lt / gt -> 0).  NUL padding compares below every
    character, which is exactly the reference's length tie-break."""

    def run(self, a_shard, b_shard, op="le", force_exchange=False):
        sk = self.sk
        cmp = {"lt": 0, "le": 1, "gt": 2, "ge": 3}[op]
        if len(a_shard):
            d, v = sk.compare_partial(a_shard, b_shard, cmp)
        else:
            d, v = sk.trivial(0), sk.trivial(0)
        if self.world == 1 and not force_exchange:
            ds, vs = [d], [v]
        else:
            ds, vs = self._gather([d, v])
        return sk.flags_first_decides(ds, vs, 1 if op in ("le", "ge") else 0)


class LevelParallel:
    """Generic multi-GPU execution of ANY op: the ranks hold the same ciphertexts, record the same DAG
    and split every PBS level; one all-gather of the level's outputs (width x 16 392 B) per level.
    Works for replace / compare / find / ... without op-specific partial results (the window sharding
    above moves less data and is what bench.py uses for contains)."""

    def __init__(self, sk, rank, world, dist, torch):
        self.sk, self.rank, self.world, self.dist, self.torch = sk, rank, world, dist, torch
        self._send = self._recv = None

    def _buffers(self, cap):
        torch = self.torch
        words = cap * 2049
        if self._send is None or self._send.numel() < words:
            self._send = torch.empty(words, dtype=torch.int64, device="cuda")
            self._recv = torch.empty(self.world * words, dtype=torch.int64, device="cuda")
        return self._send[:words], self._recv[:self.world * words]

    def flush(self):
        import ctypes as C
        L, h, torch = self.sk.ctx._L, self.sk.ctx._h, self.torch
        n_levels, max_w = C.c_uint64(), C.c_uint64()
        self.sk.ctx._check(L.fhs_flush_plan(h, C.byref(n_levels), C.byref(max_w)))
        if n_levels.value == 0:
            return
        cap_max = (max_w.value + self.world - 1) // self.world
        self._buffers(cap_max)
        for k in range(n_levels.value):
            width, cap = C.c_uint64(), C.c_uint64()
            self.sk.ctx._check(L.fhs_flush_level_exec(h, k, C.c_void_p(self._send.data_ptr()), C.byref(width),
                                                      C.byref(cap)))
            self.sk.ctx._check(L.fhs_stream_sync(h))
            send, recv = self._buffers(cap.value)
            if self.dist.get_backend() != "nccl":             # tests: ranks sharing one GPU, CPU group
                parts = [torch.empty(send.numel(), dtype=torch.int64) for _ in range(self.world)]
                self.dist.all_gather(parts, send.cpu())
                recv.copy_(torch.cat(parts))
            else:
                self.dist.all_gather_into_tensor(recv, send)   # RCCL over xGMI
            torch.cuda.synchronize()
            self.sk.ctx._check(L.fhs_flush_level_commit(h, k, C.c_void_p(recv.data_ptr())))
        self.sk.ctx._check(L.fhs_stream_sync(h))
