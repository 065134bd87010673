"""Multi-GPU string operations: thin Python glue over the library's own distributed layer (include/fhestring_hip.h
"multi-GPU inside the library": fhs_dist_*).  One process per GPU; the RCCL communicator, the all-gathers (enqueued on
the context's HIP stream) and the partial / combine DAGs all live in C++ (csrc/dist.cpp, capi_dist.cpp, strings.cpp);
this module only hands the 128-byte communicator id around (torch.distributed, any backend) and mirrors the entry points.

* window sharding: contains / find (src/server_key/mod.rs:170-177, :1010-1053) -- a rank holds its slice plus an
  (m-1)-character halo;
* position sharding: eq / eq_ignore_case / lt / le / gt / ge of two equally long padded buffers (:1122-1149, :1470-1541);
* level-parallel mode for everything else (replace with its compaction, ...): identical DAGs, every PBS level split.
"""
import ctypes as C

import numpy as np

from ._lib import lib

CHAR_WORDS = 4 * 2049
ID_BYTES = 128


def plan_windows(n_chars, m, world):
    """[(w0, w1, c0, c1)] per rank (fhs_dist_plan_windows): rank r evaluates windows [w0, w1) and holds chars [c0, c1)."""
    L = lib()
    out = []
    for r in range(world):
        v = [C.c_size_t() for _ in range(4)]
        L.fhs_dist_plan_windows(n_chars, m, world, r, *[C.byref(x) for x in v])
        out.append(tuple(int(x.value) for x in v))
    return out


def plan_positions(n_chars, world):
    """[(c0, c1)] per rank (fhs_dist_plan_positions)."""
    L = lib()
    out = []
    for r in range(world):
        a, b = C.c_size_t(), C.c_size_t()
        L.fhs_dist_plan_positions(n_chars, world, r, C.byref(a), C.byref(b))
        out.append((int(a.value), int(b.value)))
    return out


def unique_id():
    buf = C.create_string_buffer(ID_BYTES)
    rc = lib().fhs_dist_unique_id(buf)
    if rc != 0:
        raise RuntimeError("fhs_dist_unique_id failed (%d): is librccl.so.1 loadable?" % rc)
    return buf.raw


class Dist:
    """The distributed side of one MyServerKey (one context = one communicator = one HIP stream)."""

    def __init__(self, sk, rank, world):
        self.sk, self.rank, self.world = sk, rank, world
        self._cb = None
        self.transport = "none"

    # ---- set-up ---------------------------------------------------------------------------------------
    @classmethod
    def from_torch(cls, sk, dist, torch, rank=None, world=None):
        """Communicator id from rank 0 through torch.distributed (any backend).  backend nccl -> the library's own RCCL
        communicator (one rank per GPU, xGMI); any other backend -> host transport through that backend (ranks sharing
        one GPU in tests: RCCL refuses two ranks on one device)."""
        rank = dist.get_rank() if rank is None else rank
        world = dist.get_world_size() if world is None else world
        self = cls(sk, rank, world)
        if world == 1 and dist is None:
            return self
        if dist.get_backend() == "nccl":
            # the library's own communicator (ncclCommInitRank on this context's device).  ncclCommInitRank is
            # collective, so the ranks first agree -- through torch.distributed -- that EVERY one of them can load
            # librccl (fhs_dist_available: dlopen + dlsym only) and that rank 0 produced an id; only then do all of them
            # enter it.  If one cannot, all of them fall back to a host transport carried by torch.distributed
            # (slower: the gathered blocks take a detour through host memory; `transport` says which one runs, and
            # bench.py refuses to report a multi-GPU number measured on the fallback).
            ok = int(sk.ctx._L.fhs_dist_available() == 1)
            box = [None]
            if ok and rank == 0:
                try:
                    box = [unique_id()]
                except RuntimeError:
                    ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                dist.broadcast_object_list(box, src=0)
                ok = int(sk.ctx._L.fhs_dist_init(sk.ctx._h, rank, world, box[0]) == 0)
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                self.transport = "rccl"
            else:
                if ok and sk.ctx._L.fhs_dist_world(sk.ctx._h) > 1:
                    # ncclCommInitRank succeeded HERE and failed on another rank: the communicator is half-formed, a
                    # destroy may wait for peers that never joined -- abort it (no exchange with anyone)
                    sk.ctx._check(sk.ctx._L.fhs_dist_abort(sk.ctx._h))
                self.init_host_transport(lambda send: _torch_device_all_gather(dist, torch, send, world))
                self.transport = "torch.distributed (fallback: the library's RCCL communicator did not come up)"
        else:
            self.init_host_transport(lambda send: _gloo_all_gather(dist, torch, send, world))
            self.transport = "host transport over torch.distributed/%s" % dist.get_backend()
        return self

    def init_single(self):
        """world = 1 with a real RCCL communicator (exercises the stream-ordered path on one GPU)."""
        self.sk.ctx._check(self.sk.ctx._L.fhs_dist_init(self.sk.ctx._h, 0, 1, unique_id()))
        self.transport = "rccl"
        return self

    def init_host_transport(self, all_gather_bytes):
        """all_gather_bytes(send: bytes-like of k bytes) -> bytes-like of world * k bytes, rank-major."""
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)

        def cb(_user, send, recv, nbytes):
            try:
                data = all_gather_bytes(C.string_at(send, nbytes))
                C.memmove(recv, bytes(data), nbytes * self.world)
                return 0
            except Exception:                   # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1

        self._cb = CB(cb)                       # keep the trampoline alive as long as the context uses it
        if self.transport == "none":
            self.transport = "host transport (caller's all-gather)"
        self.sk.ctx._check(self.sk.ctx._L.fhs_dist_init_host_transport(self.sk.ctx._h, self.rank, self.world,
                                                                       self._cb, None))
        return self

    def shutdown(self):
        self.sk.ctx._check(self.sk.ctx._L.fhs_dist_shutdown(self.sk.ctx._h))

    def stats(self):
        """fhs_dist_stats: all-gathers issued by this context, bytes this rank contributed, the transport in use."""
        n, b, tr = C.c_uint64(), C.c_uint64(), C.c_int()
        self.sk.ctx._check(self.sk.ctx._L.fhs_dist_stats(self.sk.ctx._h, C.byref(n), C.byref(b), C.byref(tr)))
        return {"allgather_calls": int(n.value), "bytes_sent": int(b.value),
                "transport": {0: "none", 1: "rccl", 2: "host"}[int(tr.value)]}

    def level_parallel(self, on=True):
        self.sk.ctx._check(self.sk.ctx._L.fhs_dist_level_parallel(self.sk.ctx._h, int(bool(on))))

    # ---- slices ---------------------------------------------------------------------------------------
    def window_shard(self, ck, full_string, m, padding=1):
        """Encrypt and upload only this rank's slice (+ halo) of `full_string` + padding NULs.
        -> (FheString shard, global index of its first window, total chars incl. padding)"""
        n_chars = len(full_string) + padding
        w0, w1, c0, c1 = plan_windows(n_chars, m, self.world)[self.rank]
        text = full_string[c0:min(c1, len(full_string))]
        pad_here = max(0, c1 - max(c0, len(full_string)))
        return ck.encrypt(text, pad_here, None, self.sk), w0, n_chars

    def position_shard(self, ck, full_string, n_chars):
        """This rank's slice of `full_string` padded with NULs to n_chars positions."""
        c0, c1 = plan_positions(n_chars, self.world)[self.rank]
        text = full_string[c0:min(c1, len(full_string))]
        return ck.encrypt(text, (c1 - c0) - len(text), None, self.sk)

    # ---- sharded entry points (C) -------------------------------------------------------------------------
    def _h(self, chars):
        from .api import _harr
        chars = self.sk._chars(chars)
        return _harr(chars), len(chars), chars

    def _char(self, h):
        from .api import FheAsciiChar
        return FheAsciiChar(self.sk, h)

    def allgather_flags(self, flags):
        """n flag chars per rank -> [[rank 0's n], [rank 1's n], ...] on every rank; ONE all-gather."""
        arr, n, keep = self._h(flags)
        out = (C.c_uint64 * (self.world * max(1, n)))()
        self.sk.ctx._check(self.sk.ctx._L.fhs_dist_allgather_flags(self.sk.ctx._h, arr, n, out))
        return [[self._char(out[r * n + i]) for i in range(n)] for r in range(self.world)]

    def allgather_chars(self, chars):
        arr, n, keep = self._h(chars)
        out = (C.c_uint64 * (self.world * max(1, n)))()
        self.sk.ctx._check(self.sk.ctx._L.fhs_dist_allgather_chars(self.sk.ctx._h, arr, n, out))
        return [[self._char(out[r * n + i]) for i in range(n)] for r in range(self.world)]

    def contains(self, shard, pattern):
        out = C.c_uint64()
        sa, sn, k1 = self._h(shard)
        L, h = self.sk.ctx._L, self.sk.ctx._h
        if isinstance(pattern, str):
            p = pattern.encode()
            self.sk.ctx._check(L.fhs_dist_str_contains_clear(h, sa, sn, p, len(p), C.byref(out)))
        else:
            pa, pn, k2 = self._h(pattern)
            self.sk.ctx._check(L.fhs_dist_str_contains(h, sa, sn, pa, pn, C.byref(out)))
        return self._char(out.value)

    def find(self, shard, pattern, first_window, total_chars):
        out = C.c_uint64()
        sa, sn, k1 = self._h(shard)
        L, h = self.sk.ctx._L, self.sk.ctx._h
        if isinstance(pattern, str):
            p = pattern.encode()
            rc = L.fhs_dist_str_find_clear(h, sa, sn, p, len(p), first_window, total_chars, C.byref(out))
        else:
            pa, pn, k2 = self._h(pattern)
            rc = L.fhs_dist_str_find(h, sa, sn, pa, pn, first_window, total_chars, C.byref(out))
        if rc == -4:
            raise OverflowError(L.fhs_last_error(h).decode())
        self.sk.ctx._check(rc)
        return self._char(out.value)

    def eq(self, a_shard, b_shard, ignore_case=False):
        out = C.c_uint64()
        aa, an, k1 = self._h(a_shard)
        ba, bn, k2 = self._h(b_shard)
        self.sk.ctx._check(self.sk.ctx._L.fhs_dist_str_eq(self.sk.ctx._h, aa, an, ba, bn, int(ignore_case), C.byref(out)))
        return self._char(out.value)

    def eq_ignore_case(self, a_shard, b_shard):
        return self.eq(a_shard, b_shard, True)

    def compare(self, a_shard, b_shard, op):
        out = C.c_uint64()
        aa, an, k1 = self._h(a_shard)
        ba, bn, k2 = self._h(b_shard)
        cmp = {"lt": 0, "le": 1, "gt": 2, "ge": 3}[op]
        self.sk.ctx._check(self.sk.ctx._L.fhs_dist_str_compare(self.sk.ctx._h, aa, an, ba, bn, cmp, C.byref(out)))
        return self._char(out.value)

    def contains_batch(self, shards, clear_pattern):
        """contains_clear on several independent strings with ONE exchange: the local flags of all strings are
        evaluated in one DAG flush, gathered with one all-gather of len(shards) blocks per rank and OR-ed in one
        level that every rank evaluates."""
        sk = self.sk
        local = [sk.contains_clear(sh, clear_pattern) if len(sh) >= len(clear_pattern) else sk.trivial(0)
                 for sh in shards]
        if self.world == 1 and not self._force:
            return local
        parts = self.allgather_flags(local)
        return [sk.flags_or([parts[r][i] for r in range(self.world)]) for i in range(len(local))]

    _force = False


def _torch_device_all_gather(dist, torch, send, world):
    t = torch.frombuffer(bytearray(send), dtype=torch.uint8).cuda()
    out = torch.empty(world * t.numel(), dtype=torch.uint8, device="cuda")
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy().tobytes()


def _gloo_all_gather(dist, torch, send, world):
    t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return b"".join(p.numpy().tobytes() for p in parts)
