"""Oracle string layer -- TEST INFRASTRUCTURE ONLY.

Restatement, loop for loop, of the reference's `MyServerKey` string algorithms
(src/server_key/mod.rs, src/server_key/trim.rs, src/server_key/split.rs,
src/utils.rs:28-112) on top of an abstract FheAsciiChar (oracle/radix.py:
ClearChar or CipherChar).  Every function cites the reference lines it follows.

Strings are python lists of chars (FheString.bytes, fhestring.rs:6-9); the case
delta `cst` = trivial 32 (fhestring.rs:24).
"""
MAX_FIND_LENGTH = 255   # src/main.rs:20
MAX_REPETITIONS = 16    # src/main.rs:17
STRING_PADDING = 1      # src/main.rs:12


class Ops:
    """Binds a char model (ClearChar / CipherChar) and its context."""

    def __init__(self, char_cls, ctx=None):
        self.C = char_cls
        self.ctx = ctx

    def t(self, v):
        """FheAsciiChar::encrypt_trivial (fheasciichar.rs:17-25)."""
        return self.C.trivial(v, self.ctx)

    # ---- char predicates (fheasciichar.rs:106-168) -------------------------
    def is_whitespace(self, c):
        r = None
        for w in (0x20, 0x09, 0x0A, 0x0B, 0x0C, 0x0D):      # :112-129
            e = c.eq(self.t(w))
            r = e if r is None else r.bitor(e)
        return r

    def is_uppercase(self, c):                               # :132-144
        return c.ge(self.t(0x41)).bitand(c.le(self.t(0x5A)))

    def is_lowercase(self, c):                               # :146-158
        return c.ge(self.t(0x61)).bitand(c.le(self.t(0x7A)))

    # ---- utils.rs ----------------------------------------------------------
    def bubble_zeroes_right(self, s):                        # utils.rs:28-46
        s = list(s)
        zero = self.t(0)
        for _ in range(len(s)):
            for i in range(len(s) - 1):
                swap = s[i].eq(zero)
                s[i] = swap.if_then_else(s[i + 1], s[i])     # :40 (old s[i+1])
                s[i + 1] = swap.if_then_else(zero, s[i + 1])  # :41
        return s

    @staticmethod
    def adjust_end_of_pattern(e):                            # utils.rs:106-112
        return 1 if e == 0 else e

    # ---- mod.rs ------------------------------------------------------------
    def to_upper(self, s):                                   # mod.rs:65-84
        zero, cst = self.t(0), self.t(32)
        return [b.sub(self.is_lowercase(b).flip().if_then_else(zero, cst)) for b in s]

    def to_lower(self, s):                                   # mod.rs:110-128
        zero, cst = self.t(0), self.t(32)
        return [b.add(self.is_uppercase(b).flip().if_then_else(zero, cst)) for b in s]

    def contains(self, s, needle):                           # mod.rs:151-182
        if not s and not needle:
            return self.t(1)
        if len(needle) > len(s):
            return self.t(0)
        result, one = self.t(0), self.t(1)
        for i in range(len(s) - len(needle) + 1):
            cur = one
            for j, nc in enumerate(needle):
                cur = cur.bitand(s[i + j].eq(nc))
            result = result.bitor(cur)
        return result

    def clear(self, text):
        """the `_clear` twins trivially encrypt the pattern (e.g. mod.rs:198-211)."""
        return [self.t(b) for b in text.encode("ascii")]

    def ends_with(self, s, needle):                          # mod.rs:241-288
        if not s and not needle:
            return self.t(1)
        if len(needle) > len(s):
            return self.t(0)
        result, one, zero = self.t(0), self.t(1), self.t(0)
        for i in range(len(s) - len(needle) + 1):
            cur, nonzero = one, one
            for j, nc in enumerate(needle):
                cur = cur.bitand(s[i + j].eq(nc))
                nonzero = nonzero.bitand(s[i + j].ne(zero))
            result = nonzero.if_then_else(cur, result)
        return result

    def starts_with(self, s, pattern):                       # mod.rs:344-371
        if len(pattern) > len(s):
            return self.t(0)
        if not s and not pattern:
            return self.t(1)
        result = self.t(1)
        for sc, pc in zip(s[:min(len(pattern), len(s))], pattern):
            result = result.bitand(sc.eq(pc))
        return result

    def is_empty(self, s):                                   # mod.rs:431-451
        if not s:
            return self.t(1)
        zero, result = self.t(0), self.t(1)
        for c in s:
            result = result.bitand(c.eq(zero))
        return result

    def len(self, s):                                        # mod.rs:478-493
        zero = self.t(0)
        if not s:
            return zero
        result = self.t(0)
        for c in s:
            result = result.add(c.ne(zero))
        return result

    def repeat_clear(self, s, n):                            # mod.rs:517-538
        if n == 0:
            return []
        return self.bubble_zeroes_right(list(s) * n)

    def repeat(self, s, n_enc):                              # mod.rs:567-591
        zero = self.t(0)
        result = [zero] * (MAX_REPETITIONS * len(s))
        for i in range(MAX_REPETITIONS):
            flag = self.t(i).lt(n_enc)
            for j in range(len(s)):
                result[i * len(s) + j] = flag.if_then_else(s[j], zero)
        return self.bubble_zeroes_right(result)

    def replace(self, s, frm, to):                           # mod.rs:624-653
        n = self.t(0)
        if len(frm) >= len(to):
            return self._longer_from(s, frm, to, n, False)
        return self._shorter_from(s, frm, to, n, False)

    def replacen(self, s, frm, to, n):                       # mod.rs:1729-1757
        if len(frm) >= len(to):
            return self._longer_from(s, frm, to, n, True)
        return self._shorter_from(s, frm, to, n, True)

    def _longer_from(self, s, frm, to, n, use_counter):      # mod.rs:828-882
        zero, one = self.t(0), self.t(1)
        data = list(s) + [zero]                              # :841
        to = list(to) + [zero] * (len(frm) - len(to))        # :847-849
        counter = self.t(0)
        result = list(data)
        if len(frm) <= len(result):
            for i in range(self.adjust_end_of_pattern(len(result) - len(frm))):
                flag = one
                for j in range(len(frm)):
                    flag = flag.bitand(frm[j].eq(data[i + j]))   # reads the copy (:864)
                if use_counter:                               # :868-872
                    counter = counter.add(flag)
                    flag = flag.bitand(n.ge(counter))
                for k in range(len(to)):
                    result[i + k] = flag.if_then_else(to[k], result[i + k])
        return self.bubble_zeroes_right(result)              # :881

    def _shorter_from(self, s, frm, to, n, use_counter):     # mod.rs:885-980
        zero, one = self.t(0), self.t(1)
        data = list(s) + [zero]                              # :898
        size_diff = abs(len(frm) - len(to))
        counter = self.t(0)
        max_len = len(to) * len(data) + len(data) if data else len(to)   # :903-907
        if not frm:
            max_len = (len(data) + (len(data) + 1) * len(to)) + 1        # :910-914
        result = list(data) + [zero] * (max_len - len(data))
        copy_buffer = [zero] * max_len
        ignore_mask = [one] * max_len
        for i in range(len(result) - len(to)):               # :929
            flag = one
            for j in range(len(frm)):
                flag = flag.bitand(frm[j].eq(result[i + j]))
                flag = flag.bitand(ignore_mask[i + j])
            if not frm:                                      # :941-947
                flag = one if i % (len(to) + 1) == 0 else zero
            if use_counter:
                counter = counter.add(flag)
                flag = flag.bitand(n.ge(counter))
            for k in range(max_len):                         # :957-959
                copy_buffer[k] = flag.if_then_else(result[k], zero)
            for k in range(len(to)):                         # :962-968
                result[i + k] = flag.if_then_else(to[k], result[i + k])
                ignore_mask[i + k] = ignore_mask[i + k].bitand(flag.if_then_else(zero, one))
            for k in range(i + len(to), max_len):            # :971-977
                result[k] = flag.if_then_else(copy_buffer[k - size_diff], result[k])
        return result

    def rfind(self, s, pattern):                             # mod.rs:727-790
        one, zero = self.t(1), self.t(0)
        s = list(s) + [zero]                                 # :737
        pos = self.t(MAX_FIND_LENGTH)
        if len(s) >= MAX_FIND_LENGTH + len(pattern):         # :742-744
            raise OverflowError("Maximum supported size for find reached")
        if not pattern:                                      # :747-760
            last = zero
            for i in range(len(s)):
                last = s[i].ne(zero).if_then_else(self.t(i + 1), last)
            return last
        if len(pattern) > len(s):
            return self.t(255)
        for i in range(self.adjust_end_of_pattern(len(s) - len(pattern))):
            flag = one
            for j, pc in enumerate(pattern):
                flag = flag.bitand(pc.eq(s[i + j]))
            pos = flag.if_then_else(self.t(i), pos)
        return pos

    def find(self, s, pattern):                              # mod.rs:1010-1053
        if not s and not pattern:
            return self.t(0)
        one = self.t(1)
        pos = self.t(MAX_FIND_LENGTH)
        if len(s) >= MAX_FIND_LENGTH + len(pattern):         # :1025-1027
            raise OverflowError("Maximum supported size for find reached")
        if len(pattern) > len(s):
            return self.t(255)
        for i in reversed(range(len(s) - len(pattern) + 1)):
            flag = one
            for j in reversed(range(len(pattern))):
                flag = flag.bitand(pattern[j].eq(s[i + j]))
            pos = flag.if_then_else(self.t(i), pos)
        return pos

    def eq(self, a, b):                                      # mod.rs:1122-1149
        zero, one = self.t(0), self.t(1)
        is_eq = one
        len_ne = self.len(a).ne(self.len(b))
        for i in range(min(len(a), len(b))):
            same = a[i].eq(b[i])
            both0 = a[i].eq(zero).bitand(b[i].eq(zero))
            is_eq = is_eq.bitand(both0.bitor(same))
        return len_ne.if_then_else(zero, is_eq)

    def ne(self, a, b):                                      # mod.rs:1178-1186
        return self.eq(a, b).flip()

    def eq_ignore_case(self, a, b):                          # mod.rs:1221-1231
        return self.eq(self.to_lower(a), self.to_lower(b))

    def strip_prefix(self, s, pattern):                      # mod.rs:1261-1307
        zero, one = self.t(0), self.t(1)
        result = list(s)
        flag = one
        end = min(len(pattern), len(result))
        if len(pattern) > len(result):
            return result, zero
        if end == 0:
            if not pattern:
                flag = one
            elif pattern and not s:
                flag = zero
        for j in range(end):
            flag = flag.bitand(pattern[j].eq(result[j]))
        for j in range(min(len(pattern), len(result))):
            result[j] = flag.if_then_else(zero, result[j])
        return self.bubble_zeroes_right(result), flag

    def strip_suffix(self, s, needle):                       # mod.rs:1335-1404
        one, zero, t255 = self.t(1), self.t(0), self.t(255)
        s = list(s)
        if len(needle) > len(s):
            return s, zero
        end = len(s) - len(needle)
        pos = self.t(255)
        for i in range(end + 1):
            found, nonzero = one, one
            for j, nc in enumerate(needle):
                found = found.bitand(s[i + j].eq(nc))
                nonzero = nonzero.bitand(s[i + j].ne(zero))
            cur = found.if_then_else(self.t(i), t255)
            pos = nonzero.if_then_else(cur, pos)
        should_strip = pos.ne(t255)
        for i in range(end + 1):
            mask = self.t(i).eq(pos)
            for j in range(len(needle)):
                s[i + j] = mask.if_then_else(zero, s[i + j])
        return s, should_strip

    def comparison(self, a, b, op):                          # mod.rs:1470-1541
        zero, t255 = self.t(0), self.t(255)
        a, b = list(a), list(b)
        min_len = min(len(a), len(b))
        seen, became, ret = zero, zero, self.t(255)
        if min_len == 0:                                     # :1490-1494
            a.append(zero)
            b.append(zero)
            min_len = 1
        for i in range(min_len):
            cmp = getattr(a[i], op)(b[i])                    # :1497-1502
            seen = seen.bitor(a[i].ne(b[i]))                 # :1504-1506
            flag = seen.bitand(became.flip())                # :1508-1511
            became = became.bitor(flag)                      # :1512
            ret = flag.if_then_else(cmp, ret)                # :1513
        sub_eq = ret.eq(t255)                                # :1518
        l1, l2 = self.len(a), self.len(b)
        leq, lgt, llt = l1.eq(l2), l1.gt(l2), l1.lt(l2)
        by_len = {"ge": lambda: leq.bitor(lgt), "le": lambda: leq.bitor(llt),
                  "gt": lambda: lgt, "lt": lambda: llt}[op]()            # :1526-1531
        return sub_eq.if_then_else(by_len, ret)              # :1538

    def lt(self, a, b): return self.comparison(a, b, "lt")   # mod.rs:1577
    def le(self, a, b): return self.comparison(a, b, "le")   # mod.rs:1613
    def gt(self, a, b): return self.comparison(a, b, "gt")   # mod.rs:1649
    def ge(self, a, b): return self.comparison(a, b, "ge")   # mod.rs:1685

    def concatenate(self, a, b):                             # mod.rs:1864-1875
        return self.bubble_zeroes_right(list(a) + list(b))

    # ---- trim.rs -----------------------------------------------------------
    def trim_end(self, s):                                   # trim.rs:36-57
        zero = self.t(0)
        stop = zero
        result = [zero] * len(s)
        for i in reversed(range(len(s))):
            not_ws = self.is_whitespace(s[i]).flip()
            stop = stop.bitor(not_ws.bitand(s[i].ne(zero)))
            result[i] = stop.if_then_else(s[i], zero)
        return result

    def trim_start(self, s):                                 # trim.rs:86-115
        zero = self.t(0)
        stop = zero
        result = [zero] * len(s)
        for i in range(len(s)):
            not_ws = self.is_whitespace(s[i]).flip()
            stop = stop.bitor(not_ws.bitand(s[i].ne(zero)))
            result[i] = stop.if_then_else(s[i], zero)
        return self.bubble_zeroes_right(result)

    def trim(self, s):                                       # trim.rs:146-149
        return self.trim_start(self.trim_end(s))


# ---- split family (src/server_key/split.rs); returns (buffers, pattern_found) like FheSplit ----
class SplitOps(Ops):
    def _rsplit_pattern_matching(self, i, s, pattern, mask, zero, one):        # split.rs:10-67
        found = one
        if not pattern:
            cur_pad = s[i].eq(zero)
            if i >= 1:
                prev_nonpad = s[i - 1].ne(zero)
                found = prev_nonpad.bitand(cur_pad).if_then_else(one, zero)
                found = found.bitor(cur_pad.if_then_else(zero, one))
            else:
                found = cur_pad.if_then_else(zero, one)
        elif len(pattern) > len(s) or i + len(pattern) >= len(s):
            found = zero
        else:
            for j, pc in enumerate(pattern):
                found = found.bitand(s[i + j].eq(pc))
                found = found.bitand(mask[i + j])
        for j in range(len(pattern)):                                         # :58-65
            if i + j < len(s):
                mask[i + j] = mask[i + j].bitand(found.if_then_else(zero, one))
        return found

    def _split_pattern_matching(self, i, s, pattern, mask, zero, one):         # split.rs:69-108
        found = one
        if len(pattern) > len(s) or i < len(pattern) - 1:
            found = zero
        else:
            for j, pc in enumerate(pattern):
                k = i - len(pattern) + 1 + j
                found = found.bitand(s[k].eq(pc))
                found = found.bitand(mask[k])
        for j in range(len(pattern)):                                         # :99-105
            if i + j < len(s):
                mask[i + j] = mask[i + j].bitand(found.if_then_else(zero, one))
        return found

    def _copy_logic(self, i, n, s, result, allow_copying, cur_buf):            # split.rs:110-134
        for j in range(len(result)):
            flag = self.t(j).eq(cur_buf)
            if n is not None:
                flag = flag.bitand(allow_copying)
            result[j][i] = flag.if_then_else(s[i], result[j][i])

    def _handle_n_case(self, found, n, cur_buf, stop_inc, one):                # split.rs:136-173
        if n is None:
            return found.if_then_else(cur_buf.add(one), cur_buf), stop_inc
        stop_inc = stop_inc.bitor(cur_buf.eq(n.sub(one)))
        cur_buf = found.bitand(stop_inc.flip()).if_then_else(cur_buf.add(one), cur_buf)
        return cur_buf, stop_inc

    def _clear_pattern_from_result(self, n, result, pattern, zero, one, inclusive, terminator):   # split.rs:175-305
        nb = len(result)
        to = [self.t(0)] * len(pattern)
        if n is not None:
            stop = zero
            for i in range(nb):
                stop = stop.bitor(n.eq(self.t(i).add(one)))
                cur = self.bubble_zeroes_right(result[i])
                rep = self.replace(cur, pattern, to)
                for j in range(nb):
                    result[i][j] = stop.if_then_else(cur[j], rep[j])
            return
        if not inclusive:
            for i in range(nb):
                result[i] = self.replace(result[i], pattern, to)
        else:
            for i in range(nb):
                result[i] = self.bubble_zeroes_right(result[i])
        if terminator:                                                         # :266-302
            nonzero_found = zero
            for i in reversed(range(nb)):
                is_zero = one
                for j in range(nb):
                    is_zero = is_zero.bitand(result[i][j].eq(zero))
                starts = self.starts_with(result[i], pattern)
                delete = starts.bitand(is_zero).bitand(nonzero_found.flip())
                for j in range(nb):
                    result[i][j] = delete.if_then_else(zero, result[i][j])
                nonzero_found = nonzero_found.bitor(is_zero.flip())

    def _xsplit(self, s, pattern, inclusive, terminator, n, reverse):          # _rsplit :307-393, _split :883-988
        zero, one = self.t(0), self.t(1)
        s = list(s) + [zero]
        size = len(s)
        cur_buf, stop_inc = zero, zero
        result = [[zero] * size for _ in range(size)]
        global_found = zero
        allow_copying = zero
        mask = [one] * size
        if n is not None:
            allow_copying = n.ne(zero)
        if not reverse and not pattern and n is not None:                      # :925-937
            enc_len = self.len(s)
            skip = n.gt(one).bitand(n.le(enc_len))
            cur_buf = skip.if_then_else(self.t(1), cur_buf)
        order = reversed(range(size)) if reverse else range(size)
        for i in order:
            self._copy_logic(i, n, s, result, allow_copying, cur_buf)
            if reverse:
                found = self._rsplit_pattern_matching(i, s, pattern, mask, zero, one)
            else:
                found = self._split_pattern_matching(i, s, pattern, mask, zero, one)
            global_found = global_found.bitor(found)
            cur_buf, stop_inc = self._handle_n_case(found, n, cur_buf, stop_inc, one)
        self._clear_pattern_from_result(n, result, pattern, zero, one, inclusive, terminator)
        return result, global_found

    def rsplit(self, s, p): return self._xsplit(s, p, False, False, None, True)               # :394
    def rsplitn(self, s, p, n): return self._xsplit(s, p, False, False, n, True)             # :421
    def rsplit_once(self, s, p): return self._xsplit(s, p, False, False, self.t(2), True)    # :462
    def rsplit_terminator(self, s, p): return self._xsplit(s, p, False, True, None, True)    # :504
    def split(self, s, p): return self._xsplit(s, p, False, False, None, False)              # :989
    def split_inclusive(self, s, p): return self._xsplit(s, p, True, False, None, False)     # :1020
    def split_terminator(self, s, p): return self._xsplit(s, p, False, True, None, False)    # :1051
    def splitn(self, s, p, n): return self._xsplit(s, p, False, False, n, False)             # :1448

    def split_ascii_whitespace(self, s):                                       # split.rs:1377-1447
        zero, one = self.t(0), self.t(1)
        size = len(s)
        cur_buf = zero
        result = [[zero] * size for _ in range(size)]
        prev_ws = self.t(1)
        global_found = zero
        for i in range(size):
            found = self.is_whitespace(s[i])
            global_found = global_found.bitor(found)
            inc = found.bitand(prev_ws.flip())
            cur_buf = inc.if_then_else(cur_buf.add(one), cur_buf)
            for j in range(size):
                flag = self.t(j).eq(cur_buf).bitand(self.is_whitespace(s[i]).flip())
                result[j][i] = flag.if_then_else(s[i], result[j][i])
            prev_ws = found
        for j in range(size):
            for k in range(size):
                c = result[j][k]
                result[j][k] = self.is_whitespace(c).if_then_else(zero, c)
        for j in range(size):
            result[j] = self.bubble_zeroes_right(result[j])
        return result, global_found


def trim_vector(vec):
    """utils.rs:59-92: drop leading and trailing empty strings (what the split tests compare)."""
    vec = list(vec)
    while vec and vec[0] == "":
        vec.pop(0)
    while vec and vec[-1] == "":
        vec.pop()
    return vec


# ---- client side (src/client_key.rs:45-106) ---------------------------------
def pad_plain(text, padding):
    """MyClientKey::encrypt's plaintext: ascii, no NUL, then `padding` NULs (:45-58)."""
    assert all(ord(ch) < 128 and ch != "\0" for ch in text), \
        "The input string must only contain ascii letters and not include null characters"
    return list(text.encode("ascii")) + [0] * padding


def truncate_plain(byte_values):
    """MyClientKey::decrypt truncates at the first NUL (:89-106)."""
    out = []
    for b in byte_values:
        if b == 0:
            break
        out.append(b)
    return bytes(out).decode("ascii")
