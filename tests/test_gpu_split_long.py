"""Split family beyond the reference's u8 buffer index: the fused distribution phase numbers the buffers
with multi-digit prefix counts, so strings of more than 255 characters split correctly (the reference's
`encrypt_trivial(j as u8)` index, src/server_key/split.rs:125, wraps there).  Checked against Python on a 300-character
string; the reference's own 9 split vectors run in tests/test_gpu_ops.py (both modes) and tests/test_planner.py."""
import random

import pytest

pytestmark = pytest.mark.gpu
SEED = 0xF5E57121


@pytest.fixture(scope="module")
def product():
    from fhestring_amd.api import MyClientKey
    ck = MyClientKey(SEED)
    sk = ck.get_server_key(0, arith=1)
    sk.set_mode(1)
    yield ck, sk
    sk.close()
    ck.close()


def _trim(vec):
    vec = list(vec)
    while vec and vec[0] == "":
        vec.pop(0)
    while vec and vec[-1] == "":
        vec.pop()
    return vec


def _text(rnd, n, sep, pieces):
    """n characters, `pieces` separators at well separated places, no other occurrence of sep's first character"""
    body = [chr(rnd.choice([c for c in range(0x21, 0x7F) if chr(c) not in sep])) for _ in range(n)]
    step = n // (pieces + 1)
    for k in range(1, pieces + 1):
        at = k * step + rnd.randint(-3, 3)
        body[at:at + len(sep)] = sep
    return "".join(body)[:n]


def test_split_300_chars_vs_python(product):
    ck, sk = product
    rnd = random.Random(300)
    s = _text(rnd, 300, ", ", 7)
    assert len(s) == 300 and s.count(", ") == 7
    sk.stats(reset=True)
    r = sk.split(ck.encrypt(s, 1, None, sk), ck.encrypt_no_padding(", ", sk))
    got = _trim([ck.decrypt(b) for b in r.buffers])
    assert got == _trim(s.split(", "))
    assert ck.decrypt_char(r.pattern_found) == 1
    st = sk.stats()
    assert st["max_input_sum_c2"] <= 64 and st["pbs_executed"] < 4_000_000
    # buffer 6 and 7 sit beyond 255 characters into the string: a u8 position/buffer arithmetic could not address them
    assert len(got) == 8 and sum(len(x) for x in got[:6]) > 200


def test_splitn_270_chars_vs_python(product):
    """the counter stops at n - 1 (split.rs:136-173): min(count, n - 1) on 5-digit counters"""
    ck, sk = product
    rnd = random.Random(270)
    s = _text(rnd, 270, "-+-", 6)
    r = sk.splitn(ck.encrypt(s, 1, None, sk), ck.encrypt_no_padding("-+-", sk), ck.encrypt_char(3, sk))
    got = _trim([ck.decrypt(b) for b in r.buffers])
    assert got == _trim(s.split("-+-", 2))
