"""Level-skewed batching (fhs_submit / fhs_pump): requests submitted one per tick share launch groups -- level l of
request k runs in the launch of request k + l - 1 -- and still decrypt like Python.  Includes the hazards the scheduler
has to handle: inputs released by the caller while a scheduled level still has to read them (their device blocks must not
be recycled by a later job that runs earlier), jobs that consume another job's result, and a plain flush in between."""
import random

import pytest

pytestmark = pytest.mark.gpu
SEED = 0xF5E57121


@pytest.fixture(scope="module", params=["f64_fft", "exact_ntt"])
def product(request):
    from fhestring_amd.api import MyClientKey
    ck = MyClientKey(SEED)
    sk = ck.get_server_key(0, arith=1 if request.param == "f64_fft" else 0)
    sk.set_mode(1)
    yield ck, sk
    sk.close()
    ck.close()


def test_skewed_requests_decrypt_correctly(product):
    ck, sk = product
    rnd = random.Random(11)
    texts = ["".join(chr(rnd.randint(0x20, 0x7E)) for _ in range(40)) for _ in range(7)]
    pats = [t[10:13] if i % 2 == 0 else "\x7f\x7f" for i, t in enumerate(texts)]
    sk.stats(reset=True)
    results = []
    for t, p in zip(texts, pats):
        es = ck.encrypt(t, 1, None, sk)
        results.append((sk.contains_clear(es, p), sk.find_clear(es, p), sk.to_upper(es)))
        sk.submit()
        del es                  # the caller drops its inputs while scheduled levels still have to read them
        sk.pump(1)
    sk.flush()                  # drains the remaining ticks
    for (c, f, u), t, p in zip(results, texts, pats):
        assert ck.decrypt_char(c) == int(p in t)
        assert ck.decrypt_char(f) == (t.find(p) if p in t else 255)
        assert ck.decrypt(u) == t.upper()
    assert sk.stats()["max_input_sum_c2"] <= 64


def test_job_consuming_an_unfinished_job(product):
    ck, sk = product
    a, b = ck.encrypt("hello world", 1, None, sk), ck.encrypt("HELLO WORLD", 1, None, sk)
    low = sk.to_lower(b)
    sk.submit()                                  # scheduled, not enqueued
    same = sk.eq(a, low)                         # consumes every character of the unfinished job
    sk.submit()
    other = sk.contains_clear(a, "wor")          # independent: may run beside the first job
    sk.submit()
    sk.pump(2)
    mid = sk.len(a)                              # a plain flush in between drains what is scheduled
    assert ck.decrypt_char(mid) == 11
    assert ck.decrypt_char(same) == 1 and ck.decrypt_char(other) == 1 and ck.decrypt(low) == "hello world"
