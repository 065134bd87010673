"""Level-skewed batching (fhs_submit / fhs_pump): requests submitted one per tick share launch groups -- level l of
request k runs in the launch of request k + l - 1 -- and still decrypt like Python.  Includes the hazards the scheduler
has to handle: inputs released by the caller while a scheduled level still has to read them (their device blocks must not
be recycled by a later job that runs earlier), jobs that consume another job's result, and a plain flush in between."""
import random

import pytest

pytestmark = pytest.mark.gpu
SEED = 0xF5E57121


@pytest.fixture(scope="module", params=["f64_fft", "exact_ntt", "f64_fft_mb2", "exact_ntt_mb2"])
def product(request):
    from fhestring_amd.api import MyClientKey
    ck = MyClientKey(SEED)
    sk = ck.get_server_key(0, arith={"exact_ntt": 0, "f64_fft": 1, "f64_fft_mb2": 2, "exact_ntt_mb2": 3}[request.param])
    sk.set_mode(1)
    sk.test_arith = {"exact_ntt": 0, "f64_fft": 1, "f64_fft_mb2": 2, "exact_ntt_mb2": 3}[request.param]
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


def test_round_aligned_launch_groups_decrypt_correctly(product):
    """fhs_set_tick_balance: launch groups cut to whole rounds of the blind-rotation kernel (the excess of a first level
    runs one tick later with its consumers).  Small slot count here so that every step is split."""
    ck, sk = product
    sk.set_tick_balance(96)
    try:
        rnd = random.Random(13)
        texts = ["".join(chr(rnd.randint(0x20, 0x7E)) for _ in range(40)) for _ in range(6)]
        pats = [t[5:9] if i % 2 == 0 else "\x7f\x7f\x7f" for i, t in enumerate(texts)]
        sk.stats(reset=True)
        results, seen, groups = [], 0, []
        for t, p in zip(texts, pats):
            es = ck.encrypt(t, 1, None, sk)
            results.append((sk.contains_clear(es, p), sk.find_clear(es, p), sk.to_lower(es)))
            sk.submit()
            sk.pump(1)
            w = sk.level_widths()
            groups.append(sum(w[seen:]))
            seen = len(w)
        assert any(g % 96 == 0 for g in groups[1:]), groups
        sk.flush()
        for (c, f, u), t, p in zip(results, texts, pats):
            assert ck.decrypt_char(c) == int(p in t)
            assert ck.decrypt_char(f) == (t.find(p) if p in t else 255)
            assert ck.decrypt(u) == t.lower()
    finally:
        sk.set_tick_balance(0)


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


def test_automatic_partial_flush_decrypts_correctly(product):
    """fhs_set_auto_flush with a tiny threshold: the ready level is peeled hundreds of times while the DAGs are still
    being recorded (released pending nodes whose slots are reused, shared bootstraps, intermediate drops)."""
    ck, shared = product
    sk = ck.get_server_key(0, arith=shared.test_arith)    # a context driven with fhs_submit never flushes on its own: fresh one
    sk.set_mode(1)
    sk.set_auto_flush(48)
    try:
        rnd = random.Random(5)
        t = "".join(chr(rnd.randint(0x61, 0x7A)) for _ in range(96))
        t = t[:20] + "~from" + t[25:60] + "~from" + t[65:]
        es = ck.encrypt(t, 1, None, sk)
        sk.stats(reset=True)
        rep = sk.replace(es, ck.encrypt_no_padding("~from", sk), ck.encrypt_no_padding("[to]", sk))
        tmp = sk.to_upper(es)
        del tmp                                      # recorded, partly run, then dropped
        pos = sk.rfind(es, ck.encrypt_no_padding("~from", sk))
        parts = sk.split(ck.encrypt("a,b,,c", 1, None, sk), ck.encrypt_no_padding(",", sk))
        sk.flush()
        st = sk.stats()
        assert ck.decrypt(rep) == t.replace("~from", "[to]")
        assert ck.decrypt_char(pos) == t.rfind("~from")
        from fhestring_amd.api import FheSplit
        bufs, found = FheSplit.decrypt(parts, ck)
        assert found == 1 and bufs[:4] == ["a", "b", "", "c"] and all(b == "" for b in bufs[4:])
        assert st["max_input_sum_c2"] <= 64 and st["levels"] > 100
    finally:
        sk.close()
