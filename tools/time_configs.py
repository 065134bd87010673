"""Wall time of BASELINE.json configs 2-5 (fused DAGs) on one MI355X: python tools/time_configs.py [fft|exact]."""
import random, sys, time
sys.path.insert(0, ".")
from fhestring_amd.api import MyClientKey
SEED = 0xF5E57121
arith = 0 if "exact" in sys.argv[1:] else 1
BALANCE = "--balance" in sys.argv       # round-aligned launch groups inside one op (fhs_set_tick_balance)
ck = MyClientKey(SEED)
sk = ck.get_server_key(0, arith=arith)
sk.set_mode(1)
if BALANCE:
    print("round alignment on, slots", sk.set_tick_balance())
rnd = random.Random(SEED)
R = lambda n: "".join(chr(rnd.randint(0x20, 0x7E)) for _ in range(n))


def timed(name, fn, check):
    out = fn(); sk.flush()                                # warm-up, then the better of two
    best = None
    for _ in range(2):
        sk.flush(); sk.stats(reset=True)
        t0 = time.perf_counter(); out = fn(); sk.flush(); dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, sk.stats(), sk.level_widths())
    dt, st, w = best
    ok = check(out)
    print("%-34s %9.1f ms  %8d PBS  %4d levels  %8.0f PBS/s  %s  %s" % (name, dt * 1e3, st["pbs_executed"], st["levels"],
          st["pbs_executed"] / dt, "OK" if ok else "WRONG", w[:14]))


s = R(64); es = ck.encrypt(s, 1, None, sk); pat = s[20:24]
timed("cfg2 contains_clear 64, m=4", lambda: sk.contains_clear(es, pat), lambda o: ck.decrypt_char(o) == 1)
s = list(R(256)); s[200:204] = "Qz7#"; s = "".join(s); es = ck.encrypt(s, 1, None, sk); ep = ck.encrypt_no_padding("Qz7#", sk)
timed("cfg3 find 256, m=4 (encrypted)", lambda: sk.find(es, ep), lambda o: ck.decrypt_char(o) == s.find("Qz7#"))
s = list(R(1024).replace("~", "-"))
for k in range(8):
    s[20 + 120 * k:25 + 120 * k] = "~from"
s = "".join(s); es = ck.encrypt(s, 1, None, sk); ef = ck.encrypt_no_padding("~from", sk); et = ck.encrypt_no_padding("[to!]", sk)
timed("cfg4 replace 1024, 5->5", lambda: sk.replace(es, ef, et), lambda o: ck.decrypt(o) == s.replace("~from", "[to!]"))
a = R(4096); b = list(a.swapcase()); b[4000] = "a" if a[4000].lower() != "a" else "b"; b = "".join(b)
ea = ck.encrypt(a, 1, None, sk); eb = ck.encrypt(b, 1, None, sk)
timed("cfg5 eq_ignore_case 4096", lambda: sk.eq_ignore_case(ea, eb), lambda o: ck.decrypt_char(o) == 0)
timed("cfg5 le 4096", lambda: sk.le(ea, eb), lambda o: ck.decrypt_char(o) == int(a <= b))
