"""Noise margin of the fused DAGs on one MI355X: python tools/noise_report.py [fft|exact] > gpurun_out/noise.json

For BASELINE configs 2-5 (and find at the u8 index limit) every dependency level's PBS inputs are sampled
(fhs_debug_capture_pbs_inputs), decrypted with the client key and pushed through the product's own keyswitch +
modulus switch; per construct (LUT id, sum of squared coefficients) the table gives sigma and max of the error
entering blind rotation and the margin 64 / sigma.  tests/test_gpu_noise.py asserts on the same numbers."""
import json
import random
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import noise_util as nu                                    # noqa: E402
from fhestring_amd.api import MyClientKey                  # noqa: E402

SEED = 0xF5E57121
arith = 1 if (len(sys.argv) < 2 or sys.argv[1] == "fft") else 0
ck = MyClientKey(SEED)
sk = ck.get_server_key(0, arith=arith)
sk.set_mode(1)
rnd = random.Random(SEED)
R = lambda n: "".join(chr(rnd.randint(0x20, 0x7E)) for _ in range(n))     # noqa: E731

table = []
s_fresh, max_fresh, s_enc = nu.fresh_baseline(sk, ck, 8192)
s_pbs, max_pbs = nu.pbs_output_sigma(sk, ck, 4096)
table.append({"op": "fresh encryption -> KS+MS (floor)", "n": 8192, "sigma_tot": s_fresh, "max_tot": max_fresh,
              "z": 64 / s_fresh, "log2_sigma_in": __import__("math").log2(s_enc)})
table.append({"op": "one bootstrap output", "n": 4096, "log2_sigma_in": __import__("math").log2(s_pbs),
              "log2_max_in": __import__("math").log2(max_pbs)})


def run(tag, fn, rows=256):
    t, _ = nu.measure(sk, ck, tag, fn, rows)
    table.extend(t)
    print("# %s: %d constructs" % (tag, len(t)), file=sys.stderr, flush=True)


s = R(64); es = ck.encrypt(s, 1, None, sk)
run("cfg2 contains_clear 64 m=4", lambda: sk.contains_clear(es, s[20:24]), 4096)
s = list(R(256)); s[200:204] = "Qz7#"; s = "".join(s); es = ck.encrypt(s, 1, None, sk); ep = ck.encrypt_no_padding("Qz7#", sk)
run("cfg3 find 256 m=4 encrypted", lambda: sk.find(es, ep), 4096)
s = list(R(254)); s[250:253] = "Qz7"; s = "".join(s); es = ck.encrypt(s, 1, None, sk); ep = ck.encrypt_no_padding("Qz7", sk)
run("find 254 (+1 pad) m=3: largest position_of", lambda: sk.find(es, ep), 4096)
run("rfind 254 m=3", lambda: sk.rfind(es, ep), 4096)
s = list(R(1024).replace("~", "-"))
for k in range(8):
    s[20 + 120 * k:25 + 120 * k] = "~from"
s = "".join(s); es = ck.encrypt(s, 1, None, sk); ef = ck.encrypt_no_padding("~from", sk); et = ck.encrypt_no_padding("[to!]", sk)
run("cfg4 replace 1024 5->5", lambda: sk.replace(es, ef, et), 128)
a = R(4096); b = list(a.swapcase()); b[4000] = "a" if a[4000].lower() != "a" else "b"; b = "".join(b)
ea = ck.encrypt(a, 1, None, sk); eb = ck.encrypt(b, 1, None, sk)
run("cfg5 eq_ignore_case 4096", lambda: sk.eq_ignore_case(ea, eb), 512)
run("cfg5 le 4096", lambda: sk.le(ea, eb), 512)
run("len 4096", lambda: sk.len(ea), 512)
for row in table:
    if "z" in row:
        row["log2_pfail"] = nu.log2_pfail(row["z"])
    print(json.dumps(row))
