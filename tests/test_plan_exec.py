"""The product's fused DAGs executed WITHOUT any HIP kernel: a planner context (host logic only) records the operation, the
plan trace (fhs_debug_plan_trace) is replayed on real ciphertexts by the CPU oracle's bootstrap (oracle/plan_exec.py), and
the result decrypts like Python.  Checks the string layer's re-association, the rotation-sharing groups (TR_EXT rows: a
shared extraction must equal its own bootstrap with the shifted constant) and the launch-group order independently of
the GPU arithmetic -- and is the mechanism bench.py's cpu_baseline uses to run BASELINE config 3 to completion on the
host cores (src/server_key/mod.rs:1010-1053 is the reference's find)."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def run(oracle_keys, oracle_sk):
    from oracle.plan_exec import PlanRun
    r = PlanRun(oracle_sk, threads=min(8, os.cpu_count() or 1), mode=6)
    yield r
    r.close()


def _enc(keys, text, pad=1):
    return np.stack([keys.encrypt_char(b) for b in text.encode() + b"\0" * pad])


def test_fused_dags_replayed_on_the_cpu_oracle_decrypt_like_python(run, oracle_keys):
    K, sk = oracle_keys, run.sk
    text = "a needle, NEEDLE"
    s = run.upload_string(_enc(K, text))
    pat = run.upload_string(_enc(K, "dle", pad=0))
    outs = {"contains_clear": sk.contains_clear(s, "need"), "contains_clear_miss": sk.contains_clear(s, "neex"),
            "find_enc": sk.find(s, pat.chars), "find_clear": sk.find_clear(s, "NEE")}
    up = sk.to_upper(s)
    st0 = sk.stats()
    run.run()
    st = sk.stats()
    dec = lambda ch: K.decrypt_char(run.result_char(ch))
    assert dec(outs["contains_clear"]) == 1 and dec(outs["contains_clear_miss"]) == 0
    assert dec(outs["find_enc"]) == text.find("dle") and dec(outs["find_clear"]) == text.find("NEE")
    assert bytes(dec(c) for c in up.chars).split(b"\0")[0].decode() == text.upper()
    # every bootstrap of the plan ran on the CPU: rotations + shared extractions (each replayed as its own bootstrap)
    assert run.pbs == (st["pbs_executed"] - st0["pbs_executed"]) + (st["pbs_extracted"] - st0["pbs_extracted"]) > 100
    assert st["pbs_extracted"] > 0 and st["max_input_sum_c2"] <= 64


def test_comparison_and_eq_ignore_case(run, oracle_keys):
    K, sk = oracle_keys, run.sk
    a, b = run.upload_string(_enc(K, "Plan Trace")), run.upload_string(_enc(K, "pLAN tRACF"))
    le, eqi, eq = sk.le(a, b), sk.eq_ignore_case(a, b), sk.eq(a, a)
    run.run()
    dec = lambda ch: K.decrypt_char(run.result_char(ch))
    assert dec(le) == int("Plan Trace" <= "pLAN tRACF") and dec(eqi) == 0 and dec(eq) == 1
