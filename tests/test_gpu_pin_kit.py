"""-m gpu: dry run of tools/pin_against_reference/run_reference.py.  The kit drives a library through RNLHandler_Init / SetRes /
Process / Deinit in one process per case; here the library is THIS repository's (same five-function ABI), so the runner, its frame
generators and its comparison with the oracle are exercised end to end: every case must come out bit-exact under the half-up tie
rule.  (With the real reference and Intel IPP the same command writes tests/golden/reference_digests.json.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_runner_end_to_end_with_this_library(tmp_path):
    so = os.path.join(ROOT, "video-super-resolution-library_amd", "libraisr_hip.so")
    for build in ("strict", "shipped"):
        os.symlink(so, tmp_path / f"libraisr_ref_{build}.so")
    out = tmp_path / "digests.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pin_against_reference", "run_reference.py"), "--ref", ROOT,
                           "--libs", str(tmp_path), "--out", str(out), "--only", r"^(2x_highres_8b_1p_avx512|2x_lowres_8b_1p_avx2|1.5x_denoise_8b_2p_m2|2x_highres_10b_2p_m1_full)/|^C1/"],
                          timeout=900)
    d = json.load(open(out))
    assert d["cases_run"] == 9 and d["model_files_identical_to_this_repository"]
    assert d["tie_rule_that_reproduces_ipp"] == "half_up"
    for jid, rec in d["cases"].items():
        assert rec["oracle_vs_strict_px"]["half_up"] == 0 and rec["shipped_vs_strict_px"] == 0, jid
