"""Host logic of the drop-in API: the behaviours the reference's validation suite greps for
(test/validation_suite/run_tests_avxout.sh:109-178, create_wrong_files.sh:23-82) -- which model
folders / parameters must be rejected and with which message.  Runs without a GPU: every check
below fires before the HIP context is created."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

from common import ROOT, folder

BAD = -2147479550      # RNLErrorBadParameter 0x80001002
UNDEF = -2147479551    # RNLErrorUndefined    0x80001001


def _init(model, ratio=2.0, bits=8, rng=1, threads=20, asm=2, passes=1, mode=1):
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {os.path.join(ROOT, 'video-super-resolution-library_amd')!r})
        import raisr_hip as R
        rc = R.RNLHandler_Init({model!r}, {ratio}, {bits}, {rng}, {threads}, {asm}, {passes}, {mode})
        R.RNLHandler_Deinit()
        print("RC", rc)
    """)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    rc = int([l for l in p.stdout.splitlines() if l.startswith("RC ")][-1].split()[1])
    return rc, p.stdout


@pytest.fixture()
def broken(tmp_path):
    def make(mutate):
        dst = tmp_path / "model"
        if dst.exists():
            shutil.rmtree(dst)
        shutil.copytree(folder("filters_2x/filters_highres"), dst)
        mutate(dst)
        return str(dst)
    return make


def _one_diag(out):
    lines = [l for l in out.splitlines() if any(k in l for k in ("failed", "not found", "RAISR WARNING", "RAISR ERROR"))]
    return lines


@pytest.mark.parametrize("cfg", ["12 3 3 11", "24 3 3", "24 3 3 6", "24 3 3 9", "x 3 3 11", "24 3 3 11 7"])
def test_corrupted_config_is_rejected(broken, cfg):
    rc, out = _init(broken(lambda d: (d / "config").write_text(cfg)))
    assert rc == BAD
    diag = _one_diag(out)
    assert len(diag) >= 1 and ("configFile corrupted" in diag[0] or "HashTable format" in diag[0])


@pytest.mark.parametrize("victim,msg", [
    ("config", "Unable to open config file"), ("filterbin_2_8", "Unable to load model"),
    ("Qfactor_strbin_2_8", "Unable to load model"), ("Qfactor_cohbin_2_8", "Unable to load model")])
def test_missing_or_renamed_files(broken, victim, msg):
    rc, out = _init(broken(lambda d: os.rename(d / victim, d / (victim + ".renamed"))))
    assert rc == BAD and msg in out and len(_one_diag(out)) == 1


def test_second_pass_files_needed_only_for_two_pass(broken):
    m = broken(lambda d: os.remove(d / "filterbin_2_8_2"))
    rc, out = _init(m, passes=2)
    assert rc == BAD and "Unable to load model" in out and "filterbin_2_8_2" in out


def test_truncated_and_mistagged_hashtable(broken):
    def trunc(d):
        b = (d / "filterbin_2_8").read_bytes()
        (d / "filterbin_2_8").write_bytes(b[:-4])
    rc, out = _init(broken(trunc))
    assert rc == BAD and "hashtable corrupted" in out
    def tag(d):
        b = (d / "filterbin_2_8").read_bytes()
        (d / "filterbin_2_8").write_bytes(b"fp64" + b[4:])
    rc, out = _init(broken(tag))
    assert rc == BAD and "hashtable corrupted" in out


@pytest.mark.parametrize("content,ok", [("0.001 0.02 0.3", False), ("0.001", False), ("abc 0.2", False), ("0..1 0.2", False),
                                        (".5 0.2", False), ("1e-3 0.2", False)])
def test_qfactor_validation(broken, content, ok):
    rc, out = _init(broken(lambda d: (d / "Qfactor_strbin_2_8").write_text(content)))
    assert rc == BAD and "StrFile corrupted" in out


def test_pixel_type_mismatch_between_ratio_and_model():
    rc, out = _init(folder("filters_2x/filters_highres"), ratio=1.5)
    assert rc == BAD and "number of pixel types" in out
    rc, out = _init(folder("filters_1.5x/filters_highres"), ratio=2.0)
    assert rc == BAD and "number of pixel types" in out


def test_parameter_validation():
    good = folder("filters_2x/filters_highres")
    rc, out = _init(good, bits=9)
    assert rc == BAD and "bit depth: 9bits is NOT supported" in out
    rc, out = _init(good, passes=3)
    assert rc == UNDEF and "Only support passes 1 or 2" in out
    rc, out = _init(good, asm=3)
    assert rc == BAD and "OpenCL requested, but OpenCL is not enabled" in out
    rc, out = _init(good, bits=16)     # no _16 files ship
    assert rc == BAD and "Unable to load model" in out
    rc, out = _init(os.path.join(ROOT, "does-not-exist"))
    assert rc == BAD and "Unable to open config file" in out


def test_one_pass_mode2_warns_and_valid_model_reaches_the_device_layer():
    rc, out = _init(folder("filters_2x/filters_highres"), passes=1, mode=2)
    assert "[RAISR WARNING] 1 pass with upscale in 2d pass, mode = 2 ignored !" in out
    import torch
    if not torch.cuda.is_available():
        # every file check passed; the only thing missing on a CPU box is the HIP device -> loud failure
        assert rc == UNDEF and "HIP backend unavailable" in out
    else:
        assert rc == 0


def test_process_null_planes_rejected():
    code = textwrap.dedent(f"""
        import sys, ctypes
        sys.path.insert(0, {os.path.join(ROOT, 'video-super-resolution-library_amd')!r})
        import raisr_hip as R
        v = R.VideoDataType()
        print("RC", R.lib().RNLHandler_Process(None, None, None, None, None, None, 2))
        print("RC", R.lib().RNLHandler_Process(ctypes.byref(v), ctypes.byref(v), ctypes.byref(v), ctypes.byref(v), ctypes.byref(v), ctypes.byref(v), 2))
    """)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    rcs = [int(l.split()[1]) for l in p.stdout.splitlines() if l.startswith("RC ")]
    assert rcs == [BAD, BAD]
