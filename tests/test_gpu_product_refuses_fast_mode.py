"""-m gpu: the product library has no fast mode (the NON-bit-exact matrix-core filter stage of rounds 2-3): asking for it fails
loudly instead of silently running something else; a development build still passes the mode's quality bounds."""
import os
import subprocess
import sys

import pytest

from common import folder

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV_LIB = os.path.join(ROOT, "video-super-resolution-library_amd", "_exp", "libraisr_dev.so")


def test_set_fast_is_refused_by_the_product_library():
    import raisr_hip as R
    dev = R.RaisrDevice(0)
    try:
        dev.set_model_from_folder(folder("filters_2x/filters_highres"), 8, 1)
        dev.configure(96, 64, 192, 128, bits=8, passes=1, hash_variant=R.HASH_AVX512)
        dev.set_fast(0)                                       # "off" is always accepted
        with pytest.raises(RuntimeError, match="not part of the product build"):
            dev.set_fast(1)
        assert dev.fast() == 0
    finally:
        dev.close()


def test_environment_request_for_fast_mode_fails_at_create():
    code = "import sys; sys.path.insert(0, %r); import raisr_hip as R\ntry:\n    R.RaisrDevice(0)\nexcept RuntimeError as e:\n    print('REFUSED', e)\n" % os.path.join(ROOT, "video-super-resolution-library_amd")
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RAISR_HIP_FAST="2"), capture_output=True, text=True, timeout=600)
    assert "REFUSED" in out.stdout and "not part of the product build" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(not os.path.exists(DEV_LIB), reason="no development build (scripts/build_exp.sh dev -DRAISR_HIP_DEV)")
def test_development_build_keeps_the_mode_within_its_bounds():
    env = dict(os.environ, RAISR_HIP_LIB=DEV_LIB, RAISR_HIP_DEV_BUILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_fast_mode.py"), "-x", "-q", "-m", "gpu"], env=env,
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("var", ["RAISR_HIP_SPLIT", "RAISR_HIP_DEFER"])
def test_comparison_pipelines_are_refused_by_the_product_library(var, monkeypatch):
    """RAISR_HIP_SPLIT / RAISR_HIP_DEFER select pipelines that exist in the test-hooks flavour only (include/raisr_hip_debug.h): the
    product library says so instead of silently running its own pipeline; the hooks flavour honours them (tests/test_gpu_pipelines.py)."""
    import raisr_hip as R
    monkeypatch.setenv(var, "1")
    with pytest.raises(RuntimeError, match="RAISR_HIP_TESTHOOKS"):
        R.RaisrDevice(0)
    R.RaisrDevice(0, hooks=True).close()
    monkeypatch.setenv(var, "0")
    R.RaisrDevice(0).close()                                  # "off" is always accepted


def test_debug_hooks_are_not_in_the_product_library():
    import raisr_hip as R
    dev = R.RaisrDevice(0)
    try:
        with pytest.raises(RuntimeError, match="hooks=True"):
            dev.keep_stages(True)
        with pytest.raises(RuntimeError, match="hooks=True"):
            dev.certify_debug(True, False)
    finally:
        dev.close()
