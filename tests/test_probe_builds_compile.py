"""CPU: the timing-probe variants that live IN the tree behind macros still compile (device-only, gfx950) -- a probe that no longer
builds is not a record (`RAISR_PROBE_FILTER_STEPS` called a lambda with a stale signature for a whole round).
One hipcc run per variant (~20 s each); skipped where hipcc is absent."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
VARIANTS = {
    "filter steps with shared coefficient rows (kernels_probes.h)": ["-DRAISR_HIP_DEV", "-DRAISR_EXP_COEF_REUSE=4"],
    "filter stage without window reads (kernels_probes.h)": ["-DRAISR_HIP_DEV", "-DRAISR_EXP_NO_WINDOW"],
    "ceiling of a second certification level (kernels_hash_certify.h)": ["-DRAISR_PROBE_L2CERT"],
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_probe_variant_compiles(name):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc here")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-pass-failed",
               *VARIANTS[name], "--cuda-device-only", "-c", "-o", os.path.join(tmp, "probe.o"),
               os.path.join(ROOT, "video-super-resolution-library_amd", "csrc", "device_abi.hip")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
