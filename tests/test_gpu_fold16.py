"""GPU: the binary16 hash's folded thresholds equal the divisions they replace -- exhaustively.

GetHashValue_AVX512FP16 (Raisr_AVX512FP16.cpp:382-471,497-590) divides twice only to compare the quotient with constants:
strength = L1 / 100 against Qfactor_strbin, coherence = (sqrt L1 - sqrt L2) / (sqrt L1 + sqrt L2) against Qfactor_cohbin.  The
kernels compare the dividends instead (csrc/kernels_fp16.h, Pass16).  The device sweeps every binary16 L1 and every operand pair
(n, d) the fast hash can produce -- d > 0 finite or NaN, n finite or NaN: 2^31 pairs -- through both forms, for the thresholds of
every shipped model."""
import glob
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FOLDERS = sorted(d for d in glob.glob(os.path.join(ROOT, "filters_*", "filters_*")) if os.path.isdir(d))


@pytest.mark.gpu
@pytest.mark.parametrize("folder", FOLDERS, ids=[os.path.relpath(f, ROOT) for f in FOLDERS])
@pytest.mark.parametrize("bits", [8, 10])
def test_folded_thresholds_equal_the_divisions(folder, bits):
    import raisr_hip as R
    passes = 2 if glob.glob(os.path.join(folder, f"filterbin_*_{bits}_2")) else 1
    if not glob.glob(os.path.join(folder, f"filterbin_*_{bits}")):
        pytest.skip("no model for this bit depth")
    dev = R.RaisrDevice(0, hooks=True)
    try:
        dev.set_model_from_folder(folder, bits, passes)
        for p in range(passes):
            bad, pairs = dev.debug_fold16_check(p)
            assert pairs > 2_000_000_000, pairs
            assert bad == 0, (folder, bits, p, bad)
    finally:
        dev.close()
