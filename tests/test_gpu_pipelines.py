"""-m gpu: EVERY pipeline that can be selected through the environment produces the oracle's bits -- the product library's (default,
RAISR_HIP_FUSED=0, RAISR_HIP_CERTIFY=0, RAISR_HIP_CHUNKS) on the product library, the comparison pipelines (RAISR_HIP_SPLIT,
RAISR_HIP_DEFER) and every self-check run on the test-hooks flavour (libraisr_hip_testhooks.so), where they live.

The library reads RAISR_HIP_SPLIT / RAISR_HIP_LDS_FILTER / RAISR_HIP_FUSED / RAISR_HIP_CERTIFY when a context is created
(csrc/device_abi.hip: create_impl).  The default is the fused certified kernel k_hashfilter_ac; the alternatives chain other
kernels (k_hash_ac + k_fix_sparse/k_fix_dense + k_filter_lds16 | k_filter; k_hash + k_filter; the all-exact k_hashfilter;
k_hash16 + k_filter16 for the binary16 numerics).  What ships is tested: each of them against the CPU oracle, bit for bit, for
8/10/16-bit content, both hash flavours, one and two passes, 2x and 1.5x; the certified ones also in self-check mode (every
pixel through the exact path as well, certified-but-different buckets counted: must be 0)."""
import numpy as np
import pytest

from common import folder, oracle_y, dtype_for

pytestmark = pytest.mark.gpu

PIPELINES = {
    "default": {},
    "split_lds": {"RAISR_HIP_SPLIT": "1"},
    "split_l1": {"RAISR_HIP_SPLIT": "1", "RAISR_HIP_LDS_FILTER": "0"},
    "unfused": {"RAISR_HIP_FUSED": "0"},
    "exact": {"RAISR_HIP_CERTIFY": "0"},
    "exact_unfused": {"RAISR_HIP_CERTIFY": "0", "RAISR_HIP_FUSED": "0"},
    # host-plane entry: last pass in row ranges, finished rows downloaded while the next range is computed
    "chunks3": {"RAISR_HIP_CHUNKS": "3"},
    "chunks8": {"RAISR_HIP_CHUNKS": "8"},
    # the exact path of the uncertified pixels as its own kernel (k_hashfilter_ac<DEFER> + k_fix_ac): round-5 experiment, kept for comparisons
    "defer": {"RAISR_HIP_DEFER": "1"},
    "defer_chunks3": {"RAISR_HIP_DEFER": "1", "RAISR_HIP_CHUNKS": "3"},
}
CERTIFIED = ("default", "split_lds", "split_l1", "chunks3")
HOOKS_ONLY = ("split_lds", "split_l1", "defer", "defer_chunks3")

# (id, folder, ratio, bits, passes, mode, asm, full)
CASES = [
    ("2x_8b_avx512", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, False),
    ("2x_8b_avx2_2p", "filters_2x/filters_lowres", (2, 1), 8, 2, 1, 1, False),
    ("2x_10b_2p_m2", "filters_2x/filters_denoise", (2, 1), 10, 2, 2, 2, True),
    ("1.5x_8b_2p_m2", "filters_1.5x/filters_denoise", (3, 2), 8, 2, 2, 2, False),
    ("1.5x_8b_avx2", "filters_1.5x/filters_highres", (3, 2), 8, 1, 1, 1, False),
    ("2x_8b_fp16_2p", "filters_2x/filters_highres", (2, 1), 8, 2, 1, 5, False),
]


def _frames(w, h, bits):
    import synth
    return {"natural": synth.natural_y(w, h, bits, seed=4242), "random": synth.random_y(w, h, bits, seed=4243),
            "checker": synth.checker_y(w, h, bits), "constant": synth.constant_y(w, h, bits)}


def _run(R, y, case, check=None, blending=None, hooks=False):
    _, fold, (rn, rd), bits, passes, mode, asm, full = case
    h, w = y.shape
    ow, oh = w * rn // rd, h * rn // rd
    dev = R.RaisrDevice(0, hooks=hooks or check is not None)              # the environment is read here
    try:
        dev.set_model_from_folder(folder(fold), bits, passes)
        kw = {} if blending is None else {"blending": blending}
        dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm, **kw)
        if check is not None:
            dev.certify_debug(True, check)
        out = np.zeros((oh, ow), dtype_for(bits))
        dev.process_host(np.ascontiguousarray(y), out)
        st = dev.certify_stats() if check is not None else None
    finally:
        dev.close()
    return out, st


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("pipeline", sorted(PIPELINES))
def test_every_selectable_pipeline_is_bit_exact(pipeline, case, monkeypatch):
    import raisr_hip as R
    for k in ("RAISR_HIP_SPLIT", "RAISR_HIP_LDS_FILTER", "RAISR_HIP_FUSED", "RAISR_HIP_CERTIFY", "RAISR_HIP_FAST", "RAISR_HIP_CHUNKS", "RAISR_HIP_DEFER"):
        monkeypatch.delenv(k, raising=False)
    for k, v in PIPELINES[pipeline].items():
        monkeypatch.setenv(k, v)
    bits, asm = case[3], case[6]
    # 290 x 150 -> 580 x 300 at 2x: several 64 x 16 tiles and several 128 x 32 regions of k_filter_lds16 in both directions, none of them whole
    for (w, h) in ((290, 150), (70, 41)):
        for kind, y in _frames(w, h, bits).items():
            ref = oracle_y(y, case)
            out, _ = _run(R, y, case, hooks=pipeline in HOOKS_ONLY)
            bad = np.argwhere(out != ref)
            assert bad.size == 0, f"{pipeline} {case[0]} {kind} {w}x{h}: {len(bad)} mismatching pixels, first {bad[:4].tolist()}"
            if pipeline in CERTIFIED and asm != 5 and (w, h) == (290, 150):
                out2, st = _run(R, y, case, check=True, hooks=True)      # self-check: every pixel also through the exact path
                assert st["pixels"] > 0 and st["mismatches"] == 0, (pipeline, case[0], kind, st)
                assert np.array_equal(out2, ref), (pipeline, case[0], kind, "self-check output")


@pytest.mark.parametrize("pipeline", ["split_lds", "split_l1", "unfused", "exact"])
def test_pipelines_16bit_and_randomness(pipeline, monkeypatch, tmp_path):
    """16-bit content (k_filter_lds16 must step aside: samples above 10 bits are not exact in binary16) and the Randomness
    blending mode (tail columns: the AVX2 re-hash candidate replaces the first) through the alternative pipelines."""
    import shutil
    import raisr_hip as R
    for k, v in PIPELINES[pipeline].items():
        monkeypatch.setenv(k, v)
    import synth
    # Randomness, 8-bit
    case = ("rand", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, False)
    import oracle_py as O
    w, h = 150, 70
    y = synth.natural_y(w, h, 8, seed=99)
    p1 = O.make_pass(O.Model(folder(case[1]), 8, 1), 8, False, 2, O.BLEND_RANDOMNESS)
    preset = np.full((2 * h, 2 * w), 77, np.uint8)
    ref = O.run_pass(O.resize(y, 2 * w, 2 * h), p1, preset=preset).astype(np.uint8)
    dev = R.RaisrDevice(0, hooks=pipeline in HOOKS_ONLY)
    try:
        dev.set_model_from_folder(folder(case[1]), 8, 1)
        dev.configure(w, h, 2 * w, 2 * h, bits=8, passes=1, mode=1, hash_variant=2, blending=R.BLEND_RANDOMNESS)
        out = preset.copy()
        dev.process_host(y, out)
    finally:
        dev.close()
    assert np.array_equal(out, ref), (pipeline, int((out != ref).sum()))
    # 16-bit with a synthesised _16 model folder (the reference ships none: same coefficients, 16-bit thresholds file names)
    src = folder("filters_2x/filters_highres")
    dst = tmp_path / "filters16"
    shutil.copytree(src, dst)
    for stem in ("filterbin_2", "Qfactor_strbin_2", "Qfactor_cohbin_2"):
        shutil.copyfile(dst / f"{stem}_10", dst / f"{stem}_16")
    y16 = (synth.natural_y(w, h, 10, seed=5).astype(np.uint32) * 64).astype(np.uint16)
    p16 = O.make_pass(O.Model(str(dst), 16, 1), 16, False, 2)
    ref16 = O.process_y(y16, 2 * w, 2 * h, p16, None, 1, 1).astype(np.uint16)
    dev = R.RaisrDevice(0, hooks=pipeline in HOOKS_ONLY)
    try:
        dev.set_model_from_folder(str(dst), 16, 1)
        dev.configure(w, h, 2 * w, 2 * h, bits=16, passes=1, mode=1, hash_variant=2)
        out16 = np.zeros((2 * h, 2 * w), np.uint16)
        dev.process_host(y16, out16)
    finally:
        dev.close()
    assert np.array_equal(out16, ref16), (pipeline, int((out16 != ref16).sum()))


def test_configure_race_stress():
    """Start-up ordering (DESIGN s4: null-stream clears/copies are not ordered with the non-blocking streams the kernels run on):
    400 create / set-model / configure / first-frame cycles of a small two-pass job, every output compared with the oracle."""
    import raisr_hip as R
    import synth
    case = ("x", "filters_2x/filters_denoise", (2, 1), 8, 2, 2, 2, False)
    ys = [synth.constant_y(64, 40, 8), synth.natural_y(64, 40, 8, seed=3)]
    refs = [oracle_y(y, case) for y in ys]
    bad = 0
    for it in range(400):
        y, ref = ys[it & 1], refs[it & 1]
        out, _ = _run(R, y, case)
        bad += int(not np.array_equal(out, ref))
    assert bad == 0, f"{bad} of 400 first frames differ from the oracle"
