"""-m gpu: the HIP path (through the C ABI) against the CPU oracle, bit-exact, on seeded frames."""
import numpy as np
import pytest

from common import CASES, folder, dtype_for, oracle_y

pytestmark = pytest.mark.gpu


def _frames(w, h, bits):
    import synth
    return {
        "natural": synth.natural_y(w, h, bits, seed=12345),
        "random": synth.random_y(w, h, bits, seed=777),
        "checker": synth.checker_y(w, h, bits),
        "constant": synth.constant_y(w, h, bits),
    }


def _oracle(y, case):
    from common import oracle_y
    return oracle_y(y, case)


def _gpu(y, case):
    import raisr_hip as R
    _, fold, (rn, rd), bits, passes, mode, asm, full = case
    h, w = y.shape
    ow, oh = w * rn // rd, h * rn // rd
    dev = R.RaisrDevice(0)
    dev.set_model_from_folder(folder(fold), bits, passes)
    dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm)
    out = np.zeros((oh, ow), dtype_for(bits))
    dev.process_host(np.ascontiguousarray(y), out)
    dev.close()
    return out, None                                        # (the PRODUCT library: per-stage dumps need the test-hooks flavour, tests/test_gpu_golden.py)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("size", [(96, 64), (134, 50), (200, 40)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_y_bit_exact(case, size):
    w, h = size
    bits = case[3]
    for name, y in _frames(w, h, bits).items():
        ref = _oracle(y, case)
        got, _ = _gpu(y, case)
        bad = np.argwhere(ref != got)
        assert bad.size == 0, f"{case[0]} {name} {w}x{h}: {len(bad)} mismatching pixels, first at {bad[:5].tolist()}"


@pytest.mark.parametrize("asm,passes", [(2, 1), (1, 1), (5, 1), (2, 2)])
def test_randomness_blending_bit_exact(asm, passes):
    """BlendingMode Randomness: blended filtered zone, unclamped LR elsewhere, and the
    never-written pixels [c_final, W-6) of row H-7 keep the caller's bytes."""
    import oracle_py as O
    import raisr_hip as R
    import synth
    fold = folder("filters_2x/filters_highres")
    for w, h in ((96, 64), (134, 50)):
        for name in ("natural", "random"):
            y = synth.FRAME_KINDS[name](w, h, 8)
            ow, oh = 2 * w, 2 * h
            lr = O.resize(y, ow, oh)
            preset = np.full((oh, ow), 77, np.uint8)
            if asm == 5:
                p1 = O.make_pass16(fold, 8, 1, blending=O.BLEND_RANDOMNESS)
                p2 = O.make_pass16(fold, 8, 2, blending=O.BLEND_RANDOMNESS)
                run = lambda a, p, pre: O.run_pass16(a, p, preset=pre)
            else:
                p1 = O.make_pass(O.Model(fold, 8, 1), 8, False, asm, O.BLEND_RANDOMNESS)
                p2 = O.make_pass(O.Model(fold, 8, 2), 8, False, asm, O.BLEND_RANDOMNESS)
                run = lambda a, p, pre: O.run_pass(a, p, preset=pre)
            if passes == 1:
                ref = run(lr, p1, preset)
            else:
                ref = run(run(lr, p1, np.zeros((oh, ow), np.uint16)), p2, preset)
            dev = R.RaisrDevice(0)
            dev.set_model_from_folder(fold, 8, passes)
            dev.configure(w, h, ow, oh, bits=8, passes=passes, mode=1, hash_variant=asm, blending=R.BLEND_RANDOMNESS)
            out = preset.copy()
            dev.process_host(y, out)
            dev.set_blending(R.BLEND_COUNT)                      # per-frame switch, as RNLProcess allows
            out2 = np.zeros((oh, ow), np.uint8)
            dev.process_host(y, out2)
            dev.close()
            assert np.array_equal(out, ref.astype(np.uint8)), (asm, passes, w, h, name, int((out != ref).sum()))
            c_final = 6 + 8 * ((ow - 12) // 8)
            assert np.all(out[oh - 7, c_final:ow - 6] == 77)
            case = ("x", "filters_2x/filters_highres", (2, 1), 8, passes, 1, asm, False)
            assert np.array_equal(out2, _oracle(y, case))


@pytest.mark.parametrize("size,ratio,fold", [((31, 29), (2, 1), "filters_2x/filters_highres"), ((7, 7), (2, 1), "filters_2x/filters_highres"),
                                             ((13, 40), (2, 1), "filters_2x/filters_lowres"), ((86, 50), (3, 2), "filters_1.5x/filters_highres"),
                                             ((20, 12), (3, 2), "filters_1.5x/filters_denoise"), ((300, 8), (2, 1), "filters_2x/filters_highres")],
                         ids=lambda v: str(v).replace(" ", ""))
def test_small_and_odd_geometries(size, ratio, fold):
    """Ragged sizes: frames narrower than one 16-column chunk (nothing is filtered in AVX-512 mode,
    Raisr.cpp:1066), heights barely above the 12-row margin, odd output sizes."""
    import synth
    w, h = size
    for asm in (1, 2, 5):
        case = ("x", fold, ratio, 8, 1, 1, asm, False)
        y = synth.random_y(w, h, 8, seed=w * 100 + h)
        ref = _oracle(y, case)
        got, _ = _gpu(y, case)
        assert np.array_equal(ref, got), (size, asm, int((ref != got).sum()))


def test_device_planes_with_row_pitch():
    """raisr_hip_process_y_device with pitches larger than the row (device-resident frames inside bigger
    surfaces): only the addressed pixels are read/written."""
    import raisr_hip as R
    import synth
    import torch
    for bits in (8, 10):
        w, h = 100, 60
        dt = torch.uint8 if bits == 8 else torch.uint16
        y = synth.natural_y(w, h, bits, seed=21)
        src = torch.zeros((h, w + 28), dtype=dt, device="cuda")
        src[:, :w] = torch.from_numpy(y).cuda()
        dst = torch.full((2 * h, 2 * w + 56), 7, dtype=dt, device="cuda")
        dev = R.RaisrDevice(0)
        dev.set_model_from_folder(folder("filters_2x/filters_highres"), bits, 1)
        dev.configure(w, h, 2 * w, 2 * h, bits=bits)
        bps = 1 if bits == 8 else 2
        dev.process_y(src.data_ptr(), (w + 28) * bps, dst.data_ptr(), (2 * w + 56) * bps, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        dev.close()
        out = dst.cpu().numpy()
        ref = _oracle(y, ("x", "filters_2x/filters_highres", (2, 1), bits, 1, 1, 2, False))
        assert np.array_equal(out[:, :2 * w], ref) and np.all(out[:, 2 * w:] == 7)


@pytest.mark.parametrize("passes,mode,full", [(1, 1, False), (2, 1, True), (2, 2, False)])
def test_fp16_pipeline_on_10bit_content(passes, mode, full):
    """asm = avx512fp16 with 10-bit content: the reference runs its binary16 path there as well (Convert_8u16f_10bit, NF_10;
    Raisr.cpp:978-982, Raisr_AVX512FP16.cpp:146-151) -- accumulators overflow binary16 on strong gradients, deterministically.
    HIP binary16 pipeline vs the software-binary16 oracle, bit for bit."""
    import raisr_hip as R
    import synth
    w, h = 152, 94
    fold = "filters_2x/filters_denoise" if mode == 2 else "filters_2x/filters_highres"
    case = ("x", fold, (2, 1), 10, passes, mode, 5, full)
    for nm, y in (("natural", synth.natural_y(w, h, 10, seed=31)), ("random", synth.random_y(w, h, 10, seed=32)),
                  ("checker", synth.checker_y(w, h, 10))):
        ref = oracle_y(y, case)
        dev = R.RaisrDevice(0)
        try:
            dev.set_model_from_folder(folder(fold), 10, passes)
            dev.configure(w, h, 2 * w, 2 * h, bits=10, full_range=full, passes=passes, mode=mode, hash_variant=R.HASH_FP16)
            out = np.zeros((2 * h, 2 * w), np.uint16)
            dev.process_host(y, out)
        finally:
            dev.close()
        bad = np.argwhere(out != ref)
        assert bad.size == 0, (nm, len(bad), bad[:5].tolist())


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[2], CASES[4], CASES[6], CASES[7]], ids=lambda c: c[0])
def test_letterboxed_and_flat_regions_bit_exact(case):
    """Flat tiles (csrc/kernels_filter.h: not one non-zero gradient in the tile's 26 x 74 gradient tile) skip the hash stage and take the
    zero tensor's bucket.  Letterbox bars, a flat rectangle inside the picture, a bar one sample off its neighbours (gradients only at
    the seam), flat regions ending inside a tile and in the tail columns: every pixel must still equal the oracle's."""
    import synth
    bits = case[3]
    dt = dtype_for(bits)
    for (w, h) in ((416, 240), (134, 120)):
        y = synth.natural_y(w, h, bits, seed=321).astype(dt)
        top, bot = h // 5, h - h // 6
        y[:top] = 16 << (bits - 8)                       # letterbox bars at the legal-range black level
        y[bot:] = 16 << (bits - 8)
        y[top + 20:top + 60, w // 4:w // 4 + 90] = 200 << (bits - 8)          # a flat rectangle inside the picture (hard edges around it)
        y[bot - 30:bot - 10, :] = 90 << (bits - 8)                            # a full-width flat band ...
        y[bot - 20, :] += 1                                                   # ... with a one-level seam through it
        ref = _oracle(y, case)
        got, _ = _gpu(y, case)
        bad = np.argwhere(ref != got)
        assert bad.size == 0, f"{case[0]} {w}x{h}: {len(bad)} mismatching pixels, first at {bad[:5].tolist()}"
    # an all-black frame and a frame that is flat except for ONE sample
    for (w, h) in ((200, 96),):
        y = np.full((h, w), 16 << (bits - 8), dt)
        assert np.array_equal(_oracle(y, case), _gpu(y, case)[0])
        y[h // 2, w // 2] += 3
        assert np.array_equal(_oracle(y, case), _gpu(y, case)[0])
