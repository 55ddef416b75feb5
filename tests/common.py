"""Shared helpers for the test-suite (paths, case matrix, frame factories)."""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def folder(name):
    return os.path.join(ROOT, name)


def dtype_for(bits):
    return np.uint8 if bits == 8 else np.uint16


# (id, folder, ratio(num,den), bits, passes, mode, asm, full_range)
CASES = [
    ("2x_highres_8b_1p_avx512", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, False),
    ("2x_lowres_8b_1p_avx2", "filters_2x/filters_lowres", (2, 1), 8, 1, 1, 1, False),
    ("2x_highres_8b_2p_m1", "filters_2x/filters_highres", (2, 1), 8, 2, 1, 2, False),
    ("2x_denoise_8b_2p_m2", "filters_2x/filters_denoise", (2, 1), 8, 2, 2, 2, False),
    ("2x_highres_10b_1p", "filters_2x/filters_highres", (2, 1), 10, 1, 1, 2, False),
    ("2x_highres_8b_full", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, True),
    ("1.5x_highres_8b_1p", "filters_1.5x/filters_highres", (3, 2), 8, 1, 1, 2, False),
    ("1.5x_denoise_8b_2p_m2", "filters_1.5x/filters_denoise", (3, 2), 8, 2, 2, 2, False),
    ("2x_denoise_10b_2p_m2", "filters_2x/filters_denoise", (2, 1), 10, 2, 2, 2, False),
    ("2x_highres_10b_2p_m1_full", "filters_2x/filters_highres", (2, 1), 10, 2, 1, 2, True),
    ("2x_lowres_8b_2p_m1_avx2", "filters_2x/filters_lowres", (2, 1), 8, 2, 1, 1, False),
    # asm 5 = AVX512-FP16 pipeline (binary16 arithmetic); BASELINE config 4 is the last one
    ("2x_highres_8b_1p_fp16", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 5, False),
    ("2x_highres_8b_2p_m1_fp16", "filters_2x/filters_highres", (2, 1), 8, 2, 1, 5, False),
    ("1.5x_denoise_8b_2p_m2_fp16", "filters_1.5x/filters_denoise", (3, 2), 8, 2, 2, 5, False),
]


def oracle_y(y, case, tie=0):
    """CPU oracle output for one CASE (fp32 paths -> raisr_oracle.c, asm 5 -> raisr_oracle_fp16.c)."""
    import oracle_py as O
    _, fold, (rn, rd), bits, passes, mode, asm, full = case
    h, w = y.shape
    ow, oh = w * rn // rd, h * rn // rd
    if asm == 5:
        p1 = O.make_pass16(folder(fold), bits, 1, full)
        p2 = O.make_pass16(folder(fold), bits, 2, full) if passes == 2 else None
        return O.process_y16(y, ow, oh, p1, p2, passes, mode, tie).astype(dtype_for(bits))
    p1 = O.make_pass(O.Model(folder(fold), bits, 1), bits, full, asm)
    p2 = O.make_pass(O.Model(folder(fold), bits, 2), bits, full, asm) if passes == 2 else None
    return O.process_y(y, ow, oh, p1, p2, passes, mode, tie).astype(dtype_for(bits))
