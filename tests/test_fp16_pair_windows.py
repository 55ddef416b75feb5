"""CPU: index arithmetic of the binary16 filter stage's pair windows (csrc/kernels_fp16.h: build_pair_windows, filter16_phase<.., PAIRS>).

A lane's tap pair (k, k + 16) of DotProdPatch_AVX512FP16_16f (Raisr_AVX512FP16.cpp:227-242: zmm lanes l and l + 16 of a 32-tap chunk)
is read as ONE packed entry of array A (partner one row down, five columns right) or B (two rows down, six columns left).  Checked
here for every pixel of a tile, lane and chunk: the entry's two halves are the two taps' window positions, the entry is one the builder
writes, and the builder's partner reads stay inside the staged 28 x 77 window."""
TW, TH, LW = 64, 16, 77


def test_every_tap_pair_is_one_written_entry():
    for prow in range(TH):
        for pc in range(TW):
            for l in range(16):
                for ch in range(4):
                    k0 = 32 * ch + l
                    k1 = k0 + 16
                    assert k0 < 121
                    i0, j0 = divmod(k0, 11)
                    use_a = j0 + 5 < 11
                    y, x = prow + i0, pc + j0                       # entry (y, x) of A or B, window coordinates
                    assert 0 <= y < 26 and 0 <= x < 74
                    py, px = (y + 1, x + 5) if use_a else (y + 2, x - 6)
                    if k1 < 121:                                    # a real tap: its position must be the partner
                        i1, j1 = divmod(k1, 11)
                        assert (py, px) == (prow + i1, pc + j1)
                        assert 0 <= py < 26 and 0 <= px < 74
                    # written by build_pair_windows?
                    if use_a:
                        assert x <= 68
                    else:
                        assert x >= 6 and y < 25
                    # the builder reads the partner from the staged window sW = sL + LW + 1 (rows -1..26, flat index inside 28 * 77)
                    flat = (py + 1) * LW + (px + 1)
                    assert 0 <= flat < 28 * LW


def test_builder_covers_exactly_the_window():
    written_a, written_b = set(), set()
    for w in range(4):
        for i in range(7):
            y = w + 4 * i
            if y >= 26:
                continue
            for lane in range(64):
                for x in [lane] + ([64 + lane] if lane < 10 else []):
                    if x <= 68:
                        written_a.add((y, x))
                    if x >= 6 and y < 25:
                        written_b.add((y, x))
    assert written_a == {(y, x) for y in range(26) for x in range(69)}
    assert written_b == {(y, x) for y in range(25) for x in range(6, 74)}
    # the centre pixel of the accept test is read as the low half of A[prow + 5][pc + 5]
    assert all((prow + 5, pc + 5) in written_a for prow in range(TH) for pc in range(TW))
