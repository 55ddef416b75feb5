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


def _pair_read_extra_cycles(row_stride, b_offset):
    """LDS cycles lost to bank conflicts by the four pair reads of one filter step (ds_read_b32: two 32-lane groups, bank = dword
    index mod 32, identical addresses broadcast -- MI355X_MICROARCH.md, LDS): lane (g, l) of chunk ch reads dword
    array + (k0 / 11) * stride + k0 % 11 + g, k0 = 32 ch + l, array = A (0) for patch columns 0..5, B (b_offset) for 6..10."""
    extra = 0
    for ch in range(4):
        for half in range(2):
            by_bank = {}
            for g in (2 * half, 2 * half + 1):
                for l in range(16):
                    i, j = divmod(32 * ch + l, 11)
                    d = (0 if j + 5 < 11 else b_offset) + i * row_stride + j + g
                    by_bank.setdefault(d % 32, set()).add(d)
            extra += max(len(v) for v in by_bank.values()) - 1
    return extra


def test_pair_arrays_are_placed_conflict_free():
    """k_hashfilter16 places array B 15 dwords after array A (kPairPad): no lane pair of a 32-lane half hits one bank with two
    addresses.  Back to back (round 4) three of the four chunks were 2-way conflicted."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-super-resolution-library_amd", "csrc",
                            "kernels_fp16.h")).read()
    pad = int(re.search(r"constexpr int kPairPad = (\d+) \* 4;", src).group(1))
    assert _pair_read_extra_cycles(LW, 26 * LW + pad) == 0
    assert _pair_read_extra_cycles(LW, 26 * LW) == 6                # the round-4 layout: 6 extra cycles on 8
    assert (2 * 26 * LW + pad) * 4 <= (16 + 9) * 74 * 8 + 2048 * 2  # both arrays inside the hash stage's region
