"""CPU replay of the LDS addressing of the pair-column filter stage (csrc/kernels_filter.h: filter_phase<.., PC>; k_hashfilter_ac).

Group g of step s = 2 p + j filters tile column 8 p + 2 g + j, so a tap's window values for the two steps of a pair are adjacent and
arrive with ONE 8-byte LDS read.  gfx950 serves an 8-byte read at a 4-mod-8 address lane by lane (64 cycles instead of 2.8:
profiles/r05_lds_b64_probe.log), so the layout must make EVERY such read 8-byte aligned, and it should be free of bank conflicts
(ds_read_b64: two groups of 32 lanes, bank = dword address mod 64, identical addresses broadcast -- MI355X_MICROARCH.md).  The
constants are parsed from the kernel source, not retyped.
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "video-super-resolution-library_amd", "csrc", "kernels_filter.h")).read()
CERT = open(os.path.join(ROOT, "video-super-resolution-library_amd", "csrc", "kernels_hash_certify.h")).read()
COMMON = open(os.path.join(ROOT, "video-super-resolution-library_amd", "csrc", "kernels_common.h")).read()

K_TAPS = int(re.search(r"constexpr int kTaps = (\d+);", COMMON).group(1))
LW = int(re.search(r"constexpr int LW = PC \? (\d+) : \d+", SRC).group(1))
Q_IN_V = int(re.search(r"constexpr unsigned oQ = oV \+ (\d+);", SRC).group(1))
LIST_MAX = int(re.search(r"#else\s*constexpr unsigned kListMax = (\d+);", CERT).group(1))
TH, TW, GW, GH = 16, 64, 74, 26
LH = TH + 12


def layout():
    """byte offsets of the kernel's hand-carved LDS block (8-bit / fp32-gradient variant: GT = float2, VT = float4)"""
    oL = 0
    oG = oL + LH * LW * 4
    oV = oG + GH * GW * 8
    oH = oV + 3 * 4 * GW * 16
    oQ = oV + Q_IN_V
    return oL, oG, oV, oH, oQ


def tap_of(l, ch, sym):
    """tap index lane l handles in chain step ch (filter_phase's `k`); K_TAPS = padding"""
    k = 16 * ch + l
    if sym and ch >= 4:
        l2 = (8 - l) & 15
        k = 16 * ch + l2 if l <= 8 else (K_TAPS if ch == 4 else 16 * (ch - 1) + l2)
    return k


def tap_dword(l, ch, g, p, prow, sym):
    """LDS dword address of the 8-byte read of lane (g, l), chain step ch, pair p, tile row prow; None for the symmetric stage's
    padding step (it reads the wave's block of zeros: one aligned, broadcast address)"""
    oL, _, _, _, oQ = layout()
    sP = oL // 4 + LW + 1                       # window position (r0 - 5, c0 - 5)
    sQ = oQ // 4                                # the second copy: sQ[e] == sP[e]
    k = tap_of(l, ch, sym)
    if k >= K_TAPS:
        if sym:
            return None
        return sQ + prow * LW + 2 * g + 8 * p   # plain stage: "any finite pixel", taken from the aligned copy
    i, j = divmod(k, 11)
    base = sP if (j & 1) else sQ                # odd patch column: aligned in the window itself; even: in the copy
    return base + (prow + i) * LW + j + 2 * g + 8 * p


def test_layout_fits_and_copy_is_in_dead_space():
    oL, oG, oV, oH, oQ = layout()
    total = oH + 2 * TH * TW + LIST_MAX * 2 + 16
    assert total <= 40960, "four workgroups per CU need <= 40 960 B each"
    assert oQ % 8 == 0 and (oL // 4 + LW + 1) % 2 == 1, "copy aligned, window origin at an odd dword"
    assert oQ >= oV + 1024 + LIST_MAX * 16, "behind the exact path's table and tensors"
    assert oQ + 26 * LW * 4 <= oH, "inside sV"


@pytest.mark.parametrize("sym", [True, False])
def test_every_window_read_is_8_byte_aligned(sym):
    for prow in range(TH):
        for p in range(8):
            for g in range(4):
                for l in range(16):
                    for ch in range(8):
                        a = tap_dword(l, ch, g, p, prow, sym)
                        assert a is None or a % 2 == 0, (prow, p, g, l, ch)


@pytest.mark.parametrize("sym", [True, False])
def test_window_reads_are_free_of_bank_conflicts(sym):
    for prow in range(TH):
        for p in range(8):
            for ch in range(8):
                for grp in range(2):                    # lanes 0-31 (g = 0, 1) and 32-63 (g = 2, 3)
                    banks = {}
                    for g in (2 * grp, 2 * grp + 1):
                        for l in range(16):
                            a = tap_dword(l, ch, g, p, prow, sym)
                            if a is None:
                                continue
                            for d in (0, 1):
                                banks.setdefault((a + d) % 64, set()).add(a + d)
                    assert max(len(v) for v in banks.values()) == 1, (prow, p, ch, grp)


@pytest.mark.parametrize("sym", [True, False])
def test_reads_deliver_the_right_window_values(sym):
    """the two dwords of a read are tap k of the pixels in tile columns 8 p + 2 g and 8 p + 2 g + 1 (window coordinates)"""
    oL, _, _, _, oQ = layout()
    sP, sQ = oL // 4 + LW + 1, oQ // 4
    lds = {}
    for e in range(26 * LW):                            # what the kernel stores: the window, and copy_window's sQ[e] = sP[e]
        lds[sP + e] = ("win", e)
        lds[sQ + e] = ("win", e)
    for prow in (0, 7, 15):
        for p in range(8):
            for g in range(4):
                for l in range(16):
                    for ch in range(8):
                        k = tap_of(l, ch, sym)
                        if k >= K_TAPS:
                            continue
                        a = tap_dword(l, ch, g, p, prow, sym)
                        i, j = divmod(k, 11)
                        for step_parity in (0, 1):
                            col = 8 * p + 2 * g + step_parity
                            assert lds[a + step_parity] == ("win", (prow + i) * LW + col + j)


def test_every_tile_column_is_filtered_once():
    cols = sorted(8 * (s >> 1) + 2 * g + (s & 1) for s in range(16) for g in range(4))
    assert cols == list(range(TW))
