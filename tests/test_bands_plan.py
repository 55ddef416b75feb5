"""Band plan (raisr_hip_plan_bands): pure host arithmetic behind the intra-frame split of RNLProcess and of the
multi-GPU latency mode.  Checked here: the kept rows tile the output exactly once, every band's sub-frame is an
aligned crop with the frame's own upscale ratio, and artificial borders are padded by at least the dependency
reach of the pipeline (DESIGN.md "Bands")."""
import pytest


def _need_pad_out_rows(passes, in_h, out_h):
    """Output rows between an artificial border and the first row that equals the whole-frame result."""
    import math
    if passes == 0:
        return 2                              # bilinear: one replicated input row
    if passes == 1:
        return 8                              # LR row 0 wrong, 6-row unfiltered margin, +-1 row census window
    mode1 = 16                                # pass 2 reads pass-1 rows r-7..r+7
    mode2 = math.ceil(9 * out_h / in_h) + 8   # pass 1 at input size valid from input row 8, upscaled, then one pass
    return max(mode1, mode2)


@pytest.mark.parametrize("in_h,out_h", [(1080, 2160), (720, 1080), (2160, 4320), (540, 1080), (120, 240), (120, 180),
                                        (270, 540), (1080, 1080), (96, 144), (45, 90)])
@pytest.mark.parametrize("passes", [0, 1, 2])
@pytest.mark.parametrize("want", [1, 2, 3, 4, 8])
def test_plan_tiles_the_output_and_pads_artificial_borders(in_h, out_h, passes, want):
    import raisr_hip as R
    bands = R.plan_bands(in_h, out_h, passes, want)
    assert 1 <= len(bands) <= want
    row = 0
    for i, b in enumerate(bands):
        assert b["keep_begin"] == row and b["keep_count"] > 0
        row += b["keep_count"]
        # crop inside the frame, same ratio, aligned start (even rows, whole upscale periods)
        assert 0 <= b["in_row_begin"] and b["in_row_begin"] + b["in_row_count"] <= in_h
        assert b["in_row_begin"] % 2 == 0
        assert b["out_row_begin"] * in_h == b["in_row_begin"] * out_h
        assert b["out_row_count"] * in_h == b["in_row_count"] * out_h
        # kept rows inside the sub-frame, far enough from artificial borders
        top_gap = b["keep_begin"] - b["out_row_begin"]
        bot_gap = (b["out_row_begin"] + b["out_row_count"]) - (b["keep_begin"] + b["keep_count"])
        assert top_gap >= 0 and bot_gap >= 0
        if b["in_row_begin"] > 0:
            assert top_gap >= _need_pad_out_rows(passes, in_h, out_h), (b, top_gap)
        else:
            assert b["out_row_begin"] == 0
        if b["in_row_begin"] + b["in_row_count"] < in_h:
            assert bot_gap >= _need_pad_out_rows(passes, in_h, out_h), (b, bot_gap)
        else:
            assert b["out_row_begin"] + b["out_row_count"] == out_h
    assert row == out_h


def test_plan_falls_back_to_one_band():
    import raisr_hip as R
    assert len(R.plan_bands(20, 40, 1, 4)) == 1            # too small to cut
    assert len(R.plan_bands(721, 1081, 1, 4)) == 1         # ratio that no band start can keep in phase
    b = R.plan_bands(721, 1081, 1, 4)[0]
    assert (b["in_row_begin"], b["in_row_count"], b["keep_begin"], b["keep_count"]) == (0, 721, 0, 1081)
    with pytest.raises(ValueError):
        R.plan_bands(0, 10, 1, 2)
