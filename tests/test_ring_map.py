"""CPU: the frame order of the (multi-device) stream ring as pure functions of the C ABI -- raisr_hip_ring_slot, what
raisr_hip_stream_submit / _collect walk, and raisr_hip_parse_device_list_n, what RAISR_HIP_DEVICES / RNLHandler_SetDeviceList
carry.  (north_star: frames shard across the GPUs of a node, frame i -> GPU i mod N; docs/performance.md:8-13 is the
reference's N-process counterpart.)  No GPU: nothing here creates a context."""
import raisr_hip as R


def test_frame_i_runs_on_device_i_mod_n_and_lanes_rotate():
    for n in (1, 2, 3, 8):
        for depth in (1, 2, 4):
            seen = {}
            for i in range(5 * n * depth):
                d, lane = R.ring_slot(n, depth, i)
                assert d == i % n                                   # the sharding rule
                assert lane == (i // n) % depth                     # consecutive frames of one device use its lanes in turn
                seen.setdefault((d, lane), []).append(i)
            assert len(seen) == n * depth                           # every lane is used
            for frames in seen.values():                            # a lane sees every (n * depth)-th frame: depth frames in flight per device
                assert all(b - a == n * depth for a, b in zip(frames, frames[1:]))


def test_collect_order_is_submission_order():
    # collect() waits for lane tail % (n * depth): the same walk as submit, so frame k is the k-th frame collected
    n, depth = 3, 2
    submitted = [R.ring_slot(n, depth, i) for i in range(20)]
    collected = [R.ring_slot(n, depth, k) for k in range(20)]
    assert submitted == collected


def test_device_lists():
    assert R.parse_device_list("0,1,2,3", 8) == [0, 1, 2, 3]
    assert R.parse_device_list(" 2 , 0 ", 4) == [2, 0]
    assert R.parse_device_list("0,0", 1) == [0, 0]                  # a device may be listed twice
    assert R.parse_device_list("all", 8) == list(range(8))
    assert R.parse_device_list("", 8) == []
    for bad in ("0,", ",0", "0;1", "x", "-1", "0,8", "1 2", "all,0"):
        assert R.parse_device_list(bad, 8) is None, bad
    assert R.parse_device_list("0", 0) is None                      # no such device
    assert R.parse_device_list("all", 0) is None
    assert R.parse_device_list(",".join(["0"] * 17), 8) is None     # more than the ring takes
