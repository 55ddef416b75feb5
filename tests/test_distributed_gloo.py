"""N>1 path on CPU: world_size-2 gloo run of the job's only collective (filter-bank blob broadcast)
plus the frame-sharding arithmetic and the max-over-ranks timing reduction bench.py uses."""
import os
import socket
import sys

import numpy as np
import pytest

from common import ROOT, folder


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    sys.path[:0] = [os.path.join(ROOT, "video-super-resolution-library_amd")]
    import torch
    import torch.distributed as dist
    import raisr_hip as R
    import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nbytes = R.lib().raisr_hip_model_blob_bytes(216, 4)
    blob = None
    if rank == 0:
        bank, qstr, qcoh, qa = R.read_model_folder(folder("filters_2x/filters_highres"), 8, 1)
        blob = R.pack_model_blob(bank, qstr, qcoh, qa)
    t = sharding.broadcast_model_blob(blob, nbytes, torch.device("cpu"), dist)
    mine = sharding.frames_for_rank(11, rank, world)
    slow = sharding.max_over_ranks(1.0 + rank, torch.device("cpu"), dist)
    all_t = sharding.gather_over_ranks(1.0 + rank, torch.device("cpu"), dist)
    names = sharding.gather_strings(f"node {rank}", dist)
    assert all_t == [1.0, 2.0] and names == ["node 0", "node 1"]
    import hashlib
    q.put((rank, hashlib.sha256(t.numpy().tobytes()).hexdigest(), mine, slow))
    dist.barrier()
    dist.destroy_process_group()


def test_blob_broadcast_and_frame_sharding_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, h0, f0, s0), (r1, h1, f1, s1) = res
    assert h0 == h1                                  # every rank holds the same filter bank bytes
    assert sorted(f0 + f1) == list(range(11)) and not set(f0) & set(f1)
    assert f0 == [0, 2, 4, 6, 8, 10] and f1 == [1, 3, 5, 7, 9]
    assert s0 == s1 == 2.0                           # timing = max over ranks
    # the broadcast bytes are the packed device layout
    sys.path[:0] = [os.path.join(ROOT, "video-super-resolution-library_amd")]
    import hashlib
    import raisr_hip as R
    bank, qstr, qcoh, qa = R.read_model_folder(folder("filters_2x/filters_highres"), 8, 1)
    assert hashlib.sha256(R.pack_model_blob(bank, qstr, qcoh, qa).tobytes()).hexdigest() == h0


def test_job_rate_divides_by_the_slowest_rank():
    """bench.py's N-rank `value`: every rank's frames over the MAX of the per-rank times, never a sum of per-rank rates."""
    sys.path[:0] = [os.path.join(ROOT, "video-super-resolution-library_amd")]
    import sharding
    rate, per_rank = sharding.job_rate([100, 100, 100, 100], [1.0, 1.0, 2.0, 1.0])
    assert rate == 400 / 2.0 and per_rank == [100.0, 100.0, 50.0, 100.0]
    assert rate < sum(per_rank)                                   # a straggler costs the whole job
    rate1, _ = sharding.job_rate([768], [0.5])
    assert rate1 == 1536.0
