"""Multi-GPU plumbing for the frame-sharded RAISR job (one process per GPU).

Frames are independent (RNLProcess is a pure per-frame function, reference Library/Raisr.cpp:1294),
so frame i of a stream goes to rank i mod world and no rank ever exchanges pixels.  (For latency rather than
throughput a single frame can also be cut into horizontal bands, one per rank: band_for_rank.)  The only
collective is one broadcast of the packed filter-bank blob(s) from the rank that read the model
files: RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.
"""
import numpy as np


def frames_for_rank(n_frames, rank, world):
    """Indices of the frames rank `rank` owns under round-robin sharding."""
    return list(range(rank, n_frames, world))


def band_for_rank(in_height, out_height, passes, rank, world):
    """Latency mode: ONE frame split over `world` GPUs.  Returns rank `rank`'s band of the plan of
    raisr_hip_plan_bands (dict with in_row_begin/in_row_count, out_row_begin/out_row_count, keep_begin/keep_count), or
    None when the frame is too small to give this rank a band.  Bands are independent sub-frames: the rank uploads
    input rows [in_row_begin, +in_row_count), runs the ordinary pipeline configured for that sub-frame and owns output
    rows [keep_begin, +keep_count) -- no pixels are exchanged between ranks (the padding rows are recomputed instead)."""
    import raisr_hip as R
    bands = R.plan_bands(in_height, out_height, passes, world)
    return bands[rank] if rank < len(bands) else None


def broadcast_model_blob(blob, nbytes, device, dist=None, src=0):
    """blob: numpy uint8 array on the source rank (None elsewhere).  Returns a torch uint8 tensor of
    `nbytes` on `device` holding the same bytes on every rank."""
    import torch
    if blob is not None:
        t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.uint8)).to(device)
        assert t.numel() == nbytes
    else:
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if dist is not None and dist.is_initialized():   # also with one rank: same code path as N ranks
        dist.broadcast(t, src=src)
    return t


def max_over_ranks(value, device, dist=None):
    """MAX-reduce a python float over ranks (bench timing contract)."""
    import torch
    if dist is None or not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device, dist=None):
    """Every rank's python float, in rank order, on every rank (per-rank timings of the scaling record)."""
    import torch
    if dist is None or not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def gather_strings(text, dist=None):
    """Every rank's short string (e.g. its NUMA binding) in rank order; [text] without a process group."""
    if dist is None or not dist.is_initialized():
        return [text]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, text)
    return out


def job_rate(units_per_rank, seconds_per_rank):
    """Whole-job rate of a sharded run: ALL ranks' units over the time of the SLOWEST rank (the timing contract of bench.py:
    barrier, timed region, barrier, MAX over ranks).  Also returns the per-rank rates for the record."""
    slowest = max(seconds_per_rank)
    per_rank = [u / s for u, s in zip(units_per_rank, seconds_per_rank)]
    return sum(units_per_rank) / slowest, per_rank
