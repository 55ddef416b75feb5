#!/usr/bin/env python3
"""bench.py -- MI355X Enhanced-RAISR hot path, BASELINE.json metric: output-Y megapixels/s.

Workload (config.workload): BASELINE.json configs[1] -- 1080p -> 4K 2x, filters_2x/filters_highres,
1-pass, 8-bit, CountOfBitsChanged blending, AVX-512-exact numerics.  A "step" is one pass of the hot
path (cheap upscale -> structure-tensor hash -> 11x11 filter -> CT blend) over one batch of
`--frames-per-step` synthetic 1080p Y planes that are already resident in HBM.  One process per
GPU; frames are sharded across ranks with no data-path collective (weak scaling); the only
collective is one RCCL broadcast of the packed filter-bank blob at start-up.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     -- dominant kernel (k_hash): algorithmic bytes per launch / average launch duration,
                  measured with HIP events on the kernel's own stream over the timed region
  cpu_baseline -- the CPU oracle ("port") timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "video-super-resolution-library_amd"))

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-step", type=int, default=24)
    ap.add_argument("--lanes", type=int, default=4, help="frames in flight per GPU (one context+stream each); 2-4 measure within 2 %%")
    ap.add_argument("--passes", type=int, default=0, help="override the config's pass count (1 or 2)")
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS), help="BASELINE.json configuration; the bench line is C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=100, help="bounded CPU-baseline sample (~10-15 s on 16 cores)")
    ap.add_argument("--frame-kind", default="natural", choices=["natural", "constant", "random", "checker"])
    return ap.parse_args()


HBM_PEAK_GBS = 8000.0                                        # MI355X_MICROARCH.md
VALU_FP32_PEAK_TFLOPS = 157.3                                # MI355X_MICROARCH.md "Peak FP32 (vector)": 4 SIMD-32 per CU, all-FMA.
# (scripts/valu_rate_probe.hip sustains 911 G wave-inst/s = 116.6 TFLOP/s of v_fma_f32 on this power-limited part; the
#  structure tensor's mix of 2 mul + 3 fma per tap can reach at most 1013/(2*605) = 84 % of an all-FMA peak.)
# SURVEY.md s8d per-filtered-pixel FLOP model (FMA = 2): structure tensor 121 x (2 mul + 3 fma) + 45 add, hash ~60,
# filter 121 fma + 15 add
HASH_FLOP_PER_PIXEL = 121 * (2 + 6) + 45 + 60
FILTER_FLOP_PER_PIXEL = 121 * 2 + 15

# BASELINE.json configs.  The bench line is C2 (configs[1]); the others are selectable for
# profiling: (in_w, in_h, out_w, out_h, folder, bits, passes, mode, hash variant, description)
CONFIGS = {
    "C1": (960, 540, 1920, 1080, "filters_2x/filters_lowres", 8, 1, 1, 1, "540p->1080p 2x, filters_lowres, 1-pass, 8-bit, AVX2-exact"),
    "C2": (1920, 1080, 3840, 2160, "filters_2x/filters_highres", 8, 1, 1, 2, "1080p->4K 2x, filters_2x/filters_highres, 1-pass, 8-bit, AVX512-exact"),
    "C3": (1920, 1080, 3840, 2160, "filters_2x/filters_highres", 8, 2, 1, 2, "1080p->4K 2x, filters_2x/filters_highres, 2-pass, 8-bit, AVX512-exact"),
    "C4": (1280, 720, 1920, 1080, "filters_1.5x/filters_denoise", 8, 2, 2, 5, "720p->1080p 1.5x, filters_denoise, 2-pass mode 2, 8-bit, AVX512FP16-exact"),
    "C5": (3840, 2160, 7680, 4320, "filters_2x/filters_highres", 10, 1, 1, 2, "4K->8K 2x, filters_highres, 1-pass, 10-bit, AVX512-exact"),
}
IN_W = IN_H = OUT_W = OUT_H = 0
FOLDER = ""
ALGO_BYTES_PER_FRAME = 0
CFG = None


def select_config(name, passes_override=None):
    """Sets the module-level workload.  Algorithmic bytes: Y in + Y out per frame, plus the
    intermediate write + read for two passes (C2: 10 368 000 B)."""
    global IN_W, IN_H, OUT_W, OUT_H, FOLDER, ALGO_BYTES_PER_FRAME, CFG
    iw, ih, ow, oh, folder, bits, passes, mode, asm, desc = CONFIGS[name]
    if passes_override:
        passes = passes_override
    IN_W, IN_H, OUT_W, OUT_H = iw, ih, ow, oh
    FOLDER = os.path.join(ROOT, *folder.split("/"))
    bps = 1 if bits == 8 else 2
    mid = (iw * ih if mode == 2 else ow * oh) * bps
    ALGO_BYTES_PER_FRAME = (iw * ih + ow * oh) * bps + (2 * mid if passes == 2 else 0)
    CFG = dict(name=name, bits=bits, passes=passes, mode=mode, asm=asm, desc=desc, pixel_types=4 if ow == 2 * iw else 1)


def cpu_baseline(sample_frames):
    """CPU oracle (test infrastructure, oracle/) timed on the host cores: kind "port"."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    import synth
    cores = len(os.sched_getaffinity(0))
    try:    # a cgroup CPU quota (e.g. "1600000 100000" = 16 CPUs) is the real core budget of this container
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    os.environ["OMP_NUM_THREADS"] = str(cores)
    O.lib()
    bits, passes, mode, asm = CFG["bits"], CFG["passes"], CFG["mode"], CFG["asm"]
    if asm == 5:
        p1 = O.make_pass16(FOLDER, bits, 1)
        p2 = O.make_pass16(FOLDER, bits, 2) if passes == 2 else None
        run = lambda f: O.process_y16(f, OUT_W, OUT_H, p1, p2, passes, mode)
    else:
        p1 = O.make_pass(O.Model(FOLDER, bits, 1), bits, False, asm)
        p2 = O.make_pass(O.Model(FOLDER, bits, 2), bits, False, asm) if passes == 2 else None
        run = lambda f: O.process_y(f, OUT_W, OUT_H, p1, p2, passes, mode)
    frames = [synth.natural_y(IN_W, IN_H, bits, seed=12345 + i) for i in range(min(sample_frames, 4))]
    run(frames[0])                                           # warm the thread pool / page in
    t0 = time.perf_counter()
    for i in range(sample_frames):
        run(frames[i % len(frames)])
    dt = time.perf_counter() - t0
    return {"value": round(OUT_W * OUT_H * sample_frames / dt / 1e6, 3), "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": f"{sample_frames} synthetic frames of the same workload ({CFG['name']}); C oracle (gcc -O3 -mavx2 auto-vectorised, "
                      f"strict IEEE), OpenMP row bands, threads = cgroup CPU quota, {dt:.2f}s"}


def main():
    args = parse()
    select_config(args.config, args.passes or None)
    import torch
    import torch.distributed as dist
    import raisr_hip as R
    import sharding
    import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP extension has no CPU fallback)")
    # one rank per GPU; ranks beyond the visible devices wrap around (only meaningful for the 2-ranks-on-1-GPU
    # plumbing test, which also swaps RCCL for gloo via RAISR_BENCH_BACKEND -- RCCL refuses two ranks on one device)
    gpu = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    # launched by torch.distributed.run (RANK set) -> always go through RCCL, even with one rank, so the
    # collective path of the N-GPU runs is the one exercised by a 1-GPU torchrun smoke test
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        backend = os.environ.get("RAISR_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- model: rank 0 reads the files and packs the device blob; RCCL broadcast to the others ----
    passes = CFG["passes"]
    bits = CFG["bits"]
    blobs = []
    for p in range(passes):
        host_blob = None
        if rank == 0:
            bank, qstr, qcoh, qa = R.read_model_folder(FOLDER, bits, p + 1)
            host_blob = R.pack_model_blob(bank, qstr, qcoh, qa)
        nbytes = R.lib().raisr_hip_model_blob_bytes(216, CFG["pixel_types"])
        blobs.append(sharding.broadcast_model_blob(host_blob, nbytes, dev, dist if use_dist else None))
    torch.cuda.synchronize()

    lanes = []
    for _ in range(args.lanes):
        d = R.RaisrDevice(gpu)
        for p in range(passes):
            d.set_model_blob_device(p, blobs[p].data_ptr(), blobs[p].numel())
        d.configure(IN_W, IN_H, OUT_W, OUT_H, bits=bits, passes=passes, mode=CFG["mode"], hash_variant=CFG["asm"])
        lanes.append(d)

    # ---- synthetic input, resident in HBM before the timed region ----
    nf = args.frames_per_step
    uniq = min(nf, 8)
    # each rank owns its own frames of the (virtual) stream: frame index = rank + world * i
    mine = sharding.frames_for_rank(uniq * world, rank, world)
    if args.frame_kind == "natural":
        host_frames = [synth.natural_y(IN_W, IN_H, bits, seed=12345 + i) for i in mine]
    elif args.frame_kind == "random":
        host_frames = [synth.random_y(IN_W, IN_H, bits, seed=777 + i) for i in mine]
    else:
        host_frames = [synth.FRAME_KINDS[args.frame_kind](IN_W, IN_H, bits) for _ in mine]
    d_in = [torch.from_numpy(f).to(dev) for f in host_frames]
    bps = 1 if bits == 8 else 2
    d_out = [torch.empty((OUT_H, OUT_W), dtype=torch.uint8 if bps == 1 else torch.uint16, device=dev) for _ in range(args.lanes)]
    torch.cuda.synchronize()

    def step():
        for f in range(nf):
            ln = f % args.lanes
            lanes[ln].process_y(d_in[f % uniq].data_ptr(), IN_W * bps, d_out[ln].data_ptr(), OUT_W * bps)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    timing = not args.no_kernel_timing
    if timing:                       # HIP events around every kernel of ONE lane (a third of the launches at 3 lanes):
        lanes[0].timing_enable(True)  # enough launches for the average, a third of the event traffic in the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    dt = sharding.max_over_ranks(dt, dev, dist if use_dist else None)

    # per-kernel event timings (rank 0's lanes)
    kern = {}
    if timing:
        for k, v in lanes[0].timing_read().items():
            kern[k] = {"total_ms": v["total_ms"], "count": v["count"]}
        lanes[0].timing_enable(False)

    # isolated per-kernel durations: the same frames on ONE lane after the timed region (no other kernel
    # shares the chip), so a launch's duration is its own -- the overlapped average above is not
    iso = {}
    if timing and rank == 0:
        lanes[0].timing_enable(True)
        for f in range(2 * uniq):
            lanes[0].process_y(d_in[f % uniq].data_ptr(), IN_W * bps, d_out[0].data_ptr(), OUT_W * bps)
        torch.cuda.synchronize()
        iso = {k: v["total_ms"] / max(1, v["count"]) for k, v in lanes[0].timing_read().items()}
        lanes[0].timing_enable(False)

    frames_total = args.steps * nf * world
    mp_s = OUT_W * OUT_H * frames_total / dt / 1e6

    if rank == 0:
        roofline = None
        kernels_ms = {k: round(v["total_ms"] / max(1, v["count"]), 4) for k, v in kern.items()}
        # dominant kernel: the fused tensor/hash + filter kernel (fp32 paths), the hash kernel of the binary16 pipeline
        if CFG["asm"] == 5:
            dom = "k_hashfilter16" if "k_hashfilter16" in kern else "k_hash16"
        else:
            dom = "k_hashfilter" if "k_hashfilter" in kern else "k_hash"
        if dom in kern and kern[dom]["count"]:
            # with two passes the kernel runs twice per frame: one launch still processes one frame-pass
            avg_s = kern[dom]["total_ms"] / kern[dom]["count"] * 1e-3
            algo_per_launch = ALGO_BYTES_PER_FRAME / passes
            achieved = algo_per_launch / avg_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
            if os.path.exists(tpath) and CFG["name"] == "C2":
                try:
                    tj = json.load(open(tpath))
                    traffic = tj.get("dominant_kernel_hbm_bytes_per_launch") if tj.get("dominant_kernel") == dom else None
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                        "avg_launch_ms": round(avg_s * 1e3, 4), "algorithmic_bytes_per_launch": int(algo_per_launch),
                        "lanes_overlapped": args.lanes,
                        "note": "path is fp32-VALU bound (~1.3 kFLOP per output pixel vs 1.25 compulsory bytes); "
                                "HBM fraction is reported as required, VALU utilisation is the binding figure (DESIGN.md)"}
            if dom in iso and iso[dom] > 0:
                # the binding resource: fp32 VALU.  Filtered zone of one launch x the s8d FLOP model / isolated duration
                c_final = 6 + 8 * ((OUT_W - 12) // 8)
                zone_w, zone_h = c_final - 6, OUT_H - 12
                if CFG["mode"] == 2 and passes == 2:         # pass 1 of mode 2 runs at input size; average the two launches
                    c1 = 6 + 8 * ((IN_W - 12) // 8)
                    px = ((c1 - 6) * (IN_H - 12) + zone_w * zone_h) / 2
                else:
                    px = zone_w * zone_h
                flop_px = HASH_FLOP_PER_PIXEL + (FILTER_FLOP_PER_PIXEL if dom.startswith("k_hashfilter") else 0)
                tflops = px * flop_px / (iso[dom] * 1e-3) / 1e12
                roofline["valu"] = {"kernel": dom, "isolated_launch_ms": round(iso[dom], 4),
                                    "flop_per_launch": int(px * flop_px), "achieved": round(tflops, 2),
                                    "peak": VALU_FP32_PEAK_TFLOPS, "unit": "TFLOP/s (fp32 VALU)",
                                    "frac": round(tflops / VALU_FP32_PEAK_TFLOPS, 4),
                                    # what scripts/valu_rate_probe.hip sustains on this (power-limited) part with pure v_fma_f32
                                    "sustained_peak": 116.6, "frac_of_sustained": round(tflops / 116.6, 4)}
                roofline["binding"] = "fp32-valu"
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(args.cpu_sample_frames)
            except Exception as e:  # the baseline is reported data, never a reason to lose the GPU line
                cpu = {"value": None, "unit": "MP/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        line = {
            "metric": "megapixels/sec (Y-plane) 1080p->4K 2x RAISR" if CFG["name"] in ("C2", "C3") else f"megapixels/sec (Y-plane) {CFG['name']}",
            "value": round(mp_s, 2), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{CFG['name']}: {CFG['desc']}, CT blend, frames resident in HBM" + (f" [passes={passes}]" if args.passes else ""),
                       "frame_kind": args.frame_kind,
                       "frames_per_step": nf, "lanes": args.lanes, "fps": round(frames_total / dt, 2),
                       "parallelism": f"frame-shard x{world}"},
            "kernels_avg_ms": kernels_ms,
            "kernels_isolated_ms": {k: round(v, 4) for k, v in iso.items()},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)

    for d in lanes:
        d.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
