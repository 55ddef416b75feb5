#!/usr/bin/env python3
"""bench.py -- MI355X Enhanced-RAISR hot path, BASELINE.json metric: output-Y megapixels/s.

Headline workload (config.workload): BASELINE.json configs[1] ("C2") -- 1080p -> 4K 2x, filters_2x/filters_highres,
1-pass, 8-bit, CountOfBitsChanged blending, AVX-512-exact numerics.  A "step" is one pass of the hot path
(cheap upscale -> structure-tensor hash -> 11x11 filter -> CT blend) over one batch of `--frames-per-step`
synthetic 1080p Y planes that are already resident in HBM.  One process per GPU; frames are sharded across
ranks with no data-path collective (weak scaling); the only collective is one RCCL broadcast of the packed
filter-bank blob at start-up.

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (re-executes under
torch.distributed.run, RCCL backend, one rank per device) and refuses to print a line whose world size is not N.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     -- dominant kernel: algorithmic bytes per launch / average launch duration, measured with HIP
                  events on the kernel's own stream over the timed region (+ the figures that bind: `l1` = the vector-L1
                  delivery floor of the filter stage over the isolated launch, `valu` = algorithmic FLOPs over the fp32-VALU peak)
  cpu_baseline -- the CPU oracle ("port") timed on this box's host cores on a bounded sample
and, at N = 1 (outside the timed region of `value`, never mixed into it):
  c3_2pass     -- the same loop for BASELINE.json configs[2] (2-pass: north_star's target configuration)
  end_to_end   -- host planes -> host planes through RNLHandler_Process (Y + both chroma planes, PCIe inclusive:
                  the reference's own methodology, docs/performance.md:8-13): frame buffers from RNLHandler_HostAlloc,
                  and `pageable_planes` for ordinary malloc'ed ones
  stream       -- host planes -> host planes through the library's pinned-ring batch entry (uploads, kernels and
                  downloads of neighbouring frames overlapped)
  stream_single_process -- the same through the ONE-process multi-device ring with this GPU listed twice (2 slots x 2 lanes): the
                  per-slot cost of the code path `--single-process --gpus N` takes on N GPUs
  parity       -- one frame of the workload against the CPU oracle: mismatching pixels and PSNR; `certify` = the certified
                  hash stage's self-check over every benchmarked frame
  configs      -- BASELINE.json's C1, C3, C4, C5 next to C2, plus C2b = the configuration the reference publishes its numbers on
                  (1080p->4K, bits=10, 1-pass: docs/performance.md:8-14): fps, isolated kernel times, both rooflines, self-check
  frame_kinds  -- C2 on natural (synthetic) / photo (REAL pictures: the photographs installed with this image's Python packages, plain /
                  JPEG-blocky / letterboxed / enlarged / mosaic, photos.py) / random / constant / 1-px-checkerboard frames (throughput
                  depends on content through the share of pixels that take the exact hash path)
  (the matrix-core "fast mode" of rounds 2-3 left the product in round 4: it measured slower than the exact path; a development
   build keeps it, docs/EXPERIMENTS.md)
At N > 1 the line carries `stream` only: every rank streaming host-resident frames through its pinned ring at the same
time (PCIe / host-memory contention next to the HBM-resident headline).
"""
import argparse
import hashlib
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "video-super-resolution-library_amd")
sys.path.insert(0, PKG)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0                                        # MI355X_MICROARCH.md
VALU_FP32_PEAK_TFLOPS = 157.3                                # MI355X_MICROARCH.md "Peak FP32 (vector)": 4 SIMD-32 per CU, all-FMA.
L1_PROBE_B_PER_CLK_CU = 55.0                                 # scripts/l1_width_probe.hip: 54-62 B/clk/CU delivered to the lanes (dword / dwordx4 loads)
N_CUS = 256
CLOCK_HZ = 2.4e9
VALU_FP16_PEAK_TFLOPS = 314.6                                # packed binary16 (v_pk_fma_f16): two lanes per fp32 lane, same issue rate
# (scripts/valu_rate_probe.hip sustains ~911 G wave-inst/s = ~116.6 TFLOP/s of v_fma_f32 on this power-limited part: profiles/valu_rate_probe.json)
# SURVEY.md s8d per-filtered-pixel ALGORITHMIC FLOP model (FMA = 2): structure tensor 121 x (2 mul + 3 fma) + 45 add,
# hash ~60, filter 121 fma + 15 add.  This is the work the reference's algorithm defines per pixel; a kernel that
# reaches the same bits with fewer instructions (certified approximate tensor) shows up as a higher achieved rate.
HASH_FLOP_PER_PIXEL = 121 * (2 + 6) + 45 + 60
FILTER_FLOP_PER_PIXEL = 121 * 2 + 15

# BASELINE.json configs.  The bench line is C2 (configs[1]); the others are selectable for
# profiling: (in_w, in_h, out_w, out_h, folder, bits, passes, mode, hash variant, description)
CONFIGS = {
    "C1": (960, 540, 1920, 1080, "filters_2x/filters_lowres", 8, 1, 1, 1, "540p->1080p 2x, filters_lowres, 1-pass, 8-bit, AVX2-exact"),
    "C2": (1920, 1080, 3840, 2160, "filters_2x/filters_highres", 8, 1, 1, 2, "1080p->4K 2x, filters_2x/filters_highres, 1-pass, 8-bit, AVX512-exact"),
    # C2b = the configuration the reference PUBLISHES its numbers on (docs/performance.md:8-14: 1080p->2160p, bits=10, passes=1, filters_highres;
    # 176.2 fps AVX-512 fp32 / 222.5 fps AVX-512 FP16 on one 60-core Xeon 8580+ socket) -- a side leg, never the headline
    "C2b": (1920, 1080, 3840, 2160, "filters_2x/filters_highres", 10, 1, 1, 2, "1080p->4K 2x, filters_2x/filters_highres, 1-pass, 10-bit, AVX512-exact (the reference's published configuration)"),
    "C3": (1920, 1080, 3840, 2160, "filters_2x/filters_highres", 8, 2, 1, 2, "1080p->4K 2x, filters_2x/filters_highres, 2-pass, 8-bit, AVX512-exact"),
    "C4": (1280, 720, 1920, 1080, "filters_1.5x/filters_denoise", 8, 2, 2, 5, "720p->1080p 1.5x, filters_denoise, 2-pass mode 2, 8-bit, AVX512FP16-exact"),
    "C5": (3840, 2160, 7680, 4320, "filters_2x/filters_highres", 10, 1, 1, 2, "4K->8K 2x, filters_highres, 1-pass, 10-bit, AVX512-exact"),
}
DOMINANT = ("k_hashfilter_ac", "k_hashfilter", "k_hashfilter16", "k_filter_lds16", "k_hash_ac", "k_hash", "k_hash16", "k_filter", "k_filter16")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=768,
                    help="frames per step (batch); the default keeps the timed region of the default run >= 2 s")
    ap.add_argument("--lanes", type=int, default=4, help="frames in flight per GPU (one context+stream each); 2-4 measure within 2 %%")
    ap.add_argument("--batch", type=int, default=0, help="frames per launch (raisr_hip_process_y_device_batch); 0 = auto: 8 for outputs up to 1080p "
                    "(C1, C4: one launch of one small frame cannot fill the chip), 1 above")
    ap.add_argument("--passes", type=int, default=0, help="override the config's pass count (1 or 2)")
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS), help="BASELINE.json configuration; the bench line is C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the c3_2pass / end_to_end / stream / parity / configs legs")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration legs (C1, C3, C4, C5)")
    ap.add_argument("--cpu-sample-frames", type=int, default=100, help="bounded CPU-baseline sample (~10-15 s on 16 cores)")
    ap.add_argument("--extra-frames", type=int, default=256, help="frames of each extra leg (c3_2pass, end_to_end, stream)")
    ap.add_argument("--frame-kind", default="natural", choices=["natural", "constant", "random", "checker", "photo"])
    ap.add_argument("--stream", action="store_true",
                    help="headline value from HOST-resident frames streamed through the pinned-ring batch entry "
                         "(PCIe inclusive; for the C5 600-frame stream use --config C5 --stream --frames-per-step 600 --steps 1)")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives --gpus N devices through the library's multi-device ring (raisr_hip_stream_create_multi: frame i "
                         "-> device i mod N, in-order collection, model handed to the devices by the in-process RCCL broadcast); host-resident "
                         "frames, PCIe inclusive.  The C++-host counterpart of the torchrun launch; prints its own JSON line.")
    return ap.parse_args()


class Workload:
    """One BASELINE.json configuration.  Algorithmic bytes: Y in + Y out per frame, plus the
    intermediate write + read for two passes (C2: 10 368 000 B)."""

    def __init__(self, name, passes_override=None):
        iw, ih, ow, oh, folder, bits, passes, mode, asm, desc = CONFIGS[name]
        if passes_override:
            passes = passes_override
        self.name, self.desc = name, desc
        self.in_w, self.in_h, self.out_w, self.out_h = iw, ih, ow, oh
        self.folder_rel = folder
        self.folder = os.path.join(ROOT, *folder.split("/"))
        self.bits, self.passes, self.mode, self.asm = bits, passes, mode, asm
        self.bps = 1 if bits == 8 else 2
        self.pixel_types = 4 if ow == 2 * iw else 1
        mid = (iw * ih if mode == 2 else ow * oh) * self.bps
        self.algo_bytes = (iw * ih + ow * oh) * self.bps + (2 * mid if passes == 2 else 0)

    def frames(self, kind, indices):
        import synth
        if kind == "natural":
            return [synth.natural_y(self.in_w, self.in_h, self.bits, seed=12345 + i) for i in indices]
        if kind == "random":
            return [synth.random_y(self.in_w, self.in_h, self.bits, seed=777 + i) for i in indices]
        if kind == "photo":          # real pictures installed with this image's Python packages (photos.py); 13 is coprime with sources x variants
            import photos
            return [photos.photo_y(self.in_w, self.in_h, self.bits, 13 * i) for i in indices]
        return [synth.FRAME_KINDS[kind](self.in_w, self.in_h, self.bits) for _ in indices]


def host_cores():
    cores = len(os.sched_getaffinity(0))
    try:    # a cgroup CPU quota (e.g. "1600000 100000" = 16 CPUs) is the real core budget of this container
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def oracle_runner(wl, intrinsics=False):
    """(callable frame -> output plane, description) of the CPU oracle (TEST INFRASTRUCTURE, oracle/) for `wl`.  intrinsics=True:
    the hand-vectorised AVX-512 twin of the fp32 pass (oracle/raisr_oracle_avx512.c; same bits, tests/test_oracle_avx512.py)
    where the host executes AVX-512 and the workload has fp32 numerics."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.lib()
    if wl.asm == 5:
        p1 = O.make_pass16(wl.folder, wl.bits, 1)
        p2 = O.make_pass16(wl.folder, wl.bits, 2) if wl.passes == 2 else None
        return (lambda f: O.process_y16(f, wl.out_w, wl.out_h, p1, p2, wl.passes, wl.mode)), f"C oracle, software binary16, built for {O.isa()} (compiler-vectorised)"
    p1 = O.make_pass(O.Model(wl.folder, wl.bits, 1), wl.bits, False, wl.asm)
    p2 = O.make_pass(O.Model(wl.folder, wl.bits, 2), wl.bits, False, wl.asm) if wl.passes == 2 else None
    if intrinsics and O.lib512() is not None:
        return (lambda f: O.process_y_intrinsics(f, wl.out_w, wl.out_h, p1, p2, wl.passes, wl.mode)), \
            "own AVX-512 intrinsics (oracle/raisr_oracle_avx512.c: lane = pixel structure tensor, vectorised integer models of VRCP14/VRSQRT14/RCPPS/RSQRTPS, " \
            "permuted 121-tap dot product; bit-identical to the scalar oracle)"
    return (lambda f: O.process_y(f, wl.out_w, wl.out_h, p1, p2, wl.passes, wl.mode)), f"C oracle built for {O.isa()} (strict IEEE, compiler-vectorised)"


def cpu_baseline(wl, sample_frames):
    """CPU implementation of the same workload timed on the host cores: kind "port" (the reference itself cannot be built here).
    fp32 numerics: the build's own AVX-512 implementation, row bands over OpenMP threads = cgroup CPU quota, best of
    five batches after a warm-up; binary16 numerics and hosts without AVX-512: the scalar C oracle, compiler-vectorised."""
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    os.environ.setdefault("RAISR_ORACLE_ISA", "auto")
    try:    # the OpenMP runtime is already initialised when an earlier leg loaded the oracle: the environment variable comes too
        import ctypes                                   # late then (it would run one thread per visible CPU, far above the quota)
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass

    def timed(w, n_frames):
        run, how = oracle_runner(w, intrinsics=True)
        frames = w.frames("natural", range(min(n_frames, 4)))
        run(frames[0])                                       # warm the thread pool / page in
        per = max(1, n_frames // 5)
        best = None
        for b in range(5):
            t0 = time.perf_counter()
            for i in range(per):
                run(frames[i % len(frames)])
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return w.out_w * w.out_h * per / best / 1e6, per, best, how

    v, per, best, how = timed(wl, sample_frames)
    out = {"value": round(v, 3), "unit": "MP/s", "cores": cores, "kind": "port",
           "sample": f"best of 5 batches of {per} synthetic frames of the same workload ({wl.name}, {wl.passes}-pass); {how}; OpenMP row bands, "
                     f"threads = cgroup CPU quota, on {cpu_model_name()}, {best:.2f}s per batch"}
    if wl.name == "C2" and wl.passes == 1 and sample_frames >= 8:        # the 2-pass figure beside it (north_star's target config)
        w3 = Workload("C3")
        v3, per3, best3, _ = timed(w3, max(5, sample_frames // 2))
        out["two_pass"] = {"value": round(v3, 3), "unit": "MP/s", "sample": f"best of 5 batches of {per3} frames of C3, {best3:.2f}s per batch"}
    return out


def source_hash():
    """sha256 over the device code and its launch logic (csrc/*.hip, csrc/*.h; not the plugin layer's *.cpp nor the host-side
    copy helpers of host_copy.h): a committed PMC
    traffic figure is only quoted for the kernels it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")) and fn != "host_copy.h":
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()


def measured_traffic(kernel, config="C2"):
    """HBM-side bytes per launch of `kernel` in configuration `config` from the newest profiles/traffic_r*.json recorded for
    that configuration whose source hash is the hash of the kernel sources in this tree (the PMC passes are separate
    rocprofv3 runs, scripts/profile_gpu.sh); None otherwise -- a stale figure is not quoted."""
    pdir = os.path.join(ROOT, "profiles")
    try:
        cands = sorted(f for f in os.listdir(pdir) if f.startswith("traffic_r") and f.endswith(".json"))
    except OSError:
        return None, "no profiles/ directory"
    cur = source_hash()
    for fn in reversed(cands):
        try:
            j = json.load(open(os.path.join(pdir, fn)))
        except (OSError, ValueError):
            continue
        if j.get("source_sha256") != cur or j.get("config", "C2") != config or j.get("variant_env"):
            continue
        if kernel in j.get("per_kernel_bytes", {}):
            return int(j["per_kernel_bytes"][kernel]), fn
    return None, "no PMC pass recorded for these kernel sources (run scripts/profile_gpu.sh)"


def measured_binding(kernel, config="C2"):
    """What keeps `kernel` busy in configuration `config`, from the counters of the newest profiles/traffic_r*.json recorded for these
    kernel sources (scripts/summarize_profiles.py writes a `binding` object next to the traffic: vector-instruction rate against the
    measured v_fma_f32 rate, LDS-busy share, waiting share); (None, why) otherwise -- figures of another kernel or configuration are not
    quoted (round 5 printed C2's typed-in percentages on C1's line)."""
    pdir = os.path.join(ROOT, "profiles")
    try:
        cands = sorted(f for f in os.listdir(pdir) if f.startswith("traffic_r") and f.endswith(".json"))
    except OSError:
        return None, "no profiles/ directory"
    cur = source_hash()
    for fn in reversed(cands):
        try:
            j = json.load(open(os.path.join(pdir, fn)))
        except (OSError, ValueError):
            continue
        if j.get("source_sha256") != cur or j.get("config", "C2") != config or j.get("variant_env"):
            continue
        b = j.get("binding")
        if b and b.get("kernel") == kernel:
            return b, fn
    return None, "no counter pass recorded for these kernel sources and this configuration (run scripts/profile_gpu.sh)"


def valu_probe_rate():
    """Sustained v_fma_f32 rate of this part (scripts/valu_rate_probe.hip) as recorded in profiles/valu_rate_probe.json: (TFLOP/s, source)."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "valu_rate_probe.json")))
        return float(j["v_fma_f32_TFLOP_per_s"]), "profiles/valu_rate_probe.json (" + j.get("from", "") + ")"
    except (OSError, ValueError, KeyError):
        return None, "no probe record (scripts/valu_rate_probe.hip)"


def bind_to_gpu_numa_node(torch, gpu):
    """N > 1: run this rank on the CPUs of the NUMA node its GPU hangs off, so that the page-locked frame planes of the streamed leg
    (first touch) and the threads that fill them are local to the PCIe root (NUMA placement of pinned buffers).
    Returns a short description for the JSON line; never fatal."""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(32)
        if hip.hipDeviceGetPCIBusId(buf, 32, gpu) != 0:
            return "unknown (no PCI bus id)"
        bdf = buf.value.decode()
        node = int(open(f"/sys/bus/pci/devices/{bdf.lower()}/numa_node").read())
        if node < 0:
            return "single node"
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = cpus & os.sched_getaffinity(0)
        if not allowed:
            return f"node {node} (not in this process's CPU set: unbound)"
        os.sched_setaffinity(0, allowed)
        return f"node {node} ({len(allowed)} CPUs)"
    except Exception as e:  # placement is an optimisation, never a reason to lose the line
        return f"unbound ({type(e).__name__})"


def respawn_ranks(args):
    """--gpus N > 1 outside torchrun: start the N ranks ourselves, exactly as the driver would."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def auto_batch(wl):
    return 8 if wl.out_w * wl.out_h <= 1920 * 1080 else 1


def device_loop(R, torch, wl, gpu, blobs, lanes_n, host_frames, nf, steps, warmup, fence, timing, fast=None, batch=1):
    """Frames resident in HBM -> output planes in HBM, `lanes_n` launches in flight, `batch` frames per launch (one launch per
    kernel for the whole batch: raisr_hip_process_y_device_batch).  Returns (seconds, kernel timings, lanes, device inputs/outputs)."""
    dev = torch.device("cuda", gpu)
    lanes = []
    for _ in range(lanes_n):
        d = R.RaisrDevice(gpu)
        for p in range(wl.passes):
            d.set_model_blob_device(p, blobs[p].data_ptr(), blobs[p].numel())
        d.configure(wl.in_w, wl.in_h, wl.out_w, wl.out_h, bits=wl.bits, passes=wl.passes, mode=wl.mode, hash_variant=wl.asm)
        if fast is not None:
            d.set_fast(fast)
        lanes.append(d)
    tdt = torch.uint8 if wl.bps == 1 else torch.uint16
    if batch > 1:
        # the batch entry wants equally spaced planes: the frames in one allocation, every lane's outputs in another
        while len(host_frames) % batch:
            host_frames = host_frames + host_frames[:batch - len(host_frames) % batch]
        hs = np.stack(host_frames)
        all_in = torch.from_numpy(hs.view(np.int16) if wl.bps == 2 else hs).to(dev).view(tdt)
        d_in = [all_in[i] for i in range(all_in.shape[0])]
        out_blocks = [torch.empty((batch, wl.out_h, wl.out_w), dtype=tdt, device=dev) for _ in range(lanes_n)]
        d_out = [b[0] for b in out_blocks]
        in_ptrs = [[d_in[g * batch + i].data_ptr() for i in range(batch)] for g in range(len(d_in) // batch)]
        out_ptrs = [[b[i].data_ptr() for i in range(batch)] for b in out_blocks]
    else:
        d_in = [torch.from_numpy(f).to(dev) for f in host_frames]
        d_out = [torch.empty((wl.out_h, wl.out_w), dtype=tdt, device=dev) for _ in range(lanes_n)]
    torch.cuda.synchronize()
    uniq = len(d_in)

    def step():
        if batch > 1:
            for k in range(nf // batch):                       # nf is a multiple of the batch (main() rounds it)
                ln = k % lanes_n
                lanes[ln].process_y_batch(in_ptrs[k % len(in_ptrs)], wl.in_w * wl.bps, out_ptrs[ln], wl.out_w * wl.bps)
            return
        for f in range(nf):
            ln = f % lanes_n
            lanes[ln].process_y(d_in[f % uniq].data_ptr(), wl.in_w * wl.bps, d_out[ln].data_ptr(), wl.out_w * wl.bps)

    for _ in range(warmup):
        step()
    fence()
    if timing:                        # HIP events around every kernel of ONE lane: enough launches for the average,
        lanes[0].timing_enable(True)  # a fraction of the event traffic in the timed region
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    kern = {}
    if timing:
        for k, v in lanes[0].timing_read().items():
            kern[k] = {"total_ms": v["total_ms"], "count": v["count"]}
        lanes[0].timing_enable(False)
    return dt, kern, lanes, d_in, d_out


def isolated_kernel_ms(lanes, d_in, d_out, wl, torch, iters=24):
    """The same frames on ONE lane with nothing else on the chip: a launch's duration is its own."""
    lanes[0].timing_enable(True)
    for i in range(iters):
        lanes[0].process_y(d_in[i % len(d_in)].data_ptr(), wl.in_w * wl.bps, d_out[0].data_ptr(), wl.out_w * wl.bps)
    torch.cuda.synchronize()
    iso = {k: v["total_ms"] / v["count"] for k, v in lanes[0].timing_read().items() if v["count"]}
    lanes[0].timing_enable(False)
    return iso


def symmetric_model(wl):
    """Does the library run its symmetric filter stage for this workload's FIRST pass?  (at most 16 non-palindromic rows in the bank:
    csrc/device_abi.hip scan_bank_symmetry; RAISR_HIP_SYM=0 switches it off)"""
    if wl.asm == 5 or os.environ.get("RAISR_HIP_SYM", "1") == "0":
        return False
    import glob
    f = glob.glob(os.path.join(wl.folder, f"filterbin_*_{wl.bits}"))
    if not f:
        return False
    raw = open(f[0], "rb").read()
    rows = np.frombuffer(raw[16:], np.uint32).reshape(-1, 121)
    return int((rows != rows[:, ::-1]).any(axis=1).sum()) <= int(os.environ.get("RAISR_HIP_SYM_MAX_ROWS", "16"))


def filtered_zone_px(w, h):
    c_final = 6 + 8 * ((w - 12) // 8)
    return max(0, c_final - 6) * max(0, h - 12)


def roofline_of(wl, kern, iso, lanes_n, batch=1):
    """`roofline` object of one workload from the HIP-event timings of its kernels: `kern` = per-kernel totals over the timed
    region (lanes_n frames in flight), `iso` = per-launch milliseconds of the same kernels with nothing else on the chip."""
    roofline = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
    dom = next((k for k in DOMINANT if k in kern and kern[k]["count"]), None)
    if not dom:
        return roofline
    per_launch = kern[dom]["total_ms"] / kern[dom]["count"] * 1e-3           # seconds
    launches_per_frame = wl.passes                                           # one launch of the dominant kernel = one pass of one frame
    achieved = wl.algo_bytes * batch / launches_per_frame / per_launch / 1e9   # (of `batch` frames when the batch entry is in use)
    traffic, traffic_src = measured_traffic(dom, wl.name)
    # NOTE: `avg_launch_ms` is measured while `lanes` frames are in flight, so launches of different lanes share the
    # chip and each one is stretched accordingly; `isolated_launch_ms` is the same launch alone on the chip (the duration
    # roofline.valu divides by).  Throughput follows the overlapped figure: fps ~ lanes / (sum of the overlapped kernel times).
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(per_launch * 1e3, 4),
                "isolated_launch_ms": round(iso[dom], 4) if dom in iso else None,
                "algorithmic_bytes_per_launch": wl.algo_bytes * batch // launches_per_frame,
                "frames_per_launch": batch,
                "lanes_overlapped": lanes_n,
                "note": "the path is neither HBM- nor MFMA-bound (~1.3 kFLOP per output pixel vs 1.25 compulsory bytes): the HBM fraction is reported "
                        "as required; roofline.valu (algorithmic FLOPs / isolated time vs the fp32 vector peak) and roofline.binding say what limits it (DESIGN.md s5)"}
    if dom in iso:
        roofline["frac_isolated"] = round(wl.algo_bytes / launches_per_frame / (iso[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)
    # the binding roofline: algorithmic FLOPs of the hash+filter stages over their isolated durations
    stage_kernels = [k for k in iso if k.startswith(("k_hash", "k_filter", "k_fix"))]
    if stage_kernels:
        zone = filtered_zone_px(wl.out_w, wl.out_h)
        if wl.mode == 2 and wl.passes == 2:          # pass 1 of mode 2 runs at input size; average the two launches
            zone = (filtered_zone_px(wl.in_w, wl.in_h) + zone) / 2
        flop = zone * (HASH_FLOP_PER_PIXEL + FILTER_FLOP_PER_PIXEL)
        t_iso = sum(iso[k] for k in stage_kernels) * 1e-3
        tflops = flop / t_iso / 1e12
        fp16 = wl.asm == 5
        peak = VALU_FP16_PEAK_TFLOPS if fp16 else VALU_FP32_PEAK_TFLOPS
        roofline["valu"] = {"kernels": stage_kernels, "isolated_ms": round(t_iso * 1e3, 4),
                            "flop_per_frame_pass": int(flop), "achieved": round(tflops, 2), "peak": peak,
                            "unit": ("TFLOP/s (packed binary16 VALU" if fp16 else "TFLOP/s (fp32 VALU") + ", algorithmic FLOPs of SURVEY s8d)",
                            "frac": round(tflops / peak, 4)}
        if not fp16:                                 # what scripts/valu_rate_probe.hip sustains on this (power-limited) part with pure v_fma_f32
            sp, sp_src = valu_probe_rate()
            if sp:
                roofline["valu"].update({"sustained_peak": sp, "sustained_peak_source": sp_src, "frac_of_sustained": round(tflops / sp, 4)})
        roofline["binding"] = "f16-valu" if fp16 else "fp32-valu"
        # Coefficient delivery through the vector L1 (profiles/r03_l1_width_probe.md: ~55 B/clk/CU whatever the load width).  Round 3 took
        # this floor for the binding one; round 4 halved the bytes (symmetric filter stage) and the kernel moved 2 %: it is reported as a
        # floor, not as the bound.  What binds the kernel is issue / latency at 16 waves per CU (docs/EXPERIMENTS.md I.1).
        bytes_px = 256 if (fp16 or symmetric_model(wl)) else 512
        floor_ms = zone * bytes_px / (L1_PROBE_B_PER_CLK_CU * N_CUS * CLOCK_HZ) * 1e3
        roofline["l1"] = {"coefficient_bytes_per_pixel": bytes_px, "probe_rate_B_per_clk_per_cu": L1_PROBE_B_PER_CLK_CU,
                          "floor_ms": round(floor_ms, 4), "isolated_ms": round(iso[dom], 4) if dom in iso else None,
                          "frac": round(floor_ms / iso[dom], 4) if dom in iso else None,
                          "what": "vector-L1 delivery floor of the filter stage's coefficients / isolated launch of the dominant kernel (a floor, not the bound)"}
        # what the dominant kernel keeps busy: counters of THIS configuration on THESE kernel sources, or nothing (no typed-in figures)
        b, bsrc = measured_binding(dom, wl.name)
        l1f = roofline["l1"]["frac"]
        if b:
            pct = lambda x: "n/a" if x is None else f"{100 * x:.0f} %"
            roofline["binding"] = {
                "what": ("binary16 " if fp16 else "fp32 ") + "vector-instruction issue at the occupancy LDS and registers allow, no pipe saturated (DESIGN.md s5)",
                "valu_of_probe_rate": b.get("valu_of_probe_rate"), "lds_busy": b.get("lds_busy"), "vector_l1_floor_frac": l1f,
                "wave_cycles_waiting_for_an_instruction": b.get("wave_cycles_waiting_for_an_instruction"),
                "summary": f"vector ALU {pct(b.get('valu_of_probe_rate'))} of the measured v_fma_f32 rate, LDS {pct(b.get('lds_busy'))} busy, "
                           f"vector-L1 coefficient floor {pct(l1f)} of the isolated launch, {pct(b.get('wave_cycles_waiting_for_an_instruction'))} of the wave cycles waiting for an instruction",
                "source": f"profiles/{bsrc}: {b.get('from', '')}"}
        else:
            roofline["binding"] = {"what": ("f16-valu" if fp16 else "fp32-valu") + " instruction issue (DESIGN.md s5)", "vector_l1_floor_frac": l1f, "summary": None, "source": bsrc}
    return roofline


class _stdout_to_stderr:
    """The library prints the reference's banner / [RAISR ...] messages on the C stdout (buffered when piped): keep them off
    this process's stdout, where the ONE JSON line goes."""

    def __enter__(self):
        import ctypes
        self._libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def end_to_end_leg(R, wl, n_frames, gpu):
    """Reference methodology (docs/performance.md:8-13): host yuv planes in, host yuv planes out, one synchronous
    RNLHandler_Process per frame, Y + U + V, PCIe inclusive.  Measured twice: with frame buffers from RNLHandler_HostAlloc (what the
    buffer pools of ffmpeg/vf_raisr_hip.diff hand to the filter) and with ordinary numpy (malloc'ed, pageable) planes."""
    import synth
    ratio = wl.out_w / wl.in_w
    ys_np = wl.frames("natural", range(4))
    cw, ch = wl.in_w // 2, wl.in_h // 2
    ocw, och = int(cw * ratio), int(ch * ratio)
    dt_ = ys_np[0].dtype
    u_np = synth.chroma(cw, ch, wl.bits)
    v_np = synth.chroma(cw, ch, wl.bits)

    def run(alloc):
        keep = []

        def plane(shape, fill=None):
            if alloc == "host_alloc":
                hp = R.HostPlane(shape, dt_)
                keep.append(hp)
                a = hp.array
            else:
                a = np.empty(shape, dt_)
            a[...] = 0 if fill is None else fill
            return a
        ys = [plane(y.shape, y) for y in ys_np]
        u, v = plane((ch, cw), u_np), plane((ch, cw), v_np)
        oy, ou, ov = plane((wl.out_h, wl.out_w)), plane((och, ocw)), plane((och, ocw))
        R.RNLHandler_SetOpenCLContext(0, gpu)
        with _stdout_to_stderr():
            rc = R.RNLHandler_Init(wl.folder, ratio, wl.bits, R.VideoRange, 20, R.HIP if wl.asm == 2 else wl.asm, wl.passes, wl.mode)
        if rc != R.RNLErrorNone:
            raise RuntimeError(f"RNLHandler_Init rc={rc:#x}")
        try:
            if R.RNLHandler_SetRes((ys[0], u, v), (oy, ou, ov)) != R.RNLErrorNone:
                raise RuntimeError("RNLHandler_SetRes failed")
            tw, i = time.perf_counter(), 0
            while i < 8 or time.perf_counter() - tw < 0.4:      # untimed warm-up: this leg may be the first GPU work of the process (clock ramp)
                R.RNLHandler_Process((ys[i % 4], u, v), (oy, ou, ov))
                i += 1
            t0 = time.perf_counter()
            for i in range(n_frames):
                if R.RNLHandler_Process((ys[i % 4], u, v), (oy, ou, ov)) != R.RNLErrorNone:
                    raise RuntimeError("RNLHandler_Process failed")
            return n_frames / (time.perf_counter() - t0)
        finally:
            R.RNLHandler_Deinit()
            for hp in keep:
                hp.close()
    fps = run("host_alloc")
    fps_pageable = run("malloc")
    mp = wl.out_w * wl.out_h / 1e6
    return {"value": round(fps * mp, 2), "unit": "MP/s", "fps": round(fps, 2), "frames": n_frames,
            "what": "host->host yuv420p through RNLHandler_Process (synchronous, one call per frame; frame buffers from RNLHandler_HostAlloc, as the "
                    "FFmpeg filter's pools provide them: page-locked, so the last pass runs in 3 row ranges with the finished rows downloaded early; "
                    "Y+U+V, PCIe inclusive)",
            "pageable_planes": {"value": round(fps_pageable * mp, 2), "unit": "MP/s", "fps": round(fps_pageable, 2),
                                "what": "the same with ordinary malloc'ed planes: carried through the library's own page-locked bounce memory "
                                        "(rows packed / unpacked by 4 threads, range by range; csrc/host_copy.h) -- the library never hands pageable "
                                        "memory to an asynchronous copy and never page-locks memory it does not own unless RAISR_HIP_PIN=1"}}


def stream_leg(R, wl, gpu, n_frames, collect_outputs=0, blobs=None, lanes_per_device=4):
    """Host planes -> host planes through the library's ring (raisr_hip_stream_*): uploads, kernels and downloads of
    neighbouring frames overlap.  The planes are page-locked (raisr_hip_host_alloc), as a host with its own buffer pool would
    hand them over.  Returns the JSON object (and, for tests, copies of the first `collect_outputs` Y outputs)."""
    import synth
    ys_np = wl.frames("natural", range(4))
    cw, ch = wl.in_w // 2, wl.in_h // 2
    ratio = wl.out_w / wl.in_w
    ocw, och = int(cw * ratio), int(ch * ratio)
    devices = list(gpu) if isinstance(gpu, (list, tuple)) else None        # a list: the multi-device ring, one host thread
    depth = lanes_per_device * (len(devices) if devices else 1)             # frames in flight
    dt = ys_np[0].dtype
    pins = []

    def pinned(shape, fill=None):
        pl = R.PinnedPlane(shape, dt)
        pins.append(pl)
        if fill is not None:
            pl.array[...] = fill
        return pl.array
    ys = [pinned(y.shape, y) for y in ys_np]
    u = pinned((ch, cw), synth.chroma(cw, ch, wl.bits))
    frames_out = [R.PinnedFrame(wl.out_w, wl.out_h, ocw, och, wl.bits) for _ in range(depth)]     # packed: one D2H copy per frame
    pins.extend(frames_out)
    outs = [(f.y, f.u, f.v) for f in frames_out]
    st = R.RaisrStream(gpu, wl.folder, wl.in_w, wl.in_h, wl.out_w, wl.out_h, bits=wl.bits, passes=wl.passes, mode=wl.mode,
                       hash_variant=wl.asm, chroma=(cw, ch, ocw, och), depth=lanes_per_device,
                       blobs=None if blobs is None else [(b.data_ptr(), b.numel()) for b in blobs])    # the broadcast blob, not the files
    assert st.depth == depth
    kept = []
    try:
        for warm in (True, False):
            n = 8 if warm else n_frames
            t0 = time.perf_counter()
            inflight = 0
            done = 0
            for i in range(n):
                if inflight == depth:
                    st.collect()
                    if not warm and done < collect_outputs:
                        kept.append(outs[done % depth][0].copy())
                    done += 1
                    inflight -= 1
                oy, ou, ov = outs[i % depth]
                st.submit(ys[i % 4], u, u, oy, ou, ov)
                inflight += 1
            while inflight:
                st.collect()
                if not warm and done < collect_outputs:
                    kept.append(outs[done % depth][0].copy())
                done += 1
                inflight -= 1
            dt_s = time.perf_counter() - t0
    finally:
        st.close()
        for pl in pins:
            pl.close()
    fps = n_frames / dt_s
    res = {"value": round(fps * wl.out_w * wl.out_h / 1e6, 2), "unit": "MP/s", "fps": round(fps, 2), "frames": n_frames,
           "what": f"host->host yuv420p through the stream ring (depth {depth}: H2D of frame n+1 and D2H of frame n-1 overlap frame n's "
                   "kernels; page-locked planes, Y+U+V, PCIe inclusive)"}
    if devices:
        res["devices"] = devices
        res["what"] += f"; ONE process, {len(devices)} device slot(s) x {lanes_per_device} lanes, frame i -> device i mod {len(devices)}"
    return (res, kept) if collect_outputs else res


def parity_leg(R, wl, gpu, blobs, kind, fast=None):
    """One frame of the workload, HIP vs the CPU oracle (the checker, never the thing measured)."""
    import torch
    frame = wl.frames(kind, [0])[0]
    run, _ = oracle_runner(wl)
    ref = run(frame)
    d = R.RaisrDevice(gpu)
    for p in range(wl.passes):
        d.set_model_blob_device(p, blobs[p].data_ptr(), blobs[p].numel())
    d.configure(wl.in_w, wl.in_h, wl.out_w, wl.out_h, bits=wl.bits, passes=wl.passes, mode=wl.mode, hash_variant=wl.asm)
    if fast is not None:
        d.set_fast(fast)
    out = np.zeros((wl.out_h, wl.out_w), frame.dtype)
    d.process_host(frame, out)
    d.close()
    torch.cuda.synchronize()
    diff = out.astype(np.int64) - ref.astype(np.int64)
    bad = int((diff != 0).sum())
    mse = float((diff * diff).mean())
    peak = float((1 << wl.bits) - 1)
    return {"mismatches": bad, "pixels": int(out.size), "max_abs_diff": int(np.abs(diff).max()),
            "psnr": "inf" if mse == 0 else round(10 * np.log10(peak * peak / mse), 3), "vs": "CPU oracle (oracle/, parity unpinned)",
            "frame": f"{wl.name} {kind} frame 0, Y plane"}


def certify_leg(R, wl, gpu, blobs, frames):
    """Self-check of the certified hash stage on the frames that were benchmarked: every pixel ALSO takes the reference's exact
    instruction sequence and a certified bucket that differs from it is counted (`certified_wrong`, must be 0);
    `uncertified_frac` = share of the filtered pixels the production kernel sends to the exact path."""
    if wl.asm == 5:
        return {"applies": False, "why": "binary16 numerics: every pixel takes the exact path (DESIGN.md s5: no sound bound for binary16 accumulation)"}
    res = {}
    for check in (False, True):
        d = R.RaisrDevice(gpu, hooks=True)                     # the self-check hooks live in the test-hooks flavour of the library (same kernels)
        try:
            for p in range(wl.passes):
                d.set_model_blob_device(p, blobs[p].data_ptr(), blobs[p].numel())
            d.configure(wl.in_w, wl.in_h, wl.out_w, wl.out_h, bits=wl.bits, passes=wl.passes, mode=wl.mode, hash_variant=wl.asm)
            d.certify_debug(True, check)
            out = np.zeros((wl.out_h, wl.out_w), frames[0].dtype)
            for f in frames:
                d.process_host(f, out)
            st = d.certify_stats()
        finally:
            d.close()
        if check:
            res["certified_wrong"] = st["mismatches"]
            res["buckets_compared"] = st["pixels"]
        else:
            res["uncertified_frac"] = round(st["uncertain"] / max(1, st["pixels"]), 6)
            if st.get("tiles"):            # the worklist per 64 x 16 tile: tiles with a non-empty list; tiles whose list overflowed (all-exact stage)
                res["tiles_listed_frac"] = round(st["tiles_listed"] / st["tiles"], 4)
                res["tiles_overflow_frac"] = round(st["tiles_overflow"] / st["tiles"], 4)
                res["tiles_flat_frac"] = round(st["tiles_flat"] / st["tiles"], 4)
    res["frames"] = len(frames)
    return res


def config_leg(R, torch, name, gpu, load_blobs, lanes_n, n_frames, fence, kind, batch=0):
    """One BASELINE.json configuration as first-class evidence: the headline loop (frames resident in HBM, `lanes_n` in flight),
    per-kernel HIP-event timings, isolated launch durations, both rooflines, and the certified hash stage's self-check on the
    frames that ran."""
    w = Workload(name)
    b = load_blobs(w)
    frames = w.frames(kind, range(8 if w.out_w <= 3840 else 4))
    batch = batch or auto_batch(w)
    # small outputs: more frames, so that every configuration's timed loop covers about the pixel volume of the 4K legs (a
    # 256-frame loop of 1080p outputs lasts 11 ms: start-up and tail of 4 lanes x 8 launches weigh 10 % there)
    n_frames *= max(1, (3840 * 2160) // (w.out_w * w.out_h))
    n_frames = max(batch, n_frames // batch * batch)
    dt, kern, lanes, d_in, d_out = device_loop(R, torch, w, gpu, b, lanes_n, frames, n_frames, 1, 1, fence, True, batch=batch)
    iso = isolated_kernel_ms(lanes, d_in, d_out, w, torch, iters=12)
    for d in lanes:
        d.close()
    del d_in, d_out
    fps = n_frames / dt
    return {"workload": w.desc, "fps": round(fps, 2), "value": round(w.out_w * w.out_h * fps / 1e6, 2), "unit": "MP/s", "frames": n_frames,
            "frames_per_launch": batch,
            "kernels_isolated_ms": {k: round(v, 4) for k, v in iso.items()},
            "roofline": roofline_of(w, kern, iso, lanes_n, batch),
            "certify": certify_leg(R, w, gpu, b, frames[:4])}


def single_process_main(args):
    """--single-process: the multi-device ring of the C ABI (what a C++ host links: raisr_hip_stream_create_multi), N devices, one
    process, one thread.  K steps of frames_per_step host-resident frames, W warm-up steps; value = frames / wall time, PCIe inclusive.
    No torch.distributed: the model reaches the devices by raisr_hip_broadcast_model_blob_devices inside the library."""
    import raisr_hip as R
    wl = Workload(args.config, args.passes or None)
    have = R.lib().raisr_hip_device_count()
    if have < 1:
        raise SystemExit("bench.py needs a GPU (the HIP extension has no CPU fallback)")
    devs = os.environ.get("RAISR_BENCH_DEVICES")                       # e.g. "0,0": one GPU standing in for two (plumbing tests)
    devices = [int(t) for t in devs.split(",")] if devs else list(range(args.gpus))
    if len(devices) != args.gpus or any(d >= have for d in devices):
        raise SystemExit(f"bench.py --single-process: --gpus {args.gpus} needs {args.gpus} devices, found {have}")
    nf = min(args.frames_per_step, 256) if args.frames_per_step == 768 else args.frames_per_step      # PCIe-bound: a default run stays short
    if args.warmup:
        stream_leg(R, wl, devices, max(8, nf // 4))
    res = stream_leg(R, wl, devices, nf * args.steps)
    dt = nf * args.steps / res["fps"]
    line = {"metric": "megapixels/sec (Y-plane)", "value": res["value"], "unit": "MP/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if wl.asm == 5 else "f32", "data": "synthetic",
            "mode": "single process, multi-device ring, host-resident frames (PCIe inclusive) -- NOT the headline metric (that one keeps frames in HBM)",
            "config": {"workload": f"{wl.name}: {wl.desc}", "frames_per_step": nf, "fps": res["fps"], "devices": devices,
                       "parallelism": f"one process, frame i -> device i mod {args.gpus}, 4 lanes per device"},
            "stream": res}
    print(json.dumps(line))
    return 0


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.single_process:
        return single_process_main(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        respawn_ranks(args)                                   # does not return
    wl = Workload(args.config, args.passes or None)
    os.environ.setdefault("RAISR_ORACLE_ISA", "auto")        # CPU legs: widest oracle build this host executes (same bits)
    import torch
    import torch.distributed as dist
    import raisr_hip as R
    import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP extension has no CPU fallback)")
    backend = os.environ.get("RAISR_BENCH_BACKEND", "nccl")           # "nccl" is RCCL on ROCm
    ndev = torch.cuda.device_count()
    if world > ndev and backend == "nccl":
        raise SystemExit(f"bench.py: --gpus {world} needs {world} visible devices (one rank per GPU), found {ndev}")
    # (ranks beyond the visible devices wrap around only for the 2-ranks-on-1-GPU plumbing test, which swaps RCCL for
    #  gloo via RAISR_BENCH_BACKEND -- RCCL refuses two ranks on one device)
    gpu = local_rank % ndev
    numa = bind_to_gpu_numa_node(torch, gpu) if world > 1 and os.environ.get("RAISR_BENCH_NUMA", "1") != "0" else None
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    # launched by torch.distributed.run (RANK set) -> always go through RCCL, even with one rank, so the
    # collective path of the N-GPU runs is the one exercised by a 1-GPU torchrun smoke test
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")

    def load_blobs(w):
        """rank 0 reads the files and packs the device blob; RCCL broadcast to the others"""
        blobs = []
        for p in range(w.passes):
            host_blob = None
            if rank == 0:
                bank, qstr, qcoh, qa = R.read_model_folder(w.folder, w.bits, p + 1)
                host_blob = R.pack_model_blob(bank, qstr, qcoh, qa)
            nbytes = R.lib().raisr_hip_model_blob_bytes(216, w.pixel_types)
            blobs.append(sharding.broadcast_model_blob(host_blob, nbytes, dev, dist if use_dist else None))
        torch.cuda.synchronize()
        return blobs

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    blobs = load_blobs(wl)
    nf = args.frames_per_step
    timing = not args.no_kernel_timing and not args.stream
    kern, iso = {}, {}
    if args.stream:
        # host-resident frames of the (virtual) stream, frame i -> rank i mod world, through the stream ring; the timed
        # region is the submit/collect loop over this rank's frames of all K steps (ring and page-locked planes set up outside)
        mine = sharding.frames_for_rank(nf * args.steps * world, rank, world)
        fence()
        res = stream_leg(R, wl, gpu, len(mine), blobs=blobs)
        dt = len(mine) / res["fps"]
        fence()
        dt_ranks = sharding.gather_over_ranks(dt, dev, dist if use_dist else None)
        dt = sharding.max_over_ranks(dt, dev, dist if use_dist else None)
        frames_total = len(mine) * world
        lanes = []
        e2e_early = None
        host_frames = wl.frames(args.frame_kind, range(4))
    else:
        batch = args.batch or auto_batch(wl)
        nf = max(batch, nf // batch * batch)
        e2e_early = None
        uniq = min(nf, 8)
        # each rank owns its own frames of the (virtual) stream: frame index = rank + world * i
        mine = sharding.frames_for_rank(uniq * world, rank, world)
        host_frames = wl.frames(args.frame_kind, mine)
        # The synchronous host-path leg runs FIRST, in the process state a plugin host has.  After the event-heavy isolated-launch timing
        # below the runtime's copies through the library's bounce memory (pageable planes) ran a third slower in this process (1.31 k vs
        # 1.94 k fps; scripts/e2e_leg_probe.py bisects it), and a leg placed between the headline loop and that timing cooled the GPU's
        # clocks under it (isolated launch 152 -> 160 us).  The W warm-up steps of the headline follow.
        if rank == 0 and world == 1 and not args.no_extras:
            try:
                e2e_early = end_to_end_leg(R, wl, args.extra_frames, gpu)
            except Exception as e:
                e2e_early = {"value": None, "error": f"{type(e).__name__}: {e}"}
        dt, kern, lanes, d_in, d_out = device_loop(R, torch, wl, gpu, blobs, args.lanes, host_frames, nf, args.steps, args.warmup,
                                                   fence, timing, batch=batch)
        dt_ranks = sharding.gather_over_ranks(dt, dev, dist if use_dist else None)
        dt = sharding.max_over_ranks(dt, dev, dist if use_dist else None)
        frames_total = nf * args.steps * world
        if timing and rank == 0:
            iso = isolated_kernel_ms(lanes, d_in, d_out, wl, torch)

    # N > 1: the headline keeps frames resident in HBM (weak scaling of the kernels); beside it, every rank also streams
    # host-resident frames through its pinned ring AT THE SAME TIME, so the scaling record contains PCIe / host-memory contention
    multi_stream = None
    if world > 1 and not args.stream and not args.no_extras and hasattr(R, "RaisrStream"):
        try:
            fence()
            res = stream_leg(R, wl, gpu, args.extra_frames, blobs=blobs)
            dts = sharding.max_over_ranks(args.extra_frames / res["fps"], dev, dist if use_dist else None)
            multi_stream = {"value": round(wl.out_w * wl.out_h * args.extra_frames * world / dts / 1e6, 2), "unit": "MP/s",
                            "fps": round(args.extra_frames * world / dts, 2), "frames_per_rank": args.extra_frames,
                            "what": "host planes -> host planes on every rank concurrently (pinned ring, PCIe inclusive), max over ranks"}
        except Exception as e:  # all ranks take the same path; a failure here must not lose the headline
            multi_stream = {"value": None, "error": f"{type(e).__name__}: {e}"}
            sharding.max_over_ranks(0.0, dev, dist if use_dist else None)

    numa_ranks = sharding.gather_strings(numa or "unbound (single rank)", dist if use_dist else None) if world > 1 else None
    if rank == 0:
        # whole-job rate = all ranks' frames over the SLOWEST rank's time (sharding.job_rate; tests/test_distributed_gloo.py)
        fps_job, fps_ranks = sharding.job_rate([frames_total / world] * world, dt_ranks)
        assert abs(fps_job - frames_total / dt) <= 1e-6 * fps_job
        mp_s = wl.out_w * wl.out_h * fps_job / 1e6
        kernels_ms = {k: round(v["total_ms"] / max(1, v["count"]), 4) for k, v in kern.items()}
        roofline = roofline_of(wl, kern, iso, args.lanes, args.batch or auto_batch(wl))
        fast_level = lanes[0].fast() if lanes and hasattr(lanes[0], "fast") else int(os.environ.get("RAISR_HIP_FAST", "0") or 0)
        for d in lanes:
            d.close()
        # the headline's device planes are not needed by the side legs
        try:
            del d_in, d_out
        except NameError:
            pass
        torch.cuda.empty_cache()
        extras = {}
        if world == 1 and not args.no_extras:
            def leg(name, fn):
                try:
                    extras[name] = fn()
                except Exception as e:  # a reported side figure is never a reason to lose the GPU line
                    extras[name] = {"value": None, "error": f"{type(e).__name__}: {e}"}
            if wl.name == "C2" and not args.passes:
                def c3():
                    w3 = Workload("C3")
                    b3 = load_blobs(w3)
                    n = args.extra_frames
                    dt3, _, l3, _, _ = device_loop(R, torch, w3, gpu, b3, args.lanes, w3.frames(args.frame_kind, range(8)), n, 1, 1,
                                                   fence, False)
                    for d in l3:
                        d.close()
                    return {"value": round(w3.out_w * w3.out_h * n / dt3 / 1e6, 2), "unit": "MP/s", "fps": round(n / dt3, 2),
                            "frames": n, "what": w3.desc + ", CT blend, frames resident in HBM"}
                leg("c3_2pass", c3)
            if e2e_early is not None:
                extras["end_to_end"] = e2e_early
            else:
                leg("end_to_end", lambda: end_to_end_leg(R, wl, args.extra_frames, gpu))
            if hasattr(R, "RaisrStream"):
                leg("stream", lambda: stream_leg(R, wl, gpu, args.extra_frames, blobs=blobs))

                def ring_two_slots():
                    # the ONE-PROCESS multi-device ring (raisr_hip_stream_create_multi, what --single-process times on N GPUs) with this GPU
                    # listed twice: two device slots x 2 lanes = the same four frames in flight as `stream`, through the multi-device code
                    # path (frame i -> slot i mod 2, the model handed from slot 0 to slot 1 inside the library).  Next to `stream` it is the
                    # ring's per-slot cost -- the only part of the multi-GPU host path one GPU can measure.
                    r = stream_leg(R, wl, [gpu, gpu], args.extra_frames, lanes_per_device=2)
                    r["what"] += "; the same physical GPU listed twice (plumbing cost of the multi-device ring, not a scaling figure)"
                    return r
                leg("stream_single_process", ring_two_slots)
            def parity_all():
                par = parity_leg(R, wl, gpu, blobs, args.frame_kind)
                par["certify"] = certify_leg(R, wl, gpu, blobs, host_frames)        # self-check over every frame that was benchmarked
                return par
            leg("parity", parity_all)
            if wl.name == "C2" and not args.passes and not args.no_configs:
                # every BASELINE.json configuration with its own rooflines (C2 = this line's headline; C5 with fewer frames: 8K planes)
                cfgs = {"C2": {"workload": wl.desc, "fps": round(frames_total / dt, 2), "value": round(mp_s, 2), "unit": "MP/s",
                               "kernels_isolated_ms": {k: round(v, 4) for k, v in iso.items()}, "roofline": "see the top-level roofline object",
                               "certify": extras.get("parity", {}).get("certify")}}
                for cname in ("C1", "C2b", "C3", "C4", "C5"):
                    try:
                        cfgs[cname] = config_leg(R, torch, cname, gpu, load_blobs, args.lanes, args.extra_frames if cname != "C5" else max(32, args.extra_frames // 4), fence, args.frame_kind, args.batch)
                    except Exception as e:
                        cfgs[cname] = {"value": None, "error": f"{type(e).__name__}: {e}"}
                if cfgs["C2b"].get("fps"):
                    # context, not a baseline for `value`: other hardware, the reference's own build (-ffast-math), N single-stream processes
                    cfgs["C2b"]["published_reference"] = {
                        "fps_avx512_fp32": 176.2, "fps_avx512_fp16": 222.5, "hardware": "one Xeon 8580+ socket (60 cores), host->host, filter only",
                        "source": "BASELINE.md s1 (docs/images/RAISR_baremetal.png, docs/performance.md:8-14)",
                        "this_gpu_over_published_fp32": round(cfgs["C2b"]["fps"] / 176.2, 2),
                        "note": "frames resident in HBM here; the like-for-like host->host figure is this line's `end_to_end` / `stream` methodology (PCIe inclusive)"}
                extras["configs"] = cfgs
            if wl.name == "C2" and not args.passes and not args.no_configs:
                def kinds_leg():
                    # the certified hash stage makes throughput depend on content (share of pixels that fall back to the exact path):
                    # the headline's frame kind next to the others, worst case (1-px checkerboard: every tile pays both paths) included
                    out = {}
                    for kind in ("natural", "photo", "random", "constant", "checker"):
                        try:
                            fr = wl.frames(kind, range(8 if kind == "photo" else 4))
                        except RuntimeError as e:              # photo: no photographs installed in this image
                            out[kind] = {"fps": None, "error": str(e)}
                            continue
                        n = args.extra_frames
                        dtk, _, lk, _, _ = device_loop(R, torch, wl, gpu, blobs, args.lanes, fr, n, 1, 1, fence, False)
                        for d in lk:
                            d.close()
                        cert = certify_leg(R, wl, gpu, blobs, fr if kind == "photo" else fr[:2])
                        out[kind] = {"fps": round(n / dtk, 2), "value": round(wl.out_w * wl.out_h * n / dtk / 1e6, 2),
                                     "uncertified_frac": cert.get("uncertified_frac"), "certified_wrong": cert.get("certified_wrong"),
                                     "tiles_listed_frac": cert.get("tiles_listed_frac"), "tiles_overflow_frac": cert.get("tiles_overflow_frac")}
                    out["what"] = f"C2, {args.extra_frames} frames per kind, {args.lanes} lanes; `value` of this line is the `{args.frame_kind}` kind on {frames_total} frames"
                    return out
                leg("frame_kinds", kinds_leg)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(wl, args.cpu_sample_frames)
            except Exception as e:  # the baseline is reported data, never a reason to lose the GPU line
                cpu = {"value": None, "unit": "MP/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        where = "host-resident frames streamed through the pinned ring (PCIe inclusive)" if args.stream else "frames resident in HBM"
        line = {
            "metric": "megapixels/sec (Y-plane) 1080p->4K 2x RAISR" if wl.name in ("C2", "C3") else f"megapixels/sec (Y-plane) {wl.name}",
            "value": round(mp_s, 2), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{wl.name}: {wl.desc}, CT blend, {where}" + (f" [passes={wl.passes}]" if args.passes else ""),
                       "frame_kind": args.frame_kind, "mode": "exact" if not fast_level else f"fast-{fast_level} (NOT bit-exact: RAISR_HIP_FAST is set)",
                       "frames_per_step": nf, "lanes": args.lanes, "frames_per_launch": args.batch or auto_batch(wl), "fps": round(frames_total / dt, 2),
                       "timed_region_s": round(dt, 3),
                       "parallelism": f"frame-shard x{world}"} |
                      ({"numa_ranks": numa_ranks, "fps_per_rank": {"min": round(min(fps_ranks), 2), "max": round(max(fps_ranks), 2),
                                                                   "all": [round(f, 2) for f in fps_ranks]}} if world > 1 else {}),
            "kernels_avg_ms": kernels_ms,
            "kernels_isolated_ms": {k: round(v, 4) for k, v in iso.items()},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        line.update(extras)
        if multi_stream is not None:
            line["stream"] = multi_stream
        import ctypes
        ctypes.CDLL(None).fflush(None)                       # nothing buffered by the C runtime may follow the JSON line
        print(json.dumps(line), flush=True)
    else:
        for d in lanes:
            d.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
