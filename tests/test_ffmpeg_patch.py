"""CPU: the FFmpeg deliverables (ffmpeg/).  vf_raisr_hip.diff must apply to the reference's vf_raisr.c (only available in
the build container: skipped elsewhere), and the patched filter must pass `gcc -fsyntax-only` against the library's own
headers plus a declaration-only lint subset of the libavfilter API (tests/ffmpeg_stub/)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/ffmpeg/vf_raisr.c"


def test_build_system_patch_mentions_every_touch_point():
    s = open(os.path.join(ROOT, "ffmpeg", "0001-ffmpeg-raisr-hip-filter.patch")).read()
    for needle in ("--enable-libraisr", "raisr_filter_deps=\"libraisr\"", "-lraisr_hip -lamdhip64 -lstdc++", "RNLHandler_Init",
                   "OBJS-$(CONFIG_RAISR_FILTER)", "extern const AVFilter ff_vf_raisr;",
                   # the hardware-frames filter (ffmpeg/vf_raisr_hipframes.c), as the reference patch wires raisr_opencl
                   'raisr_hip_filter_deps="libraisr vaapi libdrm"', "OBJS-$(CONFIG_RAISR_HIP_FILTER)              += vf_raisr_hipframes.o",
                   "extern const AVFilter ff_vf_raisr_hip;"):
        assert needle in s, needle
    assert "libipp" not in s                        # the CPU library's IPP dependency is gone


@pytest.mark.skipif(not os.path.exists(REF) or shutil.which("patch") is None, reason="needs the reference tree and patch(1)")
def test_filter_diff_applies_and_the_result_compiles_against_our_headers(tmp_path):
    work = tmp_path / "libavfilter"
    work.mkdir()
    shutil.copy(REF, work / "vf_raisr.c")
    diff = os.path.join(ROOT, "ffmpeg", "vf_raisr_hip.diff")
    dry = subprocess.run(["patch", "--dry-run", "-p1", "-d", str(work), "-i", diff], capture_output=True, text=True)
    assert dry.returncode == 0, dry.stdout + dry.stderr
    real = subprocess.run(["patch", "-p1", "-d", str(work), "-i", diff], capture_output=True, text=True)
    assert real.returncode == 0, real.stdout + real.stderr
    src = (work / "vf_raisr.c").read_text()
    assert 'asm_t = HIP;' in src and '{.str = "hip"}' in src
    assert "if (asm_t == OpenCL)" not in src        # the device hook now runs for every asm value
    # async=N: Submit / Collect ring with both frames kept alive, drained at EOF and in uninit
    for needle in ('{"async",', "RNLHandler_SetAsyncDepth(raisr->async)", "RNLHandler_Submit(", "RNLHandler_Collect()",
                   ".request_frame = request_frame", "ret == AVERROR_EOF && raisr->q_count > 0", "collect_oldest(ctx, 0)"):
        assert needle in src, needle
    # devices=0,1,..|all: the ring spans several GPUs, the filter keeps async frames in flight on each
    for needle in ('{"devices",', "RNLHandler_SetDeviceList(raisr->devices)", "RNLHandler_AsyncCapacity()", "raisr->q_count == raisr->q_cap",
                   "raisr->async > 0 && raisr->q_cap < 1", "if (raisr->q_count <= 0)"):   # (round 6: no capacity is an init error; an empty queue is never collected from)
        assert needle in src, needle
    # pinned=1: input and output frames from buffer pools over the library's page-locked allocator (the buffer owns the memory:
    # nothing is page-locked behind FFmpeg's back, nothing outlives its buffer)
    for needle in ('{"pinned",', "RNLHandler_HostAlloc(size)", "RNLHandler_HostFree(data)", "av_buffer_pool_init2(size + 4 * 64 /* tail padding",
                   ".get_buffer.video = get_video_buffer_input", "av_buffer_pool_uninit(&raisr->pool_out)"):
        assert needle in src, needle
    cc = shutil.which("gcc") or shutil.which("cc")
    out = subprocess.run([cc, "-std=gnu11", "-fsyntax-only", "-Wall", "-Wno-unused-variable", "-Wno-declaration-after-statement",
                          "-I", os.path.join(ROOT, "tests", "ffmpeg_stub"), "-I", os.path.join(ROOT, "include"),
                          str(work / "vf_raisr.c")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]


def test_hardware_frames_filter_compiles_against_our_headers():
    """ffmpeg/vf_raisr_hipframes.c (the counterpart of the reference's vf_raisr_opencl.c: VAAPI surfaces -> DRM PRIME dma-bufs ->
    HIP device pointers -> ASMType HIPExternal): `gcc -fsyntax-only` against the library's headers, the real HIP runtime API header
    and the declaration-only libavfilter / hwcontext subset; the calls it makes exist with these signatures."""
    hip_inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(hip_inc, "hip", "hip_runtime_api.h")):
        pytest.skip("needs the HIP runtime headers")
    cc = shutil.which("gcc") or shutil.which("cc")
    src = os.path.join(ROOT, "ffmpeg", "vf_raisr_hipframes.c")
    out = subprocess.run([cc, "-std=gnu11", "-fsyntax-only", "-Wall", "-Wno-unused-variable", "-D__HIP_PLATFORM_AMD__",
                          "-I", os.path.join(ROOT, "tests", "ffmpeg_stub"), "-I", os.path.join(ROOT, "include"), "-I", hip_inc, src],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    text = open(src).read()
    for needle in ("hipImportExternalMemory", "hipExternalMemoryGetMappedBuffer", "HIPExternal", "RAISR_HIP_INTERLEAVED2",
                   "RNLHandler_SetOpenCLContext(ctx->stream, NULL, 0, ctx->device)", "DRM_FORMAT_MOD_LINEAR_", "FF_FILTER_FLAG_HWFRAME_AWARE"):
        assert needle in text, needle


def test_build_system_patch_hunks_are_well_formed():
    """Every hunk of the n6.0 build-system patch has the line counts its header announces, and the new-file offsets of a file's
    hunks grow by exactly the lines the earlier hunks added (the patch cannot be applied here: no FFmpeg tree in the image)."""
    import re
    s = open(os.path.join(ROOT, "ffmpeg", "0001-ffmpeg-raisr-hip-filter.patch")).read()
    body = s[s.index("diff --git"):]
    added_total = 0
    for filediff in body.split("diff --git")[1:]:
        lines = filediff.split("\n")
        shift = 0
        i = 0
        while i < len(lines):
            m = re.match(r"@@ -(\d+),(\d+) \+(\d+),(\d+) @@", lines[i])
            if not m:
                i += 1
                continue
            a, b, c, d = map(int, m.groups())
            assert c == a + shift, (lines[i], shift)
            i += 1
            ctx = add = rem = 0
            while i < len(lines) and not lines[i].startswith(("@@", "diff --git")) and (ctx + rem < b or ctx + add < d):
                if lines[i].startswith("+"):
                    add += 1
                elif lines[i].startswith("-"):
                    rem += 1
                else:
                    ctx += 1
                i += 1
            assert ctx + rem == b and ctx + add == d, (m.group(0), ctx, add, rem)
            shift += add - rem
            added_total += add
    stat = re.search(r"(\d+) files changed, (\d+) insertions", s)
    assert stat and int(stat.group(2)) == added_total, (stat.group(0) if stat else None, added_total)
