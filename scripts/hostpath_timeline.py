"""(build container) One steady-state frame of the synchronous plugin path as a timeline: kernels and PCIe copies from a rocprofv3
--kernel-trace --memory-copy-trace run of scripts/e2e_probe.py (scripts/r03_call9.sh), written to profiles/<tag>_hostpath_timeline.md.
usage: python scripts/hostpath_timeline.py <trace dir> <tag> <title>"""
import csv, glob, os, re, sys

d, tag, title = sys.argv[1], sys.argv[2], sys.argv[3]
ev = []
for fn in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"\b(k_\w+|__amd_rocclr_\w+)", r["Kernel_Name"])
        nm = m.group(1) if m else r["Kernel_Name"][:40]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", nm, f'{r["Grid_Size_X"]}x{r["Grid_Size_Y"]}'))
for fn in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r["Direction"].replace("MEMORY_COPY_", ""), ""))
ev.sort()
# frames = groups that start with the upload of the luma plane: find the k_resize2x launches and cut half-way between frames
res = [i for i, e in enumerate(ev) if e[3] == "HOST_TO_DEVICE" and e[1] - e[0] > 30000]      # the luma upload (2 MB: > 30 us)
assert len(res) > 20, "no frames found"
mid = res[len(res) // 2]
# the frame's first event: the last HOST_TO_DEVICE copy run before this k_resize2x that follows a DEVICE_TO_HOST copy
i0 = mid
while i0 > 0 and not (ev[i0 - 1][3] == "DEVICE_TO_HOST"):
    i0 -= 1
nxt = res[len(res) // 2 + 1]
i1 = nxt
while i1 > 0 and not (ev[i1 - 1][3] == "DEVICE_TO_HOST"):
    i1 -= 1
frame = ev[i0:i1]
t0 = frame[0][0]
period = (ev[res[len(res) // 2 + 8]][0] - ev[res[len(res) // 2 - 8]][0]) / 16 / 1e3
lines = [f"# {title}", "", f"rocprofv3 --kernel-trace --memory-copy-trace of `scripts/e2e_probe.py` (1080p -> 4K yuv420p, page-locked planes); one steady-state frame,",
         f"times in us relative to the frame's first copy; frame period in this run {period:.0f} us ({1e6 / period:.0f} frames/s under the profiler).", "",
         "| start | end | dur | what |", "|---|---|---|---|"]
for a, b, kind, nm, grid in frame:
    lines.append(f"| {(a - t0) / 1e3:.1f} | {(b - t0) / 1e3:.1f} | {(b - a) / 1e3:.1f} | {kind} {nm} {grid} |")
busy = sum(b - a for a, b, k, n, g in frame if k == "copy" and n == "DEVICE_TO_HOST") / 1e3
lines += ["", f"Device-to-host copies of this frame: {busy:.0f} us in total; last event ends at {(max(b for a, b, *_ in frame) - t0) / 1e3:.0f} us."]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_hostpath_timeline.md")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
