"""CPU: the row-copy threads of the host path (csrc/host_copy.h: pageable planes are packed into / unpacked from page-locked
bounce memory by a few persistent threads).  Compiled on its own with g++ (the HIP API header only declares what HostBounce's
inline functions would call; nothing of it is linked) and run with 1, 2 and 4 threads."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h") or shutil.which("g++") is None, reason="needs g++ and the HIP API header")
def test_row_copy_pool_copies_exactly_the_rows(tmp_path):
    exe = str(tmp_path / "row_copy_pool_check")
    cc = subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                         "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "tools", "row_copy_pool_check.cpp")],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    for n in ("1", "2", "4"):
        run = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, RAISR_HIP_COPY_THREADS=n), timeout=120)
        assert run.returncode == 0 and "ok" in run.stdout, (n, run.stdout, run.stderr[-500:])


@pytest.mark.skipif(not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h") or shutil.which("g++") is None, reason="needs g++ and the HIP API header")
def test_numa_placement_of_the_copy_pools_on_a_fake_two_node_host(tmp_path):
    """raisr_numa (csrc/host_copy.h): a device's PCI function -> NUMA node -> that node's CPUs -> a row-copy pool whose threads are bound
    there.  The build container has one node, so the test hands the code a fake /sys with two nodes made of this process's own CPUs."""
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 2:
        pytest.skip("needs two CPUs")
    half = len(cpus) // 2
    root = tmp_path / "sys"
    for node, part in ((0, cpus[:half]), (1, cpus[half:])):
        d = root / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in part) + "\n")
    for bdf, node in (("0000:c1:00.0", 1), ("0000:01:00.0", 0), ("0000:ff:00.0", -1)):
        d = root / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    exe = str(tmp_path / "row_copy_pool_check")
    cc = subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                         "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "tools", "row_copy_pool_check.cpp")],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, RAISR_HIP_SYSFS_ROOT=str(root), RAISR_HIP_COPY_THREADS="3"), timeout=120)
    assert run.returncode == 0 and "ok" in run.stdout, (run.stdout, run.stderr[-500:])
