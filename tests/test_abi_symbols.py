"""The C-ABI library loads and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import glob
import os
import re

from common import ROOT


def _declared(headers):
    names = set()
    for h in headers:
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(raisr_hip_[a-z_0-9]+|RNLHandler_[A-Za-z]+)\s*\(", txt))
    return names


PRODUCT_HEADERS = [os.path.join(ROOT, "include", "raisr_hip.h"), os.path.join(ROOT, "include", "raisr", "RaisrHandler.h")]
DEBUG_HEADER = os.path.join(ROOT, "include", "raisr_hip_debug.h")


def test_every_declared_symbol_is_exported():
    import raisr_hip as R
    R.build()
    assert sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))) == sorted([PRODUCT_HEADERS[0], DEBUG_HEADER])
    lib = ctypes.CDLL(os.path.join(ROOT, "video-super-resolution-library_amd", "libraisr_hip.so"))
    decl = _declared(PRODUCT_HEADERS)
    assert len(decl) >= 20
    missing = [n for n in sorted(decl) if not hasattr(lib, n)]
    assert not missing, missing


def test_test_hooks_live_in_their_own_library_only():
    """include/raisr_hip_debug.h (introspection for tests, comparison pipelines) is served by libraisr_hip_testhooks.so; the product
    library exports none of it.  (raisr_hip_dev_phase_stats: development builds only.)"""
    pkg = os.path.join(ROOT, "video-super-resolution-library_amd")
    product = ctypes.CDLL(os.path.join(pkg, "libraisr_hip.so"))
    hooks = ctypes.CDLL(os.path.join(pkg, "libraisr_hip_testhooks.so"))
    dbg = _declared([DEBUG_HEADER]) - _declared(PRODUCT_HEADERS)
    assert len(dbg) >= 8
    assert [n for n in sorted(dbg) if hasattr(product, n)] == []
    assert [n for n in sorted(dbg - {"raisr_hip_dev_phase_stats"}) if not hasattr(hooks, n)] == []
    missing = [n for n in sorted(_declared(PRODUCT_HEADERS)) if not hasattr(hooks, n)]      # ... and it is a superset of the product ABI
    assert not missing, missing


def test_pure_host_entry_points_work_without_a_gpu():
    import numpy as np
    import raisr_hip as R
    L = R.lib()
    assert L.raisr_hip_model_blob_bytes(216, 4) == 64 + 216 * 4 * 128 * 4 + 216 * 4 * 64 * 4
    assert L.raisr_hip_model_blob_bytes(0, 4) == 0
    bank = np.arange(216 * 4 * 121, dtype=np.float32).reshape(216, 4, 121)
    bank = (bank * np.float32(1e-3)).astype(np.float32)
    blob = R.pack_model_blob(bank, [0.1, 0.2], [0.3, 0.4], 24)
    nf32 = 216 * 4 * 128 * 4
    body = blob[64:64 + nf32].view(np.float32).reshape(216 * 4, 128)
    assert np.array_equal(body[:, :121], bank.reshape(-1, 121)) and np.all(body[:, 121:] == 0)
    # binary16 bank: [row][chunk][lane] pairs (tap 32c+l, tap 32c+l+16), RNE conversion, zero padding
    h = blob[64 + nf32:].view(np.float16).reshape(216 * 4, 4, 16, 2)
    b16 = np.zeros((216 * 4, 128), np.float16); b16[:, :121] = bank.reshape(-1, 121).astype(np.float16)
    want = np.stack([b16.reshape(-1, 4, 32)[:, :, :16], b16.reshape(-1, 4, 32)[:, :, 16:]], axis=-1)
    assert np.array_equal(h.view(np.uint16), want.view(np.uint16))
    hdr = blob[:64]
    assert hdr[:4].tobytes() == b"RASR" and hdr[4:16].view(np.int32).tolist() == [216, 4, 24]
    assert np.float32(hdr[16:20].view(np.float32)[0]) == np.float32(24) / np.float32(3.141592653)
    assert b"raisr-hip" in L.raisr_hip_version()
    # asynchronous plugin entries: state checks come before anything touches a device
    assert R.RNLHandler_FramesInFlight() == 0
    assert R.RNLHandler_Collect() == R.RNLErrorBadParameter
    assert R.RNLHandler_SetAsyncDepth(17) == R.RNLErrorBadParameter and R.RNLHandler_SetAsyncDepth(4) == 0
    y = np.zeros((8, 8), np.uint8)
    assert R.RNLHandler_Submit((y, y, y), (y, y, y)) == R.RNLErrorBadParameter          # not initialised
    assert R.RNLHandler_Deinit() == 0


def test_headers_are_plain_c_and_link_from_a_c_program(tmp_path):
    """The drop-in headers must be consumable from C (FFmpeg's vf_raisr.c is C): compile a C translation unit against
    include/raisr/RaisrHandler.h + include/raisr_hip.h with gcc, link it to the library and run its GPU-free calls."""
    import subprocess
    import raisr_hip as R
    R.build()
    libdir = os.path.join(ROOT, "video-super-resolution-library_amd")
    src = tmp_path / "c_client.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include <raisr/RaisrHandler.h>
#include <raisr_hip.h>
int main(void) {
    VideoDataType v; memset(&v, 0, sizeof v);
    raisr_hip_band b[4];
    if (sizeof(VideoDataType) != sizeof(void *) + 4 * sizeof(unsigned)) return 2;
    if (HIP != 6 || AVX512_FP16 != 5 || RNLErrorBadParameter != (int)0x80001002) return 3;
    if (raisr_hip_plan_bands(1080, 2160, 1, 4, b) != 4 || b[3].keep_begin + b[3].keep_count != 2160) return 4;
    if (!strstr(raisr_hip_version(), "raisr-hip")) return 5;
    /* a folder without a config file must be refused before any GPU work (Raisr.cpp:1531-1539) */
    if (RNLHandler_Init("/nonexistent-model-folder", 2.0f, 8, VideoRange, 20, AVX512, 1, 1) != RNLErrorBadParameter) return 6;
    if (RNLHandler_Process(&v, &v, &v, &v, &v, &v, CountOfBitsChanged) != RNLErrorBadParameter) return 7;
    /* the entry points added in round 2 reject null handles with an error code (no GPU work before the check) */
    if (raisr_hip_set_fast(NULL, 1) == RAISR_HIP_OK || raisr_hip_use_streams(NULL, NULL, NULL, NULL) == RAISR_HIP_OK) return 8;
    if (raisr_hip_broadcast_model_blob(NULL, 0, NULL, 0, NULL) == RAISR_HIP_OK || raisr_hip_stream_set_fast(NULL, 1) == RAISR_HIP_OK) return 9;
    if (raisr_hip_stream_depth(NULL) != 0 || raisr_hip_get_fast(NULL) != 0) return 10;
    puts("c-client-ok");
    return 0;
}
''')
    exe = tmp_path / "c_client"
    cc = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", f"-I{os.path.join(ROOT, 'include')}", str(src), "-o", str(exe),
          f"-L{libdir}", "-lraisr_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    out = subprocess.run(cc, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "c-client-ok" in run.stdout, (run.returncode, run.stdout[-500:], run.stderr[-500:])


def test_cpp_flavour_compiles_with_default_arguments(tmp_path):
    """include/raisr/Raisr.h from C++ with the reference's default arguments (Library/Raisr.h:14-33)."""
    import subprocess
    import raisr_hip as R
    R.build()
    libdir = os.path.join(ROOT, "video-super-resolution-library_amd")
    src = tmp_path / "cpp_client.cpp"
    src.write_text(r'''
#include <iostream>
#include <raisr/Raisr.h>
int main() {
    std::string folder = "/nonexistent-model-folder";
    if (RNLInit(folder, 2.0f) != RNLErrorBadParameter) return 2;           // bits=8, VideoRange, 20 threads, AVX512, 1 pass
    VideoDataType v{};
    if (RNLProcess(&v, &v, &v, &v, &v, &v) != RNLErrorBadParameter) return 3;   // blending defaults to CountOfBitsChanged
    if (RNLDeinit() != RNLErrorNone) return 4;
    std::cout << "cpp-client-ok\n";
    return 0;
}
''')
    exe = tmp_path / "cpp_client"
    cc = ["g++", "-std=c++17", "-Wall", "-Werror", f"-I{os.path.join(ROOT, 'include')}", str(src), "-o", str(exe),
          f"-L{libdir}", "-lraisr_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    out = subprocess.run(cc, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "cpp-client-ok" in run.stdout, (run.returncode, run.stdout[-500:], run.stderr[-500:])


def test_static_flavour_links_from_plain_c(tmp_path):
    """libraisr.a (the reference ships a static library, Library/CMakeLists.txt:24,44-45): a C99 program links against it with
    -lstdc++ -lamdhip64 and runs the host-only entry points (no device needed for these)."""
    import shutil
    import subprocess
    pkg = os.path.join(ROOT, "video-super-resolution-library_amd")
    rocm_lib = "/opt/rocm/lib"
    if not (shutil.which("ar") and os.path.exists(os.path.join(rocm_lib, "libamdhip64.so"))):
        import pytest
        pytest.skip("needs ar and the HIP runtime to link against")
    subprocess.check_call(["make", "-s", "-C", pkg, "libraisr.a"])
    src = tmp_path / "main.c"
    src.write_text('#include <stdio.h>\n#include "raisr/RaisrHandler.h"\n#include "raisr_hip.h"\n'
                   'int main(void) { if (RNLHandler_Collect() != RNLErrorBadParameter) return 2;\n'
                   '  printf("%s %d\\n", raisr_hip_version(), (int)RNLHandler_Deinit()); return 0; }\n')
    exe = tmp_path / "main"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                           os.path.join(pkg, "libraisr.a"), "-L", rocm_lib, "-lamdhip64", "-lstdc++", "-ldl", "-lpthread", "-lm", "-o", str(exe)])
    out = subprocess.run([str(exe)], env=dict(os.environ, LD_LIBRARY_PATH=rocm_lib), capture_output=True, text=True)
    assert out.returncode == 0 and "raisr-hip" in out.stdout and out.stdout.strip().endswith(" 0"), (out.stdout, out.stderr)
