"""The C-ABI library loads and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import glob
import os
import re

from common import ROOT


def _declared():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "include", "raisr", "RaisrHandler.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(raisr_hip_[a-z_0-9]+|RNLHandler_[A-Za-z]+)\s*\(", txt))
    return names


def test_every_declared_symbol_is_exported():
    import raisr_hip as R
    R.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "video-super-resolution-library_amd", "libraisr_hip.so"))
    decl = _declared()
    assert len(decl) >= 20
    missing = [n for n in sorted(decl) if not hasattr(lib, n)]
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
