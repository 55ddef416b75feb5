"""-m gpu: raisr_hip_broadcast_model_blob -- the path's only collective, from the C ABI with a real RCCL communicator.
A 1-GPU box can only form a one-rank communicator, so what is checked is the binding itself (librccl resolved at run
time, ncclBroadcast called with the right datatype/count on the caller's stream, error reporting) and that the blob that went
through it drives a bit-exact frame; the world-size-2 sharding logic is covered on CPU (tests/test_distributed_gloo.py)."""
import ctypes
import os

import numpy as np
import pytest

from common import folder, oracle_y

pytestmark = pytest.mark.gpu
CASE = ("2x_8b_avx512", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, False)


class NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def _rccl():
    L = ctypes.CDLL("librccl.so.1")
    L.ncclGetUniqueId.argtypes = [ctypes.POINTER(NcclUniqueId)]
    L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, NcclUniqueId, ctypes.c_int]
    L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    return L


def test_model_blob_broadcast_through_rccl_then_bit_exact_frame():
    import torch
    import raisr_hip as R
    import synth
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    bank, qstr, qcoh, qa = R.read_model_folder(folder(CASE[1]), 8, 1)
    host = R.pack_model_blob(bank, qstr, qcoh, qa)
    blob = torch.from_numpy(host.copy()).cuda()
    before = blob.clone()

    nccl = _rccl()
    uid = NcclUniqueId()
    assert nccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    assert nccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        stream = torch.cuda.Stream()
        rc = R.lib().raisr_hip_broadcast_model_blob(comm, 0, blob.data_ptr(), blob.numel(), stream.cuda_stream)
        assert rc == 0, R.last_error()
        stream.synchronize()
        assert torch.equal(blob, before)
        # argument checks come back as error codes with a message, not as crashes
        assert R.lib().raisr_hip_broadcast_model_blob(None, 0, blob.data_ptr(), blob.numel(), None) != 0
        assert "bad argument" in R.last_error()
        assert R.lib().raisr_hip_broadcast_model_blob(comm, 0, blob.data_ptr(), 8, None) != 0
    finally:
        nccl.ncclCommDestroy(comm)

    y = synth.natural_y(200, 120, 8, seed=3)
    dev = R.RaisrDevice(0)
    try:
        dev.set_model_blob_device(0, blob.data_ptr(), blob.numel())
        dev.configure(200, 120, 400, 240, bits=8, passes=1, hash_variant=2)
        out = np.zeros((240, 400), np.uint8)
        dev.process_host(y, out)
    finally:
        dev.close()
    assert np.array_equal(out, oracle_y(y, CASE))
