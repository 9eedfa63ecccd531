"""The library's ghost exchange (csrc/comm.hip, include/allegro_hip.h `ahip_comm_*`): what the reference gets from LAMMPS' forward /
reverse communication (/root/reference/pair_nequip_allegro.cpp:149,366-368).  CPU: the pack / unpack kernels through the host-emulation
build with a loop-back transport; GPU: a ONE-rank RCCL communicator on the box, so that every RCCL entry point the multi-GPU bench uses has
executed before the driver's 8-GPU run."""
import ctypes

import numpy as np
import pytest
import torch

from pair_allegro_amd import capi


def _loopback(ops):
    """one rank: a send to myself is matched with my receive of the same group, in order"""
    sends = [(p, n) for k, peer, p, n in ops if k == 0]
    recvs = [(p, n) for k, peer, p, n in ops if k == 1]
    assert len(sends) == len(recvs)
    for (sp, sn), (rp, rn) in zip(sends, recvs):
        assert sn == rn
        ctypes.memmove(rp, sp, sn)


def test_hosted_selftest_and_plans_cpu(emu_lib):
    c = capi.Comm(emu_lib, 0, 1, xfer=_loopback)
    c.selftest(257)
    # a 1-D periodic line of 6 atoms, one rank, ghosts = images of the two end atoms; as a swap plan (two self-swaps of dimension 0) ...
    x = torch.tensor([[0.5, 0, 0], [1.5, 0, 0], [2.5, 0, 0], [3.5, 0, 0], [4.5, 0, 0], [5.5, 0, 0], [9, 9, 9], [9, 9, 9]], dtype=torch.float64)
    lo_idx = torch.tensor([0], dtype=torch.int64)            # sent "down": re-appears above the box
    hi_idx = torch.tensor([5], dtype=torch.int64)
    c.set_plan([0, 0], [0, 0], [0, 0], [6.0, -6.0], [1, 1], [1, 1], [6, 7], [lo_idx.data_ptr(), hi_idx.data_ptr()])
    c.forward(x.data_ptr())
    np.testing.assert_allclose(x[6:].numpy(), [[6.5, 0, 0], [-0.5, 0, 0]])
    f = torch.zeros((8, 3), dtype=torch.float64)
    f[6] = torch.tensor([1.0, 2.0, 3.0]); f[7] = torch.tensor([10.0, 20.0, 30.0]); f[0, 0] = 0.25
    c.reverse(f.data_ptr())
    np.testing.assert_allclose(f[0].numpy(), [1.25, 2.0, 3.0])
    np.testing.assert_allclose(f[5].numpy(), [10.0, 20.0, 30.0])
    # ... and as the resolved single-rank plan
    src = torch.tensor([0, 5], dtype=torch.int64)
    sh = torch.tensor([[6.0, 0, 0], [-6.0, 0, 0]], dtype=torch.float64)
    x[6:] = 9.0
    c.set_plan_local(6, 2, src.data_ptr(), sh.data_ptr())
    c.forward(x.data_ptr())
    np.testing.assert_allclose(x[6:].numpy(), [[6.5, 0, 0], [-0.5, 0, 0]])
    f = torch.zeros((8, 3), dtype=torch.float64); f[6, 1] = 2.0; f[7, 2] = -1.0
    c.reverse(f.data_ptr())
    assert f[0, 1] == 2.0 and f[5, 2] == -1.0
    c.close()


def test_plan_validation(emu_lib):
    c = capi.Comm(emu_lib, 0, 1)
    idx = torch.zeros(1, dtype=torch.int64)
    with pytest.raises(capi.AhipError):                       # swaps come in pairs
        c.set_plan([0], [0], [0], [0.0], [1], [1], [1], [idx.data_ptr()])
    with pytest.raises(capi.AhipError):                       # a self-swap receives what it sends
        c.set_plan([0, 0], [0, 0], [0, 0], [0.0, 0.0], [1, 1], [2, 1], [1, 2], [idx.data_ptr(), idx.data_ptr()])
    with pytest.raises(capi.AhipError):                       # a remote rank that does not exist
        c.set_plan([0, 0], [1, 0], [0, 0], [0.0, 0.0], [1, 1], [1, 1], [1, 2], [idx.data_ptr(), idx.data_ptr()])
    c.close()


@pytest.mark.gpu
def test_rccl_one_rank_communicator(hip_lib):
    """ncclGetUniqueId / ncclCommInitRank / grouped ncclSend + ncclRecv (to itself) / ncclAllReduce (sum, max) / ncclCommDestroy."""
    uid = capi.Comm.unique_id(hip_lib)
    assert len(uid) == 128
    c = capi.Comm(hip_lib, 0, 1, rccl_id=uid, device=0)
    assert c.transport == "rccl"
    c.selftest(4096)
    c.selftest(3)
    # the exchange kernels on device memory (self-swaps need no transport)
    x = torch.tensor([[0.5, 0, 0], [5.5, 0, 0], [9, 9, 9], [9, 9, 9]], dtype=torch.float64, device="cuda")
    i0 = torch.tensor([0], dtype=torch.int64, device="cuda"); i1 = torch.tensor([1], dtype=torch.int64, device="cuda")
    c.set_plan([0, 0], [0, 0], [0, 0], [6.0, -6.0], [1, 1], [1, 1], [2, 3], [i0.data_ptr(), i1.data_ptr()])
    c.forward(x.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_allclose(x[2:].cpu().numpy(), [[6.5, 0, 0], [-0.5, 0, 0]])
    c.close()
