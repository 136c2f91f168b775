"""The C ABI's pose all-gather on a real device: a one-rank RCCL communicator (a one-GPU box cannot host more ranks --
RCCL refuses two ranks on one device; the world-size-2 logic runs in tests/test_distributed_cpu.py on the file transport)."""
import numpy as np
import pytest

from spvo import capi

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_communicator_gathers_its_own_poses():
    uid = capi.comm_unique_id()
    assert len(uid) == capi.COMM_ID_BYTES and any(uid)
    comm = capi.Comm.rccl(0, 0, 1, uid)
    assert (comm.rank, comm.world) == (0, 1)
    rng = np.random.RandomState(0)
    for n in (1, 64, 300):                                   # 300 > the initial buffer: the communicator grows
        poses = rng.randn(n, 7)
        out = comm.allgather(poses)
        assert out.shape == (1, n, 7) and np.array_equal(out[0], poses)
    comm.close()


def test_posegather_uses_the_c_abi_on_a_gpu():
    import os
    import torch
    import torch.distributed as dist
    from spvo import posegather
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29541", "RANK": "0", "WORLD_SIZE": "1"})
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        pg = posegather.PoseGather(torch.device("cuda", 0), force=True)
        assert pg.transport == "c:rccl", pg.transport_note
        for k in range(130):
            pg.gather_async([0, 0, 0, 1], [k, 2, 3])
        seq = pg.collect()
        assert seq.shape == (130, 1, 7) and np.allclose(seq[:, 0, 4], np.arange(130))
        pg.close()
    finally:
        dist.destroy_process_group()
