"""world_size-2 tests of the multi-GPU layer (one process per GPU, pose all-gather): through the C ABI's communicator
(spvo_comm_create_host: the file transport that stands in for RCCL where there is no GPU) and through torch.distributed
(gloo)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spvo import posegather, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, transport):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pg = posegather.PoseGather(transport=transport)
    assert pg.transport == ("c:host" if transport == "c" else "torch")
    # every rank owns a different stream
    poses = synth.ego_motion(3, seed=posegather.stream_seed(rank))
    R, t = synth.relative_pose(poses[0], poses[1])
    first = pg.gather(None, None)                                   # frame 0: no pose yet
    mine = np.array([0.0, 0.01 * (rank + 1), 0.0, 1.0])
    allp = pg.gather(mine, t)
    # non-blocking form (what bench.py uses): three steps enqueued, read back once, in order
    pg.gather_async(None, None)
    pg.gather_async(mine, t)
    pg.gather_async(mine, 2 * t)
    seq = pg.collect()
    assert seq.shape == (3, world, 7) and np.allclose(seq[0], posegather.IDENTITY_POSE) and np.allclose(seq[1], allp)
    assert np.allclose(seq[2][rank, 4:], 2 * t) and pg.collect().shape == (0, world, 7)
    # more steps than one batch (64) and more batches than staging slots (4): 5 full batches + a ragged one, in order
    n = 5 * posegather.PoseGather.BATCH + 17
    for k in range(n):
        pg.gather_async(mine, (k + 1) * t)
    seq = pg.collect()
    assert seq.shape == (n, world, 7)
    for k in (0, 63, 64, 200, n - 1):
        assert np.allclose(seq[k][rank, 4:], (k + 1) * t) and np.allclose(seq[k][rank, :4], mine)
        assert np.allclose(seq[k][1 - rank, :4], [0.0, 0.01 * (2 - rank), 0.0, 1.0])
    q.put((rank, first, allp, t))
    dist.barrier()
    pg.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["c", "torch"])
def test_pose_allgather_world2(transport):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ts = [r[3] for r in res]
    assert not np.allclose(ts[0], ts[1])                            # different streams per rank
    for rank, first, allp, _ in res:
        assert first.shape == (2, 7) and np.allclose(first, posegather.IDENTITY_POSE)
        assert allp.shape == (2, 7)
        for r in range(world):                                      # every rank sees every pose, in rank order
            assert np.allclose(allp[r, :4], [0.0, 0.01 * (r + 1), 0.0, 1.0]) and np.allclose(allp[r, 4:], ts[r])


def test_single_process_gather_is_identity_passthrough():
    pg = posegather.PoseGather()
    out = pg.gather([0, 0, 0, 1], [1, 2, 3])
    assert out.shape == (1, 7) and np.allclose(out[0], [0, 0, 0, 1, 1, 2, 3])
    pg.gather_async([0, 0, 0, 1], [1, 2, 3])
    assert np.allclose(pg.collect(), out[None])
    for k in range(130):
        pg.gather_async([0, 0, 0, 1], [k, 2, 3])
    seq = pg.collect()
    assert seq.shape == (130, 1, 7) and np.allclose(seq[:, 0, 4], np.arange(130))


def test_c_abi_comm_rejects_bad_arguments_and_needs_a_gpu_for_rccl(tmp_path):
    """spvo_comm_*: argument checks, the loud failure of the RCCL transport without a device, and a one-rank gather
    through the file transport (what a C++ host calls: include/spvo.h)."""
    import ctypes as C
    from spvo import capi
    lib = capi.load()
    h = C.c_void_p()
    assert lib.spvo_comm_create_host(str(tmp_path).encode(), 2, 2, C.byref(h)) == -1          # rank >= world
    assert lib.spvo_comm_create_host(b"/nonexistent/dir", 0, 1, C.byref(h)) == -3
    if not torch.cuda.is_available():
        with pytest.raises(capi.SpvoError) as e:
            capi.Comm.rccl(0, 0, 1, bytes(128))
        assert "no CPU path" in str(e.value) or "librccl" in str(e.value)
    c = capi.Comm.host(str(tmp_path), 0, 1)
    poses = np.arange(21, dtype=np.float64).reshape(3, 7)
    assert (c.rank, c.world) == (0, 1) and np.array_equal(c.allgather(poses), poses[None])
    assert np.array_equal(c.allgather(poses[0]), poses[None, :1])
    c.close()
    assert not any(f.startswith("pose_") for f in os.listdir(tmp_path))                        # files cleaned up


# ---- bench.py's own launcher (`python bench.py --gpus N` without torch.distributed.run): a rank that dies must not leave its
# ---- siblings waiting in a collective until somebody's timeout (BASELINE config 4: one try on the 8-GPU node)
_RANK_SCRIPT = '''
import os, sys, time
import torch.distributed as dist
rank = int(os.environ["RANK"])
open(os.path.join(sys.argv[1], f"pid{rank}"), "w").write(str(os.getpid()))
dist.init_process_group("gloo")
dist.barrier()
sys.stderr.write(f"rank {rank}: past the first barrier\\n")
sys.stderr.flush()
if sys.argv[2] == "die" and rank == 1:
    sys.stderr.write("rank 1: simulated failure (out of memory)\\n")
    sys.stderr.flush()
    os._exit(3)
if sys.argv[2] == "die":
    time.sleep(600)          # rank 0: stuck, as in a collective whose peer is gone
dist.barrier()
dist.destroy_process_group()
'''


def _pid_alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    try:   # a zombie of ours would still answer: it must have been reaped
        return open(f"/proc/{pid}/stat").read().split()[2] != "Z"
    except OSError:
        return False


def test_launcher_stops_the_job_when_a_rank_dies(tmp_path, capfd):
    import sys
    import time
    import bench
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    env_keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
    saved = {k: os.environ.pop(k) for k in env_keys if k in os.environ}
    try:
        t0 = time.time()
        rc = bench.spawn_ranks(2, cmd=[sys.executable, str(script), str(tmp_path), "die"], grace_s=2.0)
        took = time.time() - t0
    finally:
        os.environ.update(saved)
    err = capfd.readouterr().err
    assert rc == 3 and took < 30.0, (rc, took, err)
    assert "rank 1 of 2 exited with code 3" in err and "simulated failure" in err
    for r in (0, 1):   # nobody is left behind
        assert not _pid_alive(int((tmp_path / f"pid{r}").read_text()))


def test_launcher_returns_zero_and_forwards_rank0_diagnostics(tmp_path, capfd):
    import sys
    import bench
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    rc = bench.spawn_ranks(2, cmd=[sys.executable, str(script), str(tmp_path), "ok"])
    err = capfd.readouterr().err
    assert rc == 0 and "rank 0: past the first barrier" in err and "rank 1" not in err


def test_launcher_deadline_terminates_a_hung_job(tmp_path, capfd):
    import sys
    import time
    import bench
    script = tmp_path / "hang.py"
    script.write_text("import os, sys, time\nopen(os.path.join(sys.argv[1], 'pid' + os.environ['RANK']), 'w').write(str(os.getpid()))\ntime.sleep(600)\n")
    t0 = time.time()
    rc = bench.spawn_ranks(2, cmd=[sys.executable, str(script), str(tmp_path)], deadline_s=3.0, grace_s=2.0)
    assert rc == 124 and time.time() - t0 < 20.0
    assert "did not finish within 3 s" in capfd.readouterr().err
    for r in (0, 1):
        assert not _pid_alive(int((tmp_path / f"pid{r}").read_text()))


def test_host_transport_directory_can_serve_a_second_communicator(tmp_path):
    """spvo_comm_create_host in a directory an earlier communicator has used: the done_<rank> markers of the first one must not
    satisfy the destroy handshake of the second (a rank would remove files a peer is still reading).  Two ranks as threads
    (ctypes releases the GIL), two communicators one after the other in the same directory, three gathers each."""
    import threading
    from spvo import capi
    assert isinstance(capi.comm_available(), bool)          # dlopen probe: no GPU and no bootstrap state needed
    errors = []

    def rank_main(rank, generation):
        try:
            comm = capi.Comm.host(str(tmp_path), rank, 2)
            for k in range(3):
                mine = np.array([[0.0, 0.0, 0.0, 1.0, generation, rank, k]])
                got = comm.allgather(mine)
                assert got.shape == (2, 1, 7) or got.shape == (2, 7), got.shape
                got = got.reshape(2, 7)
                assert np.array_equal(got[:, 4], [generation, generation]) and np.array_equal(got[:, 5], [0, 1]) and np.array_equal(got[:, 6], [k, k])
            comm.close()
        except Exception as exc:      # noqa: BLE001
            errors.append((rank, generation, repr(exc)))

    for generation in (1, 2):
        ts = [threading.Thread(target=rank_main, args=(r, generation)) for r in (0, 1)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=60)
        assert not errors and not any(t.is_alive() for t in ts), errors
