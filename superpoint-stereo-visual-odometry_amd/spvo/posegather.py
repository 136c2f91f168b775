"""Multi-GPU layer of the path: N independent stereo streams, one rank per GPU, and the ONE
collective the path needs -- an all-gather of each rank's per-frame relative pose
(7 doubles: quaternion x, y, z, w + translation; 56 bytes per rank).

The collective itself is the C ABI's (include/spvo.h: spvo_comm_create / spvo_pose_allgather_n, RCCL over xGMI on a
stream of the communicator's own) -- the same entry points a C++ ROS host calls; this module is the thin Python caller
bench.py and the tests use.  torch.distributed only bootstraps it (hands rank 0's RCCL id to the other ranks) and
stays available as a second transport (`transport="torch"`: backend "nccl" = RCCL, "gloo" on CPU).  The message is
latency-bound (no bandwidth term): poses are staged on the host and leave in batches of BATCH frames, one collective
per batch (a per-frame collective issued from the framework's default stream cost 26 % of the frame rate on one GPU,
DESIGN.md section 6).  Streams never exchange image or feature data.
"""
from __future__ import annotations

import os
import tempfile
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

IDENTITY_POSE = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)


def stream_seed(rank: int, base_seed: int = 0) -> int:
    """Each rank renders / reads its own stereo stream."""
    return base_seed + rank


def _pose(q_xyzw, t) -> np.ndarray:
    return IDENTITY_POSE if q_xyzw is None else np.concatenate([np.asarray(q_xyzw, np.float64), np.asarray(t, np.float64)])


class PoseGather:
    BATCH = 64    # frames per collective of the non-blocking form
    SLOTS = 4     # torch transport: batches that may be in flight

    def __init__(self, device: Optional[torch.device] = None, force: bool = False, transport: str = "auto"):
        """transport: "c" = the C ABI's communicator (RCCL on a GPU rank, the file transport of the CPU tests
        otherwise), "torch" = torch.distributed collectives, "auto" = "c" with a fall-back to "torch" when the
        communicator cannot be created.  force: run the collectives in a one-rank process group as well (test hook)."""
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.local_only = self.world == 1 and not (force and dist.is_initialized())
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.device = device or torch.device("cpu")
        self.comm = None
        self.transport = "local" if self.local_only else transport
        self.transport_note = ""
        self.init_ms = 0.0                 # wall time of creating the communicator (ncclCommInitRank on every rank + the bootstrap exchange): once, outside any timed region
        self.collective_ms: List[float] = []   # host wall time of every batched collective since reset_timing(): what a frame loop pays per BATCH frames
        import time as _time
        self._clock = _time.perf_counter
        t_init = self._clock()
        if self.transport in ("auto", "c"):
            # Every rank takes the same decision: _make_comm agrees on it collectively (all ranks reach every broadcast /
            # all-reduce whatever happened locally), so no rank is left waiting in a collective its peers have abandoned.
            self.comm, why = self._make_comm()
            if self.comm is not None:
                self.transport = "c:rccl" if self.device.type == "cuda" else "c:host"
            elif transport == "c":
                raise RuntimeError("C-ABI communicator unavailable: " + why)
            else:
                self.transport, self.transport_note = "torch", f"C-ABI communicator unavailable ({why}); torch.distributed used"
        self.init_ms = 1e3 * (self._clock() - t_init)
        self.buf = torch.zeros(7, dtype=torch.float64, device=self.device)
        self.out: List[torch.Tensor] = [torch.zeros(7, dtype=torch.float64, device=self.device) for _ in range(self.world)]
        self._pending: list = []
        self._host = None
        self._fill = 0
        self._stage = self._send = self._ready = None
        self._side = None

    def _all_ok(self, ok: bool) -> bool:
        """True iff `ok` on EVERY rank (one MIN all-reduce; every rank calls it the same number of times)."""
        if self.world == 1 or not dist.is_initialized():
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def _make_comm(self):
        """(communicator, "") on every rank, or (None, reason) on every rank.  Protocol: (1) each rank checks locally that it can
        take part (GPU ranks: librccl loads and hands out an id -- rank 0's is the one that is used; host ranks: nothing to check)
        and the ranks agree on that; (2) rank 0 broadcasts its id / directory -- always, None included; (3) each rank creates its
        communicator and the ranks agree on the outcome, closing what was created when a peer failed.  (A rank that fails INSIDE
        ncclCommInitRank leaves its peers in that call: not recoverable from here -- the launcher stops the job when a rank dies.)"""
        from . import capi
        box, err = [None], ""
        cuda = self.device.type == "cuda"
        try:
            if cuda:
                if not capi.comm_available():        # dlopen + dlsym on THIS rank, before any rank enters ncclCommInitRank (no bootstrap state)
                    raise RuntimeError(capi.load().spvo_last_error(None).decode())
                if self.rank == 0:
                    box[0] = capi.comm_unique_id()   # only the rank whose id is used asks RCCL for one
            elif self.rank == 0:
                box[0] = tempfile.mkdtemp(prefix="spvo_comm_")
        except Exception as exc:               # noqa: BLE001
            err = repr(exc)
        if not self._all_ok(not err):
            return None, err or "a peer rank cannot open the communicator library"
        if self.world > 1:
            dist.broadcast_object_list(box, src=0)
        comm = None
        try:
            if cuda:
                comm = capi.Comm.rccl(self.device.index or 0, self.rank, self.world, box[0])
            else:
                self._host_dir = box[0]
                comm = capi.Comm.host(box[0], self.rank, self.world)
        except Exception as exc:               # noqa: BLE001
            err = repr(exc)
        if not self._all_ok(comm is not None):
            if comm is not None:
                comm.close()
            return None, err or "a peer rank failed to create its communicator"
        return comm, ""

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
            d = getattr(self, "_host_dir", None)
            if d and self.rank == 0 and os.path.isdir(d):   # every peer has closed (spvo_comm_destroy waits for their done_<rank> markers)
                import shutil
                shutil.rmtree(d, ignore_errors=True)

    def gather(self, q_xyzw, t) -> np.ndarray:
        """Returns [world, 7]; a rank with no pose yet (first frame) contributes the identity."""
        pose = _pose(q_xyzw, t)
        if self.local_only:
            return pose[None].copy()
        if self.comm is not None:
            return self.comm.allgather(pose)[:, 0, :]
        self.buf.copy_(torch.from_numpy(pose))
        dist.all_gather(self.out, self.buf)
        return torch.stack(self.out).cpu().numpy()

    # ---- non-blocking form: the poses are only COLLECTED (nothing downstream of the front end waits for the other
    # streams' poses), so a step need not block on the collective: they are staged on the host and leave in batches.
    def _flush(self) -> None:
        n = self._fill
        if n == 0:
            return
        self._fill = 0
        if self.local_only:
            self._pending.append((self._host[:n].copy()[:, None, :], n))
            return
        if self.comm is not None:          # one spvo_pose_allgather_n per batch: [world, n, 7] -> [n, world, 7]
            t0 = self._clock()
            self._pending.append((self.comm.allgather(self._host[:n]).transpose(1, 0, 2).copy(), n))
            self.collective_ms.append(1e3 * (self._clock() - t0))
            return
        t0 = self._clock()
        slot = self._slot
        self._slot = (slot + 1) % self.SLOTS
        if self._ready[slot] is not None:
            self._ready[slot].synchronize()          # the slot's previous batch has left the staging buffers
        stage, send = self._stage[slot], self._send[slot]
        stage[:n].copy_(torch.from_numpy(self._host[:n]))
        if n < self.BATCH:
            stage[n:].zero_()
        out = torch.empty((self.world, self.BATCH, 7), dtype=torch.float64, device=self.device)
        if self._side is not None:
            with torch.cuda.stream(self._side):
                send.copy_(stage, non_blocking=True)
                dist.all_gather_into_tensor(out.view(self.world * self.BATCH, 7), send)
                ev = torch.cuda.Event()
                ev.record(self._side)
                self._ready[slot] = ev
        else:
            send.copy_(stage)
            dist.all_gather_into_tensor(out.view(self.world * self.BATCH, 7), send)
        self._pending.append((out, n))
        self.collective_ms.append(1e3 * (self._clock() - t0))    # (torch transport on a side stream: the enqueue, not the completion)

    def reset_timing(self) -> None:
        self.collective_ms = []

    def timing(self) -> dict:
        """what the collectives cost the calling thread since reset_timing(): per batched collective of <= BATCH frames"""
        a = np.asarray(self.collective_ms, np.float64)
        if a.size == 0:
            return {"collectives": 0, "frames_per_collective": self.BATCH}
        return {"collectives": int(a.size), "frames_per_collective": self.BATCH, "mean_ms": round(float(a.mean()), 4), "p50_ms": round(float(np.median(a)), 4),
                "max_ms": round(float(a.max()), 4), "total_ms": round(float(a.sum()), 3)}

    def gather_async(self, q_xyzw, t) -> None:
        if self._host is None:
            self._host = np.zeros((self.BATCH, 7), np.float64)
            self._fill = 0
            if not self.local_only and self.comm is None:   # torch transport: pinned staging + device rows, SLOTS batches in flight
                self._stage = [torch.zeros((self.BATCH, 7), dtype=torch.float64) for _ in range(self.SLOTS)]
                if self.device.type == "cuda":
                    self._stage = [s.pin_memory() for s in self._stage]
                    self._side = torch.cuda.Stream(device=self.device)
                self._send = [torch.zeros((self.BATCH, 7), dtype=torch.float64, device=self.device) for _ in range(self.SLOTS)]
                self._ready = [None] * self.SLOTS
                self._slot = 0
        self._host[self._fill, :] = _pose(q_xyzw, t)
        self._fill += 1
        if self._fill == self.BATCH:
            self._flush()

    def collect(self) -> np.ndarray:
        """[steps, world, 7] of everything enqueued since the last collect (flushes the open batch and blocks until the
        collectives are done)."""
        if self._host is not None:
            self._flush()
        if not self._pending:
            return np.zeros((0, self.world, 7))
        if self._side is not None:
            self._side.synchronize()
        rows = []
        for out, n in self._pending:
            if isinstance(out, np.ndarray):
                rows.append(out)
            else:
                rows.append(out.to("cpu")[:, :n, :].permute(1, 0, 2).numpy())
        self._pending.clear()
        return np.concatenate(rows)
