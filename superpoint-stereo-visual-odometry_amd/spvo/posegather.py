"""Multi-GPU layer of the path: N independent stereo streams, one rank per GPU, and the ONE
collective the path needs -- an all-gather of each rank's per-frame relative pose
(7 doubles: quaternion x, y, z, w + translation; 56 bytes per rank).

torch.distributed is plumbing here: backend "nccl" is RCCL over xGMI on the MI355X node,
"gloo" is used by the CPU tests.  The message is latency-bound (no bandwidth term), so there is
nothing to bucket or overlap; streams never exchange image or feature data.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

IDENTITY_POSE = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)


def stream_seed(rank: int, base_seed: int = 0) -> int:
    """Each rank renders / reads its own stereo stream."""
    return base_seed + rank


class PoseGather:
    SLOTS = 4     # batches of the non-blocking form that may be in flight

    def __init__(self, device: Optional[torch.device] = None, force: bool = False):
        """force: run the collectives in a one-rank process group as well (test hook)"""
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.local_only = self.world == 1 and not (force and dist.is_initialized())
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.device = device or torch.device("cpu")
        self.buf = torch.zeros(7, dtype=torch.float64, device=self.device)
        self.out: List[torch.Tensor] = [torch.zeros(7, dtype=torch.float64, device=self.device) for _ in range(self.world)]
        self._pending: list = []
        self._host = None
        self._fill = 0
        self._stage = self._send = self._ready = None
        self._side = None

    def gather(self, q_xyzw, t) -> np.ndarray:
        """Returns [world, 7]; a rank with no pose yet (first frame) contributes the identity."""
        pose = IDENTITY_POSE if q_xyzw is None else np.concatenate([np.asarray(q_xyzw, np.float64), np.asarray(t, np.float64)])
        if self.local_only:
            return pose[None].copy()
        self.buf.copy_(torch.from_numpy(pose))
        dist.all_gather(self.out, self.buf)
        return torch.stack(self.out).cpu().numpy()

    # ---- non-blocking form: the poses are only COLLECTED (nothing downstream of the front end waits for the other
    # streams' poses), so a step need not block on the collective.  Poses are staged on the host and leave in batches of
    # BATCH steps: one [BATCH, 7] all-gather on a side stream instead of one collective per frame -- a per-frame collective
    # costs a kernel launch, two stream hand-overs and, issued from the framework's default (NULL) stream, an implicit
    # synchronisation with every blocking stream of the process (measured: 707 -> 524 frames/s on one GPU).
    BATCH = 64

    def _flush(self) -> None:
        n = self._fill
        if n == 0:
            return
        self._fill = 0
        if self.local_only:
            self._pending.append((torch.from_numpy(self._host[:n].copy())[:, None, :], n))
            return
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

    def gather_async(self, q_xyzw, t) -> None:
        if self._host is None:
            self._host = np.zeros((self.BATCH, 7), np.float64)
            self._fill = 0
            if not self.local_only:   # pinned staging + device rows, SLOTS batches may be in flight
                self._stage = [torch.zeros((self.BATCH, 7), dtype=torch.float64) for _ in range(self.SLOTS)]
                if self.device.type == "cuda":
                    self._stage = [s.pin_memory() for s in self._stage]
                    self._side = torch.cuda.Stream(device=self.device)
                self._send = [torch.zeros((self.BATCH, 7), dtype=torch.float64, device=self.device) for _ in range(self.SLOTS)]
                self._ready = [None] * self.SLOTS
                self._slot = 0
        self._host[self._fill, :] = IDENTITY_POSE if q_xyzw is None else np.concatenate([np.asarray(q_xyzw, np.float64), np.asarray(t, np.float64)])
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
            o = out.to("cpu")
            rows.append(o if self.local_only else o[:, :n, :].permute(1, 0, 2))
        self._pending.clear()
        return torch.cat(rows).numpy()
