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
    RING = 4096   # outstanding non-blocking gathers between two collect() calls

    def __init__(self, device: Optional[torch.device] = None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.device = device or torch.device("cpu")
        self.buf = torch.zeros(7, dtype=torch.float64, device=self.device)
        self.out: List[torch.Tensor] = [torch.zeros(7, dtype=torch.float64, device=self.device) for _ in range(self.world)]
        self._pending: List[torch.Tensor] = []
        self._stage = None
        self._send = None

    def gather(self, q_xyzw, t) -> np.ndarray:
        """Returns [world, 7]; a rank with no pose yet (first frame) contributes the identity."""
        pose = IDENTITY_POSE if q_xyzw is None else np.concatenate([np.asarray(q_xyzw, np.float64), np.asarray(t, np.float64)])
        if self.world == 1:
            return pose[None].copy()
        self.buf.copy_(torch.from_numpy(pose))
        dist.all_gather(self.out, self.buf)
        return torch.stack(self.out).cpu().numpy()

    # ---- non-blocking form: the poses are only COLLECTED (nothing downstream of the front end waits for the other
    # streams' poses), so a step need not block on the collective: it is enqueued and read back later, in order.
    def gather_async(self, q_xyzw, t) -> None:
        pose = IDENTITY_POSE if q_xyzw is None else np.concatenate([np.asarray(q_xyzw, np.float64), np.asarray(t, np.float64)])
        if self.world == 1:
            self._pending.append(torch.from_numpy(pose[None].copy()))
            return
        k = len(self._pending) % self.RING
        if self._stage is None:   # pinned staging ring + one device row per outstanding step
            self._stage = torch.zeros((self.RING, 7), dtype=torch.float64)
            if self.device.type == "cuda":
                self._stage = self._stage.pin_memory()
            self._send = torch.zeros((self.RING, 7), dtype=torch.float64, device=self.device)
        if len(self._pending) >= self.RING:
            raise RuntimeError("PoseGather: collect() at least every %d steps" % self.RING)
        self._stage[k].copy_(torch.from_numpy(pose))
        self._send[k].copy_(self._stage[k], non_blocking=True)
        out = torch.empty((self.world, 7), dtype=torch.float64, device=self.device)
        dist.all_gather_into_tensor(out, self._send[k:k + 1])   # [1, 7] in, [world, 7] out: the same chunking on RCCL and gloo
        self._pending.append(out)

    def collect(self) -> np.ndarray:
        """[steps, world, 7] of everything enqueued since the last collect (blocks until the collectives are done)."""
        if not self._pending:
            return np.zeros((0, self.world, 7))
        res = torch.stack([p.to("cpu") for p in self._pending]).numpy()
        self._pending.clear()
        return res
