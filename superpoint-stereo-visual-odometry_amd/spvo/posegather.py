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
    def __init__(self, device: Optional[torch.device] = None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.device = device or torch.device("cpu")
        self.buf = torch.zeros(7, dtype=torch.float64, device=self.device)
        self.out: List[torch.Tensor] = [torch.zeros(7, dtype=torch.float64, device=self.device) for _ in range(self.world)]

    def gather(self, q_xyzw, t) -> np.ndarray:
        """Returns [world, 7]; a rank with no pose yet (first frame) contributes the identity."""
        pose = IDENTITY_POSE if q_xyzw is None else np.concatenate([np.asarray(q_xyzw, np.float64), np.asarray(t, np.float64)])
        if self.world == 1:
            return pose[None].copy()
        self.buf.copy_(torch.from_numpy(pose))
        dist.all_gather(self.out, self.buf)
        return torch.stack(self.out).cpu().numpy()
