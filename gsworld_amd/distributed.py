"""Multi-GPU layer of BASELINE.json configs[3]: independent scenes shard one-per-GPU, the only exchange is the
gather of finished uint8 frames (RCCL over xGMI on the GPU box, gloo in the CPU tests).

One process per GPU (``torch.distributed``).  No collective sits on the render path: a batch of ``K`` frames is
gathered at once (0.92 MB per 640x480 frame; K amortises the collective launch) and, on GPUs, on a side stream so
that the gather of batch b overlaps the rendering of batch b+1.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

SCENES_PER_NODE = 8


def init_from_env(device: torch.device | None = None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if device is not None and device.type == "cuda":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    return rank, world, local_rank


def scene_for_rank(rank: int, names) -> tuple[str, int]:
    """rank r renders scene r (mod the number of configured scenes) with seed r + 1."""
    return names[rank % len(names)], 1 + rank


class FrameGather:
    """Batches ``batch`` frames of shape (H, W, 3) uint8 per rank and all-gathers them: result
    ``(world * batch, H, W, 3)`` ordered by rank, then by frame slot."""

    def __init__(self, height: int, width: int, batch: int = 16, device="cpu", world: int | None = None):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.batch = max(1, int(batch))
        self.device = torch.device(device)
        self.frames = torch.empty((self.batch, height, width, 3), dtype=torch.uint8, device=self.device)
        self.gathered = torch.empty((self.world * self.batch, height, width, 3), dtype=torch.uint8,
                                    device=self.device) if self.world > 1 else self.frames
        self.stream = torch.cuda.Stream(self.device) if (self.world > 1 and self.device.type == "cuda") else None
        self.num_gathers = 0

    def slot(self, i: int) -> torch.Tensor:
        """Frame buffer that step ``i`` renders / packs into."""
        return self.frames[i % self.batch]

    def step_done(self, i: int) -> bool:
        """Call after step ``i`` wrote its slot.  Launches the gather when the batch is full; returns True then."""
        if (i % self.batch) != self.batch - 1:
            return False
        if self.world > 1:
            if self.stream is not None:
                cur = torch.cuda.current_stream(self.device)
                self.stream.wait_stream(cur)
                with torch.cuda.stream(self.stream):
                    dist.all_gather_into_tensor(self.gathered, self.frames)
                cur.wait_stream(self.stream)  # the next batch overwrites `frames`
            else:
                parts = list(self.gathered.view(self.world, *self.frames.shape).unbind(0))
                dist.all_gather(parts, self.frames)
        self.num_gathers += 1
        return True
