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
        # GSWORLD_DIST_BACKEND=gloo: test hook (exercise the N > 1 control flow of bench.py with several ranks sharing
        # the one GPU of a test box; RCCL needs one device per rank)
        backend = os.environ.get("GSWORLD_DIST_BACKEND", "")
        if device is not None and device.type == "cuda" and backend != "gloo":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    return rank, world, local_rank


def scene_for_rank(rank: int, names) -> tuple[str, int]:
    """rank r renders scene r (mod the number of configured scenes) with seed r + 1."""
    return names[rank % len(names)], 1 + rank


class FrameGather:
    """Batches ``batch`` frames of shape (H, W, 3) uint8 per rank and all-gathers them: result
    ``(world * batch, H, W, 3)`` ordered by rank, then by frame slot.

    ``buffers = 2`` double-buffers the frame slots (and the gathered result): while the collective of batch b runs on
    the side stream, batch b + 1 is rendered into the other half, so nothing on the render streams waits for RCCL
    (8 ranks x 16 frames x 0.92 MB = 118 MB per gather would otherwise stall every batch).  A stream that is about to
    overwrite the slots of a batch calls :meth:`wait_reusable` first; that gather was issued a whole batch earlier."""

    def __init__(self, height: int, width: int, batch: int = 16, device="cpu", world: int | None = None,
                 buffers: int = 1):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.batch = max(1, int(batch))
        self.buffers = max(1, int(buffers))
        self.device = torch.device(device)
        self.frames = torch.empty((self.buffers * self.batch, height, width, 3), dtype=torch.uint8, device=self.device)
        self._gathered = [torch.empty((self.world * self.batch, height, width, 3), dtype=torch.uint8,
                                      device=self.device) if self.world > 1 else self._half(b)
                          for b in range(self.buffers)]
        self.gathered = self._gathered[0]  # result of the most recent gather
        self.stream = torch.cuda.Stream(self.device) if (self.world > 1 and self.device.type == "cuda") else None
        self._done = [None] * self.buffers  # event after the gather that last read half b
        self.num_gathers = 0

    def _half(self, b: int) -> torch.Tensor:
        return self.frames if self.buffers == 1 else self.frames[b * self.batch:(b + 1) * self.batch]

    @property
    def num_slots(self) -> int:
        return self.buffers * self.batch

    def slot(self, i: int) -> torch.Tensor:
        """Frame buffer that step ``i`` renders / packs into."""
        return self.frames[i % self.num_slots]

    def wait_reusable(self, i: int, stream=None) -> None:
        """Makes ``stream`` (default: current) wait until the gather that last read step ``i``'s half has finished."""
        ev = self._done[(i // self.batch) % self.buffers]
        if ev is not None:
            (stream if stream is not None else torch.cuda.current_stream(self.device)).wait_event(ev)

    def step_done(self, i: int) -> bool:
        """Call after step ``i`` wrote its slot.  Launches the gather when the batch is full; returns True then."""
        if (i % self.batch) != self.batch - 1:
            return False
        b = (i // self.batch) % self.buffers
        src, dst = self._half(b), self._gathered[b]
        if self.world > 1:
            if self.stream is not None:
                cur = torch.cuda.current_stream(self.device)
                self.stream.wait_stream(cur)
                with torch.cuda.stream(self.stream):
                    dist.all_gather_into_tensor(dst, src)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                self._done[b] = ev
                if self.buffers == 1:
                    cur.wait_stream(self.stream)  # the next batch overwrites the same slots
            else:
                parts = list(dst.view(self.world, *src.shape).unbind(0))
                dist.all_gather(parts, src)
        self.gathered = dst
        self.num_gathers += 1
        return True
