"""Multi-GPU layer of BASELINE.json configs[3]: independent scenes shard one-per-GPU, the only exchange is the
gather of finished uint8 frames (RCCL over xGMI on the GPU box, gloo in the CPU tests).

One process per GPU (``torch.distributed``).  No collective sits on the render path: a batch of ``K`` frames is
gathered at once (0.92 MB per 640x480 frame; K amortises the collective launch) and, on GPUs, on a side stream so
that the gather of batch b overlaps the rendering of batch b+1.
"""
from __future__ import annotations

import os
import time
from collections import deque
from concurrent.futures import ThreadPoolExecutor

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


def rccl_info() -> dict:
    """What the N > 1 bench line reports about the collective library (RCCL is torch's "nccl" backend on ROCm)."""
    info = {"backend": dist.get_backend() if dist.is_initialized() else None,
            "world_size": dist.get_world_size() if dist.is_initialized() else 1}
    try:
        info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as ex:  # noqa: BLE001
        info["rccl_version"] = f"unavailable ({type(ex).__name__})"
    info["hip"] = getattr(torch.version, "hip", None)
    return info


def scene_for_rank(rank: int, names) -> tuple[str, int]:
    """rank r renders scene r (mod the number of configured scenes) with seed r + 1."""
    return names[rank % len(names)], 1 + rank


class FrameGather:
    """Batches ``batch`` frames of shape (H, W, 3) uint8 per rank and gathers them over the process group: result
    ``(world * batch, H, W, 3)`` ordered by rank, then by frame slot.

    ``collective = "gather"`` (default): gather-to-root -- only rank ``dst`` receives the frames, which is what
    north_star asks for ("RCCL over xGMI only to gather frames") and moves ``(world - 1) x batch x 0.92 MB`` into ONE
    rank per batch instead of into every rank.  ``"all_gather"``: every rank receives every frame (8x the traffic on an
    8-GPU node; for consumers that need all frames everywhere).

    ``buffers = 2`` double-buffers the frame slots (and the gathered result): while the collective of batch b runs on
    the side stream, batch b + 1 is rendered into the other half, so nothing on the render streams waits for RCCL.  A
    stream that is about to overwrite the slots of a batch calls :meth:`wait_reusable` first (that collective was issued
    a whole batch earlier); a consumer of the gathered frames calls :meth:`wait_gathered` before reading
    :attr:`gathered`."""

    def __init__(self, height: int, width: int, batch: int = 16, device="cpu", world: int | None = None,
                 buffers: int = 1, collective: str = "gather", dst: int = 0, background: bool = False,
                 timing: bool = False, force_collective: bool = False):
        """``background`` (host tensors only): the collective of a full batch runs on a worker thread, the CPU
        counterpart of the GPU path's side stream -- the caller goes on filling the other half and only
        :meth:`wait_reusable` / :meth:`wait_gathered` block (what the scheduling tests exercise).  ``timing``: record
        how long every collective took (:attr:`gather_ms`; HIP events on the side stream, wall clock on the host) and
        how long callers were held in :meth:`wait_reusable` (:attr:`waits`).  ``force_collective``: issue the collective
        (side stream, events, double buffering and all) even in a process group of ONE rank -- the only way a 1-GPU box can
        put RCCL itself under this class (tools/rccl_world1.py, tests/test_distributed_gpu.py)."""
        if collective not in ("gather", "all_gather"):
            raise ValueError("collective must be 'gather' or 'all_gather'")
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self._multi = self.world > 1 or (bool(force_collective) and dist.is_initialized())
        self.rank = dist.get_rank() if (self._multi and dist.is_initialized()) else 0
        self.collective = collective
        self.dst = dst
        self.batch = max(1, int(batch))
        self.buffers = max(1, int(buffers))
        self.device = torch.device(device)
        self.frames = torch.empty((self.buffers * self.batch, height, width, 3), dtype=torch.uint8, device=self.device)
        self.receives = self._multi and (collective == "all_gather" or self.rank == dst)
        self._gathered = [torch.empty((self.world * self.batch, height, width, 3), dtype=torch.uint8,
                                      device=self.device) if self.receives else self._half(b)
                          for b in range(self.buffers)]
        self.gathered = self._gathered[0]  # result of the most recent gather (this rank's own frames if it is not a receiver)
        self.stream = torch.cuda.Stream(self.device) if (self._multi and self.device.type == "cuda") else None
        self._done = [None] * self.buffers  # event after the collective that last read half b
        self._done_batch = [None] * self.buffers  # ... and which batch that was
        self._last = None  # event of the most recent collective (what `gathered` waits for)
        self.num_gathers = 0
        self.timing = bool(timing)
        self.gather_ms = []      # per collective (timing=True)
        self._timers = []        # (start event, end event) pairs not yet read (GPU)
        # (step, batch waited for, seconds blocked) of the last wait_reusable calls that found a collective -- kept only
        # with timing=True, and bounded: a render loop runs for hours
        self.waits = deque(maxlen=4096)
        self._pool = ThreadPoolExecutor(1) if (background and self.device.type != "cuda" and self._multi) else None

    def _half(self, b: int) -> torch.Tensor:
        return self.frames if self.buffers == 1 else self.frames[b * self.batch:(b + 1) * self.batch]

    @property
    def num_slots(self) -> int:
        return self.buffers * self.batch

    def slot(self, i: int) -> torch.Tensor:
        """Frame buffer that step ``i`` renders / packs into."""
        return self.frames[i % self.num_slots]

    def wait_reusable(self, i: int, stream=None):
        """Makes ``stream`` (default: current) wait until the collective that last read step ``i``'s half has finished.
        Returns the index of the batch whose collective it depended on (None: nothing to wait for) -- with ``buffers``
        halves that is always the batch ``buffers`` batches before step ``i``'s own, never the one just issued."""
        h = (i // self.batch) % self.buffers
        ev = self._done[h]
        if ev is None:
            return None
        if hasattr(ev, "result"):  # host path with a worker thread
            t0 = time.perf_counter()
            ev.result()
            if self.timing:
                self.waits.append((i, self._done_batch[h], time.perf_counter() - t0))
        else:
            (stream if stream is not None else torch.cuda.current_stream(self.device)).wait_event(ev)
            if self.timing:
                self.waits.append((i, self._done_batch[h], 0.0))  # (a stream dependency: the host is never blocked)
        return self._done_batch[h]

    def wait_gathered(self, stream=None) -> torch.Tensor:
        """Makes ``stream`` (default: current) wait for the most recent collective and returns :attr:`gathered`: the
        collective runs on a side stream, so a consumer must call this before it reads the frames."""
        if self._last is not None:
            if hasattr(self._last, "result"):
                self._last.result()
            else:
                (stream if stream is not None else torch.cuda.current_stream(self.device)).wait_event(self._last)
        return self.gathered

    def _drain_timers(self, keep: int = 0):
        """Reads the finished event pairs into ``gather_ms`` (all but the ``keep`` youngest: no wait for collectives
        still in flight) and bounds the history."""
        done = self._timers[:len(self._timers) - keep] if keep else self._timers
        for e0, e1 in done:
            e1.synchronize()
            self.gather_ms.append(e0.elapsed_time(e1))
        self._timers = self._timers[len(done):]
        if len(self.gather_ms) > 8192:
            self.gather_ms = self.gather_ms[-4096:]

    def gather_time_ms(self):
        """Mean / max duration of the collectives issued so far (``timing=True``; synchronises the side stream)."""
        self._drain_timers()
        if not self.gather_ms:
            return None
        return {"mean": sum(self.gather_ms) / len(self.gather_ms), "max": max(self.gather_ms), "count": len(self.gather_ms)}

    def _timed_collective(self, dst_buf, src):
        t0 = time.perf_counter()
        self._collective(dst_buf, src)
        if self.timing:
            self.gather_ms.append(1e3 * (time.perf_counter() - t0))

    def _collective(self, dst_buf: torch.Tensor, src: torch.Tensor) -> None:
        if self.device.type == "cuda" and dist.get_backend() == "gloo":
            # test hook only (GSWORLD_DIST_BACKEND=gloo, several ranks on one GPU): gloo moves host tensors
            host = src.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)] if self.receives else None
            if self.collective == "all_gather":
                dist.all_gather(parts, host)
            else:
                dist.gather(host, gather_list=parts, dst=self.dst)
            if self.receives:
                dst_buf.copy_(torch.cat(parts))
            return
        if self.collective == "all_gather":
            if self.device.type == "cuda":
                dist.all_gather_into_tensor(dst_buf, src)
            else:
                dist.all_gather(list(dst_buf.view(self.world, *src.shape).unbind(0)), src)
        else:
            parts = list(dst_buf.view(self.world, *src.shape).unbind(0)) if self.rank == self.dst else None
            dist.gather(src, gather_list=parts, dst=self.dst)

    def step_done(self, i: int) -> bool:
        """Call after step ``i`` wrote its slot.  Launches the collective when the batch is full; returns True then."""
        if (i % self.batch) != self.batch - 1:
            return False
        b = (i // self.batch) % self.buffers
        src, dst_buf = self._half(b), self._gathered[b]
        if self._multi:
            if self.stream is not None:
                cur = torch.cuda.current_stream(self.device)
                self.stream.wait_stream(cur)
                with torch.cuda.stream(self.stream):
                    if self.timing:
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record(self.stream)
                    self._collective(dst_buf, src)
                    ev = torch.cuda.Event(enable_timing=self.timing)
                    ev.record(self.stream)
                    if self.timing:
                        self._timers.append((e0, ev))
                        if len(self._timers) >= 256:  # (drained here as well: nobody may ever call gather_time_ms)
                            self._drain_timers(keep=64)
                self._done[b] = ev
                self._done_batch[b] = i // self.batch
                self._last = ev
                if self.buffers == 1:
                    cur.wait_stream(self.stream)  # the next batch overwrites the same slots
            elif self._pool is not None:
                fut = self._pool.submit(self._timed_collective, dst_buf, src)
                self._done[b] = fut
                self._done_batch[b] = i // self.batch
                self._last = fut
                if self.buffers == 1:
                    fut.result()
            else:
                self._timed_collective(dst_buf, src)
        self.gathered = dst_buf
        self.num_gathers += 1
        return True
