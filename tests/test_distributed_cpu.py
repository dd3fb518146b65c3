"""world_size-2 gloo test of the N > 1 path (BASELINE.json configs[3]): scene sharding and the frame gather."""
import os
import socket

import torch
import torch.multiprocessing as mp

from gsworld_amd import distributed as gd
from gsworld_amd import scenes


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, buffers, collective):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = gd.init_from_env(torch.device("cpu"))
    assert (r, w) == (rank, world)
    name, seed = gd.scene_for_rank(rank, scenes.SCENE_NAMES)
    H, W, K = 6, 8, 4
    fg = gd.FrameGather(H, W, batch=K, device="cpu", buffers=buffers, collective=collective)
    assert fg.num_slots == buffers * K
    got = []
    for i in range(2 * K + 1):  # the last frame starts a batch that is never gathered
        fg.slot(i).fill_((10 * rank + i) % 256)
        if fg.step_done(i):
            got.append(fg.wait_gathered().clone())
    torch.distributed.barrier()
    q.put((rank, name, seed, fg.num_gathers, [g[:, 0, 0, 0].tolist() for g in got]))
    torch.distributed.destroy_process_group()


import pytest


@pytest.mark.parametrize("buffers,collective", [(1, "gather"), (2, "gather"), (2, "all_gather")])
def test_two_rank_frame_gather_and_scene_sharding(buffers, collective):
    # buffers = 2: the double-buffered slots bench.py uses for N > 1 (gather of batch b overlaps rendering of b + 1)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, buffers, collective)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [o[1] for o in out] == ["xarm6_align", "xarm6_rot_banana"] and [o[2] for o in out] == [1, 2]
    for rank, _, _, n, batches in out:
        assert n == 2
        if collective == "all_gather" or rank == 0:
            # the receiver(s) hold every rank's frames, rank-major then slot order
            assert batches[0] == [0, 1, 2, 3, 10, 11, 12, 13]
            assert batches[1] == [4, 5, 6, 7, 14, 15, 16, 17]
        else:  # gather-to-root: the other ranks only ever see their own frames
            assert batches[0] == [10, 11, 12, 13] and batches[1] == [14, 15, 16, 17]


def test_single_process_gather_is_identity():
    fg = gd.FrameGather(4, 4, batch=2, device="cpu", world=1)
    fg.slot(0).fill_(1)
    assert not fg.step_done(0)
    fg.slot(1).fill_(2)
    assert fg.step_done(1) and fg.gathered is fg.frames and fg.num_gathers == 1
    assert gd.scene_for_rank(9, scenes.SCENE_NAMES) == ("xarm6_rot_banana", 10)
