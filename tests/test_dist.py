"""CPU, world_size 2 over gloo: the multi-GPU framebuffer path (tile-compact buffers of interleaved screen tiles of a
batch of frames, parallel.collect_parts as bench.py calls it — gather to rank 0, to rank 1 and all-gather —, assemble) with the
per-rank rendering done by the host-compiled core."""
import ctypes as C
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from raytracing_amd import device, host, parallel, scenes


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hs = host.HostScene(scenes.config(1)[0])
    HC = C.CDLL(os.path.join(ROOT, "tests", "hostcore", "libhostcore.so"))
    vp = C.c_void_p
    n_frames = 3  # a batch: one camera per frame
    frames = []
    for f in range(n_frames):
        cam = hs.camera(w, h)
        cam.position[0] = 0.4 * f
        prm = device.make_params(w, h, 1)
        px = np.zeros((h, w), np.uint32)
        HC.hc_render(vp(hs.flatten()), C.byref(cam), C.byref(prm), px.ctypes.data_as(vp), None, None, None, None, 2)
        frames.append(px)
    # this rank keeps only ITS tiles of every frame, in the tile-compact order the device writes (layout = 1),
    # frame after frame: what rthip_render_batch_device leaves in HBM
    mine = torch.from_numpy(np.concatenate([parallel.frame_to_part(px, rank, world) for px in frames]).view(np.int32))
    ok = True
    # bench.py sends call k's parts to rank k % world: both destinations
    for how, dst in (("gather", 0), ("gather", 1), ("all_gather", 0)):
        gathered = torch.zeros(mine.numel() * world, dtype=torch.int32)
        complete = parallel.collect_parts(mine, gathered, how, dst=dst)
        assert complete == (how == "all_gather" or rank == dst)
        if complete:  # [rank][frame][slot] -> frames: the layout rthip_assemble_batch reads
            g = gathered.numpy().view(np.uint32).reshape(world, n_frames, -1)
            for f in range(n_frames):
                ok = ok and np.array_equal(parallel.assemble_parts(g[:, f], w, h), frames[f])
        elif gathered.abs().sum().item() != 0:
            ok = False  # a gather must leave the other ranks' buffers alone
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_the_frame():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 200, 120, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}, "frames differ after the exchange + assemble"


def test_partition_covers_every_pixel_exactly_once():
    for (w, h, n) in ((1920, 1080, 8), (200, 120, 3), (33, 9, 2), (7680, 4320, 8)):
        owner = parallel.pixel_owner(w, h, n)
        n_tiles = -(-w // 32) * -(-h // 8)
        assert owner.shape == (h, w) and owner.min() == 0 and owner.max() == min(n, n_tiles) - 1
        counts = np.bincount(owner.ravel(), minlength=n)
        assert counts.sum() == w * h
        assert counts.max() - counts.min() <= 2 * 256 * max(1, 1)  # interleave keeps shares even
        frame = np.arange(w * h, dtype=np.uint32).reshape(h, w)
        parts = np.stack([parallel.frame_to_part(frame, r, n) for r in range(n)])
        assert np.array_equal(parallel.assemble_parts(parts, w, h), frame)
