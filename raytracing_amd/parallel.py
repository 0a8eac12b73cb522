"""Screen partition used by the multi-GPU path, restated on the host with numpy: interleaved
tile_w x tile_h tiles (tile t -> part t % part_count), 8x8 sub-blocks inside a tile, four 4x4 patches
inside a sub-block — the same
slot order the device writes with rthip_params.layout = 1 and reads in rthip_assemble (k_assemble).
Used by the CPU (gloo) tests and as documentation of the layout; the GPU path gathers with
torch.distributed (RCCL) and assembles on the device."""
import numpy as np


def _slot_xy(w, h, part, part_count, tile_w=32, tile_h=8):
    """(x, y, valid) for every slot of `part`'s tile-compact buffer."""
    tiles_x, tiles_y = -(-w // tile_w), -(-h // tile_h)
    n_tiles = tiles_x * tiles_y
    per_tile = tile_w * tile_h
    part_tiles = -(-n_tiles // part_count)
    slot = np.arange(part_tiles * per_tile, dtype=np.int64)
    q, r = slot // per_tile, slot % per_tile
    t = part + q * part_count
    tx, ty = t % tiles_x, t // tiles_x
    sb, l = r >> 6, r & 63
    sbx = tile_w >> 3
    x = tx * tile_w + (sb % sbx) * 8 + ((l & 3) | ((l >> 2) & 4))
    y = ty * tile_h + (sb // sbx) * 8 + (((l >> 2) & 3) | ((l >> 3) & 4))
    valid = (t < n_tiles) & (x < w) & (y < h)
    return x, y, valid


def part_slots(w, h, part_count, tile_w=32, tile_h=8):
    return -(-(-(-w // tile_w) * -(-h // tile_h)) // part_count) * tile_w * tile_h


def frame_to_part(frame, part, part_count, tile_w=32, tile_h=8):
    h, w = frame.shape
    x, y, valid = _slot_xy(w, h, part, part_count, tile_w, tile_h)
    out = np.zeros(len(x), dtype=frame.dtype)
    out[valid] = frame[y[valid], x[valid]]
    return out


def assemble_parts(parts, w, h, tile_w=32, tile_h=8):
    part_count = parts.shape[0]
    frame = np.zeros((h, w), dtype=parts.dtype)
    for p in range(part_count):
        x, y, valid = _slot_xy(w, h, p, part_count, tile_w, tile_h)
        frame[y[valid], x[valid]] = parts[p][valid]
    return frame


def pixel_owner(w, h, part_count, tile_w=32, tile_h=8):
    tiles_x = -(-w // tile_w)
    ys, xs = np.mgrid[0:h, 0:w]
    return ((ys // tile_h) * tiles_x + xs // tile_w) % part_count


def collect_parts(mine, gathered, how="gather", dst=0):
    """The framebuffer exchange of one call (torch.distributed; RCCL on GPUs, gloo in the CPU tests): every rank
    contributes `mine` (its tile-compact part buffers, frame after frame), `gathered` receives them rank after rank.
    how = "gather": only rank `dst` receives (grouped send / recv: each peer sends its part once over its own xGMI
    link; the other ranks' `gathered` stays untouched); "all_gather": every rank receives everything.
    Returns True on the ranks whose `gathered` is complete."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    assert gathered.numel() == mine.numel() * world
    if mine.is_cuda and dist.get_backend() == "gloo":
        # rehearsal of the multi-rank control flow with every rank on ONE GPU (bench.py --rehearse-on-one-gpu; RCCL refuses
        # two ranks on a device, gloo has no device gather): the exchange is staged through host memory
        host_mine, host_all = mine.cpu(), gathered.cpu()
        complete = collect_parts(host_mine, host_all, how, dst)
        if complete:
            gathered.copy_(host_all)
        return complete
    if how == "all_gather":
        dist.all_gather_into_tensor(gathered, mine)
        return True
    if how != "gather":
        raise ValueError(how)
    chunks = list(gathered.chunk(world)) if rank == dst else None
    dist.gather(mine, chunks, dst=dst)
    return rank == dst
