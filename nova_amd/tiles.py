"""Image-tile partition of a frame across the GPUs of a node, and the final gather.

The reference renders on one device only.  Pixels are independent — the RNG stream
depends only on (x, y, W, time) (reference ``kernels/kernels.cu:27``) and the scene is read-only —
so the frame is cut into 64x64 tiles dealt round-robin over ranks (``tile_id % ranks == rank``,
the rule ``NOVA_HIP_OPT_TILE_RANK/RANKS`` implements inside ``kernel_raytrace``), with the scene
replicated per GPU and NO communication per sample.  Once per displayed/final frame the finished
tiles are gathered to rank 0 with one collective (RCCL all_gather over equal-sized tile stacks;
every peer has its own xGMI link to the root, so this is one hop).

One process per GPU.  The production path is ``finish_frame(..., via="capi")``: ONE call of the C ABI,
``nova_hip_tiles_allgather``, packs a rank's tiles into a contiguous stack, moves the stacks with
``ncclAllGather`` (RCCL, linked by the library itself) and scatters them into the whole image, all on the
library's stream — the same call a multi-process launch of Nova makes from C++.
Its communicator is made by ``init_comm``: rank 0 draws an ncclUniqueId through the C ABI and
``torch.distributed`` (plumbing) hands the 128 bytes to the other ranks.  ``via="torch"`` keeps round 2's
path for rehearsals where several ranks share one GPU (RCCL refuses two ranks on a device): the C ABI packs
(``nova_hip_tiles_pack``), ``torch.distributed.all_gather_into_tensor`` moves the stacks (``nccl`` or, through
host memory, ``gloo``), rank 0 unpacks (``nova_hip_tiles_unpack``).  With FXAA the float4 running means are gathered and rank 0
post-processes the whole frame (FXAA reads up to ~26 px across tile borders, reference
``kernels/anti_aliasing.hpp:99-116``); without it every rank post-processes its own pixels and the
uchar4 pixels are gathered (a quarter of the bytes).  ``pack_tiles``/``unpack_tiles`` are the same
layout in torch, used by the CPU (gloo) tests.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

TILE = 64


def tile_grid(width: int, height: int) -> Tuple[int, int]:
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def tiles_of_rank(width: int, height: int, rank: int, ranks: int) -> List[int]:
    tx, ty = tile_grid(width, height)
    return list(range(rank, tx * ty, ranks))


def max_tiles_per_rank(width: int, height: int, ranks: int) -> int:
    tx, ty = tile_grid(width, height)
    return (tx * ty + ranks - 1) // ranks


def pack_tiles(frame, width: int, height: int, rank: int, ranks: int):
    """frame: torch tensor [H, W, C] -> [max_tiles_per_rank, 64, 64, C] holding this rank's tiles
    in tile-id order (zero padded at the image border and at the end of the stack)."""
    import torch

    tx, ty = tile_grid(width, height)
    c = frame.shape[-1]
    padded = torch.zeros((ty * TILE, tx * TILE, c), dtype=frame.dtype, device=frame.device)
    padded[:height, :width] = frame
    tiles = padded.view(ty, TILE, tx, TILE, c).permute(0, 2, 1, 3, 4).reshape(tx * ty, TILE, TILE, c)
    mine = torch.as_tensor(tiles_of_rank(width, height, rank, ranks), device=frame.device, dtype=torch.long)
    out = torch.zeros((max_tiles_per_rank(width, height, ranks), TILE, TILE, c), dtype=frame.dtype,
                      device=frame.device)
    out[: mine.numel()] = tiles[mine]
    return out


def unpack_tiles(stacks, width: int, height: int, ranks: int):
    """stacks: [ranks, max_tiles_per_rank, 64, 64, C] -> frame [H, W, C]."""
    import torch

    tx, ty = tile_grid(width, height)
    c = stacks.shape[-1]
    tiles = torch.zeros((tx * ty, TILE, TILE, c), dtype=stacks.dtype, device=stacks.device)
    for r in range(ranks):
        ids = tiles_of_rank(width, height, r, ranks)
        if ids:
            tiles[torch.as_tensor(ids, device=stacks.device)] = stacks[r, : len(ids)]
    frame = tiles.view(ty, tx, TILE, TILE, c).permute(0, 2, 1, 3, 4).reshape(ty * TILE, tx * TILE, c)
    return frame[:height, :width].contiguous()


def gather_frame(local_frame, width: int, height: int, rank: int, ranks: int, group=None):
    """All ranks call this with their locally rendered frame (only their own tiles need to be
    valid).  Returns the assembled frame on rank 0 and None elsewhere."""
    import torch
    import torch.distributed as dist

    mine = pack_tiles(local_frame, width, height, rank, ranks)
    if ranks == 1:
        return unpack_tiles(mine.unsqueeze(0), width, height, 1)
    gathered = torch.empty((ranks * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype,
                           device=mine.device)
    dist.all_gather_into_tensor(gathered, mine, group=group)  # rank r's stack lands at [r*n, (r+1)*n)
    if rank != 0:
        return None
    return unpack_tiles(gathered.view((ranks,) + tuple(mine.shape)), width, height, ranks)


def gather_image(image, rank: int, ranks: int, coll_device: str = "cuda", group=None):
    """All ranks: ``image`` (capi.Image, float4 or uchar4) holds valid pixels in this rank's tiles.
    After the call rank 0's image holds every rank's tiles.  One collective."""
    import torch
    import torch.distributed as dist

    from . import capi

    nbytes = capi.tiles_stack_bytes(image.width, image.height, ranks, image.texel_bytes)
    mine = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    capi.tiles_pack(image, rank, ranks, mine.data_ptr())  # returns after the device has finished
    if ranks == 1:
        capi.tiles_unpack(image, 1, mine.data_ptr())
        return
    send = mine if coll_device == "cuda" else mine.cpu()
    gathered = torch.empty(ranks * nbytes, dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(gathered, send, group=group)  # rank r's stack lands at [r*n, (r+1)*n)
    if rank != 0:
        return
    if coll_device != "cuda":
        gathered = gathered.cuda()
    torch.cuda.synchronize()  # the collective runs on torch's stream, the unpack on the library's
    capi.tiles_unpack(image, ranks, gathered.data_ptr())


def init_comm(rank: int, ranks: int, group=None) -> None:
    """The library's own RCCL communicator (``nova_hip_tiles_comm_init``) for ``finish_frame(via="capi")``:
    rank 0 draws the unique id, ``torch.distributed`` broadcasts its 128 bytes.  Collective over the group."""
    import torch.distributed as dist

    from . import capi

    box = [capi.tiles_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    capi.tiles_comm_init(box[0], rank, ranks)


def finish_frame(rt, rank: int, ranks: int, coll_device: str = "cuda", group=None, via: str = "torch") -> bool:
    """The final frame of a tile-partitioned render: afterwards rank 0's ``rt.color_img`` holds the whole
    post-processed frame (returns True there).  ``rt`` is a nova_amd.raytracer.Raytracer whose running
    mean (``temp_color_img2``) is up to date on this rank's tiles.  ``via="capi"``: the gather is
    ``nova_hip_tiles_allgather`` (after ``init_comm``); ``via="torch"``: ``gather_image``."""
    from . import capi

    if via == "capi":
        gather = lambda image: capi.tiles_allgather(image)  # noqa: E731
    else:
        gather = lambda image: gather_image(image, rank, ranks, coll_device, group)  # noqa: E731
    if bool(rt.params["anti_aliasing"]):
        gather(rt.temp_color_img2)
        if rank == 0:
            capi.set_option(capi.OPT_TILE_RANKS, 1)  # the image is whole now
            try:
                rt.kernel_post_process()
            finally:
                capi.set_option(capi.OPT_TILE_RANKS, ranks)
    else:
        rt.kernel_post_process()
        gather(rt.color_img)
    return rank == 0


class DeviceArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so that torch can wrap the
    backend's image memory without a copy (torch.as_tensor(DeviceArray(...), device='cuda'))."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def image_as_tensor(image, device):
    """capi.Image -> torch tensor view [H, W, 4] (float32 or uint8) on `device`."""
    import torch

    typestr = "<f4" if image.texel_bytes == 16 else "|u1"
    arr = DeviceArray(image.data_ptr, (image.height, image.width, 4), typestr)
    return torch.as_tensor(arr, device=device)


def weak_scaling_dims(n_gpus: int, base=(1920, 1080)) -> Tuple[int, int]:
    """Frame size with n_gpus x the pixels of `base` at (nearly) the same aspect: 1 -> 1920x1080,
    4 -> 3840x2160; widths are multiples of 64 so that tiles are whole."""
    if n_gpus == 1:
        return base
    w = int(round(base[0] * np.sqrt(n_gpus) / 64.0)) * 64
    h = int(round(w * base[1] / base[0]))
    return w, h
