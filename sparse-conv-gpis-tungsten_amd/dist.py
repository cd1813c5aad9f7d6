"""Multi-GPU sharding of the scene-S tile driver: one process per GPU over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The path shards by independent units: every sample's randomness is a pure
function of (pixel, spp index, bounce, sceneSeed, globalSeed), so ranks never exchange anything
while marching.  Two shardings are provided:

  "spp"   rank r renders sample indices [r*spp, (r+1)*spp) of every pixel (per-GPU work fixed →
          weak scaling); the per-rank radiance sums are added with ONE reduce(sum) to rank 0.
  "rows"  the image's 16-pixel tile rows (PathTraceIntegrator.hpp:27) are dealt round-robin to the
          ranks (total work fixed → strong scaling, the same image at every world size); each rank
          renders its rows as ONE batch (gpis_scene_s.shard_index / shard_count); partial images are
          disjoint, so every rank packs the rows it rendered and ONE gather to rank 0 assembles the
          frame (1/N of the frame per rank on the wire: 4.1 MB per rank for the 8-GPU 3840x2160 frame,
          against 33 MB per rank for a reduce(sum) of zero-filled full frames).

Both need exactly two collectives per job: the broadcast of the POD parameter block from rank 0
and the final gather / reduce.  `render_into(scene_record, radiance)` is injected so that the
CPU tests can drive the same logic with a CPU renderer.  `render_sharded` returns the seconds this
rank spent rendering and in the collective, so that bench.py can report the spread over the ranks
(an imbalance of the interleaved rows would otherwise hide behind the max).
"""
import time

import numpy as np


def broadcast_params(params_rank0, dtype, dist, device=None):
    """Rank 0 owns the medium parameters (a numpy record of `dtype`); every rank gets a copy."""
    import torch
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.array(params_rank0, dtype=dtype)
    n = dtype.itemsize
    if dist.get_rank() == 0:
        blob = torch.from_numpy(np.frombuffer(np.array(params_rank0, dtype=dtype).tobytes(), dtype=np.uint8).copy())
    else:
        blob = torch.zeros(n, dtype=torch.uint8)
    if device is not None:
        blob = blob.to(device)
    dist.broadcast(blob, src=0)
    return np.frombuffer(blob.cpu().numpy().tobytes(), dtype=dtype)[0].copy()


def rows_tile(scene, world):
    """Height of the tile rows the "rows" sharding deals out: 16 pixels (the reference's tile size) unless that leaves a rank
    with fewer than 16 tile rows, then halved until it does not — 1080 rows over 8 ranks in 16-pixel rows are 9 rows for four
    ranks and 8 for the others (6 % imbalance, measured 0.89 compute efficiency in profiles/r03_shard_timing.json); in 8-pixel
    rows 17 and 16 (0.7 %).  Results do not depend on it: every seed derives from the pixel."""
    tile = 16
    while tile > 1 and int(scene["y_count"]) // tile < 16 * max(world, 1):
        tile //= 2
    return tile


def shard_scene(scene, rank, world, mode="spp", tile=None):
    """The list of scene records (row ranges / spp slices) this rank renders."""
    scene = np.array(scene)
    if tile is None:
        tile = rows_tile(scene, world)
    if world == 1:
        return [scene.copy()]
    if mode == "spp":
        part = scene.copy()
        spp = int(scene["spp_count"])
        part["spp_begin"] = int(scene["spp_begin"]) + rank * spp
        return [part]
    if mode == "rows":
        # ONE record per rank: the driver itself walks the rank's interleaved tile rows (gpis_scene_s.shard_*),
        # so every stage of the frame stays a single launch per rank
        part = scene.copy()
        part["tile_size"] = tile
        part["shard_index"] = rank
        part["shard_count"] = world
        return [part]
    raise ValueError("unknown sharding mode %r" % mode)


def shard_rows(scene, rank, world, tile=None):
    """Image rows of rank `rank` under the "rows" sharding (for tests and bookkeeping)."""
    if tile is None:
        tile = rows_tile(scene, world)
    y0, n = int(scene["y_begin"]), int(scene["y_count"])
    return [y0 + y for y in range(n) if world <= 1 or (y // tile) % world == rank]


def total_spp(scene, world, mode):
    return int(scene["spp_count"]) * (world if mode == "spp" else 1)


def gather_rows(radiance, scene, dist, tile=None):
    """"rows" sharding: every rank packs the image rows it rendered into one contiguous buffer (padded to the largest share),
    ONE gather to rank 0, which scatters the shares into its frame.  `radiance`: float32 tensor of height*width."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    W, H = int(scene["width"]), int(scene["height"])
    shares = [shard_rows(scene, r, world, tile) for r in range(world)]
    cap = max(len(sh) for sh in shares) * W
    frame = radiance.view(H, W)
    packed = torch.zeros(cap, dtype=radiance.dtype, device=radiance.device)
    mine = torch.tensor(shares[rank], dtype=torch.long, device=radiance.device)
    if len(shares[rank]):
        packed[:len(shares[rank]) * W] = frame.index_select(0, mine).reshape(-1)
    bufs = [torch.zeros_like(packed) for _ in range(world)] if rank == 0 else None
    dist.gather(packed, bufs, dst=0)
    if rank == 0:
        for r in range(1, world):
            if len(shares[r]):
                idx = torch.tensor(shares[r], dtype=torch.long, device=radiance.device)
                frame.index_copy_(0, idx, bufs[r][:len(shares[r]) * W].view(-1, W))
    return cap * radiance.element_size()            # bytes this rank put on the wire


def render_sharded(scene, render_into, radiance, dist=None, mode="spp", sync=None):
    """Renders this rank's share with `render_into(part_scene, radiance)` (which ACCUMULATES into
    `radiance`, a float32 tensor of height*width) and assembles the frame on rank 0: a gather of the
    disjoint tile rows ("rows"), a reduce(sum) of the spp slices ("spp").  `sync()` (e.g.
    torch.cuda.synchronize) is called before the clocks are read.  Returns {"render_s", "collective_s",
    "wire_bytes"} of this rank."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    t0 = time.perf_counter()
    for part in shard_scene(scene, rank, world, mode):
        render_into(part, radiance)
    if sync is not None and world > 1:
        sync()
    t1 = time.perf_counter()
    wire = 0
    if world > 1:
        if mode == "rows":
            wire = gather_rows(radiance, scene, dist)
        else:
            dist.reduce(radiance, dst=0, op=dist.ReduceOp.SUM)
            wire = radiance.numel() * radiance.element_size()
        if sync is not None:
            sync()
    return {"render_s": t1 - t0, "collective_s": time.perf_counter() - t1, "wire_bytes": wire}
