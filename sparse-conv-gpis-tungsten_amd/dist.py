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
          disjoint, so the same reduce(sum) assembles them (a gather of disjoint tiles expressed as
          a sum of zeros).

Both need exactly two collectives per job: the broadcast of the POD parameter block from rank 0
and the final reduce.  `render_fn(scene_record) -> radiance tensor/array` is injected so that the
CPU tests can drive the same logic with a CPU renderer.
"""
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


def shard_scene(scene, rank, world, mode="spp", tile=16):
    """The list of scene records (row ranges / spp slices) this rank renders."""
    scene = np.array(scene)
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


def shard_rows(scene, rank, world, tile=16):
    """Image rows of rank `rank` under the "rows" sharding (for tests and bookkeeping)."""
    y0, n = int(scene["y_begin"]), int(scene["y_count"])
    return [y0 + y for y in range(n) if world <= 1 or (y // tile) % world == rank]


def total_spp(scene, world, mode):
    return int(scene["spp_count"]) * (world if mode == "spp" else 1)


def render_sharded(scene, render_into, radiance, dist=None, mode="spp"):
    """Renders this rank's share with `render_into(part_scene, radiance)` (which ACCUMULATES into
    `radiance`, a float32 tensor of height*width) and reduces the sums to rank 0."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    for part in shard_scene(scene, rank, world, mode):
        render_into(part, radiance)
    if world > 1:
        dist.reduce(radiance, dst=0, op=dist.ReduceOp.SUM)
    return radiance
