"""Strong-scaling rehearsal on ONE GPU: the time of rank r's share of a C1 frame under the "rows" sharding for world sizes
1, 2, 4, 8 (every share rendered in turn on this GPU), against the whole frame.  What it shows: how evenly the interleaved tile
rows split the work and what fixed cost a share carries — the compute side of the N-GPU curve; the gather is not in it.
One JSON object on stdout."""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gpis_pkg  # noqa: E402

pkg = _gpis_pkg.load_package()


def main():
    import torch
    W, H, SPP = 1920, 1080, 64
    med = pkg.Medium(pkg.params_for_config("C1"))
    med.build_guide(16, 64)
    scene = np.array(pkg.default_scene_s(W, H, SPP), dtype=pkg.SCENE_S)
    rad = torch.zeros(H * W, dtype=torch.float32, device="cuda")

    def run(part):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        med.call("gpis_render_scene_s", part.ctypes.data_as(ctypes.c_void_p), rad.data_ptr(), None, None)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(scene); run(scene)                      # first call chunked, second grows the workspace
    whole = min(run(scene) for _ in range(3))
    out = {"whole_frame_ms": whole * 1e3, "worlds": {}}
    for world in (2, 4, 8):
        shares = []
        for r in range(world):
            part = pkg.dist.shard_scene(scene, r, world, "rows")[0]          # tile-row height by dist.rows_tile
            run(part)
            shares.append(min(run(part) for _ in range(2)) * 1e3)
        out["worlds"][str(world)] = {"tile_rows_px": pkg.dist.rows_tile(scene, world), "share_ms": shares, "max_share_ms": max(shares), "ideal_ms": whole * 1e3 / world,
                                     "compute_efficiency": whole * 1e3 / world / max(shares)}
        print(world, out["worlds"][str(world)], file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
