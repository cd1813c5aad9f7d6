"""world_size-2 `gloo` test of the multi-GPU sharding logic on CPU: the same
`scgt_amd.dist.render_sharded` driver bench.py uses, with the oracle renderer injected (the HIP
renderer needs a GPU).  Checks: parameter broadcast, both shardings, the gather of disjoint tile rows
("rows") / reduce ("spp") to rank 0, the per-rank timing record, and that
the sharded image equals the single-process image bit for bit (per-pixel sums of disjoint spp
slices differ only by fp32 association — compared with a tight tolerance for "spp", exactly for
"rows")."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import _gpis_pkg
    import oracle_bindings as ob
    pkg = _gpis_pkg.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = pkg.params_for_config("C0") if rank == 0 else np.zeros((), dtype=pkg.PARAMS)
    params = pkg.dist.broadcast_params(params, pkg.PARAMS, dist)
    assert params["impulse_density"] == 8 and params["seed"] == 7
    orc = ob.Oracle(params, threads=2)
    w, h, spp = 40, 40, 2
    scene = ob.default_scene_s(w, h, spp)
    rad = torch.zeros(h * w, dtype=torch.float32)

    def render_into(part, acc):
        acc += torch.from_numpy(orc.render_scene_s(part).reshape(-1))

    st = pkg.dist.render_sharded(scene, render_into, rad, dist=dist, mode=mode)
    assert st["render_s"] > 0 and st["collective_s"] >= 0
    # "rows": only this rank's rows travel (padded to the largest share: 20 of the 40 rows — the tile rows shrink from 16 pixels
    # until every rank has at least 16 of them, dist.rows_tile); "spp": the whole frame
    assert pkg.dist.rows_tile(scene, world) == 1 and pkg.dist.rows_tile(ob.default_scene_s(1920, 1080, 1), 8) == 8
    assert pkg.dist.rows_tile(ob.default_scene_s(1920, 1080, 1), 2) == 16
    assert st["wire_bytes"] == (20 * w * 4 if mode == "rows" else h * w * 4)
    if rank == 0:
        np.save(out_path, rad.numpy().reshape(h, w) / pkg.dist.total_spp(scene, world, mode))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["spp", "rows"])
def test_two_rank_sharding_matches_single_process(tmp_path, mode, pkg, ob):
    import torch.multiprocessing as mp
    world = 2
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), mode, out), nprocs=world, join=True)
    got = np.load(out)
    orc = ob.Oracle(pkg.params_for_config("C0"), threads=4)
    spp_total = 2 * (world if mode == "spp" else 1)
    want = orc.render_scene_s(ob.default_scene_s(40, 40, spp_total)) / spp_total
    assert want.max() > 0
    if mode == "rows":
        assert np.array_equal(got, want)
    else:
        assert np.allclose(got, want, rtol=1e-6, atol=1e-7)


def test_shard_scene_partitions(pkg, ob):
    scene = ob.default_scene_s(64, 50, 4)
    for world in (1, 2, 3, 8):
        rows = np.zeros(50, dtype=int)
        for r in range(world):
            parts = pkg.dist.shard_scene(scene, r, world, "rows")
            assert len(parts) == 1          # one batch per rank
            if world > 1:
                assert int(parts[0]["shard_index"]) == r and int(parts[0]["shard_count"]) == world
            rows[pkg.dist.shard_rows(scene, r, world)] += 1
        assert (rows == 1).all()
        begins = sorted(int(pkg.dist.shard_scene(scene, r, world, "spp")[0]["spp_begin"]) for r in range(world))
        assert begins == [4 * r for r in range(world)]
