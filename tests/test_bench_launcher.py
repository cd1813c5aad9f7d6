"""`python bench.py --gpus N` must start its own ranks (the driver runs it exactly like that, with no
WORLD_SIZE in the environment).  The launcher is exercised here for real — torch.distributed.run, two rank
processes, gloo — through bench.py's --dry-run mode, which runs the broadcast / tile-row sharding / reduce of
the real job with a pixel-index renderer instead of the GPU kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, size=("48", "40", "2")):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--width", size[0], "--height", size[1], "--spp", size[2]] + extra,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout          # ONE JSON line, from rank 0, relayed by the parent
    return json.loads(lines[0])


def test_bench_starts_its_own_two_ranks_rows():
    res = _run(["--gpus", "2"])
    assert res["dry_run"] and res["ranks_seen"] == 2 and res["n_gpus"] == 2
    assert res["shard"] == "rows" and res["coverage_ok"]      # every pixel rendered exactly once across the two ranks
    assert res["impulse_density"] == 32.0                     # rank 1 received the C1 parameter block by broadcast


def test_bench_three_ranks_spp_and_single():
    res = _run(["--gpus", "3", "--shard", "spp"])
    assert res["ranks_seen"] == 3 and res["coverage_ok"]
    one = _run([])
    assert one["ranks_seen"] == 1 and one["coverage_ok"]


def test_c3_eight_gpu_geometry():
    """BASELINE.json configs[3] as the driver will launch it — 8 ranks, 3840x2160, 256 spp, multi-resolution medium — through the
    real launcher, broadcast, tile-row sharding and gather (gloo, no GPU work): every pixel exactly once, rank 1 received the C3
    parameter block, and the per-rank plan (rows, samples, chunks of at most 2^27 samples, workspace, bytes on the wire)."""
    res = _run(["--gpus", "8", "--config", "C3"], size=("3840", "2160", "256"))
    assert res["ranks_seen"] == 8 and res["coverage_ok"] and res["impulse_density"] == 64.0
    plan = res["plan"]
    assert plan["samples_total"] == 3840 * 2160 * 256 == 2123366400
    assert sum(plan["rows_per_rank"]) == 2160 and max(plan["rows_per_rank"]) - min(plan["rows_per_rank"]) <= 16      # 135 tile rows over 8 ranks
    assert all(c == 2 for c in plan["chunks_per_rank"])                      # ~265 M samples per rank: two chunks of <= 2^27
    assert max(plan["workspace_bytes_per_rank"]) <= 60e9                     # 375 B x 2^27 samples = 50 GB of the 288
    # the gather moves each rank's own rows only: 1/8 of the 33 MB frame (padded to the largest share)
    assert all(b == max(plan["rows_per_rank"]) * 3840 * 4 for b in plan["wire_bytes_per_rank"])

