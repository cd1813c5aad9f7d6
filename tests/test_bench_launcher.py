"""`python bench.py --gpus N` must start its own ranks (the driver runs it exactly like that, with no
WORLD_SIZE in the environment).  The launcher is exercised here for real — torch.distributed.run, two rank
processes, gloo — through bench.py's --dry-run mode, which runs the broadcast / tile-row sharding / reduce of
the real job with a pixel-index renderer instead of the GPU kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--width", "48", "--height", "40", "--spp", "2"] + extra,
                       env=env, capture_output=True, text=True, timeout=600)
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
