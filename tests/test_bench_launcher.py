"""`python bench.py --gpus N` starts its own rank processes when it is not already under torch.distributed.run.
The launcher (bench.launch_ranks: fresh child processes, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
MASTER_PORT, rank 0 relays the JSON line) is driven here on the CPU: the children run the gloo self-test instead of
the GPU benchmark, everything before that point is the production path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n", [2, 3])
def test_bench_spawns_its_ranks_and_relays_rank0(n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--launcher-selftest"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                 # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["world"] == n and out["selftest"] == n * (n + 1) / 2 and out["backend"] == "gloo"


def test_single_process_path_is_not_spawned():
    import bench
    assert callable(bench.launch_ranks) and callable(bench.launcher_selftest)
