"""bench.py's collective path (RCCL communicator, final all_gather_into_tensor of the pose table, barrier, max-reduced
time, gathered-rows check) rehearsed at world size 1 under torch.distributed.run, and the ONE-JSON-line contract of its
stdout.  N = 2, 4, 8 cannot be run on the one-GPU box; this takes the remaining unknowns out of the N > 1 launch.

Collected first on purpose (the file name sorts before every other test file): the ranks are started as a CHILD process,
which must happen before this process has touched the GPU — it only counts devices, which does not initialise HIP."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_collective_path_at_world_size_1_prints_one_json_line():
    import torch
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU visible")
    if torch.cuda.is_initialized():
        pytest.skip("the GPU was initialised by an earlier test in this process: run this file first (its name sorts first)")
    env = dict(os.environ, POSEHOT_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
           "--pairs", "8", "--cpu-pairs", "0", "--sustained-s", "0"]
    proc = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=110)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # RCCL's banner and every progress line go to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["unit"] == "pairs/s" and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "all_gather" in d["config"]["parallelism"] and d["vs_baseline"] is None
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert "other_configs" not in d                     # N = 1 without the forced communicator only
    # bench.py itself raises SystemExit when the gathered table does not hold this rank's rows or a row lacks STATUS_OK
