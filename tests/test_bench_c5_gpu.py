"""bench.py --config c5 (BASELINE config 5, north_star's scaling target: 2048^3 over 8 GPUs, 4 cameras per sweep +
GPU marching cubes): the JSON contract, rehearsed on one GPU -- one process playing rank 3 of 8, and two ranks
launched by torch.distributed.run as the driver launches them (both on GPU 0, gloo)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _one_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def _check(d, world, steps, warmup):
    assert d["metric"] == "voxel updates/sec" and d["unit"] == "Gvox/s" and d["scaling"] == "weak"
    assert d["steps"] == steps and d["warmup"] == warmup and d["vs_baseline"] is None and d["higher_is_better"] is True
    c = d["config"]
    assert "config 5" in c["workload"] and c["frames_per_step"] == 4 and c["parallelism"] == "z-slab x%d" % world
    assert c["voxels_visited_per_step"] == 4 * 2048 * 2048 * 256 * world
    assert 0 < c["voxels_touched_per_step"] <= c["voxel_updates_per_step"] <= 4 * c["voxels_touched_per_step"]
    assert abs(d["value"] - c["voxel_updates_per_step"] / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["kernel_ms"] <= d["ms_per_step"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert d["marching_cubes"]["triangles"] > 100000 and d["marching_cubes"]["ms_per_extraction_slowest_rank"] > 0


def test_one_gpu_plays_rank_3_of_8():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c5", "--as-rank", "8:3", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 1 and "3 of 8" in d["config"]["emulated_rank"]
    # (an emulated rank reports its own slab only: totals are per GPU)
    assert d["config"]["voxels_touched_per_step"] > 10 ** 8 and "cpu_baseline" not in d


def test_two_ranks_through_torch_distributed_run():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, YAK_BENCH_SHARE_GPU="1", YAK_BENCH_CONFIG="c5")  # the driver cannot pass --config: the environment can
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    _check(d, 2, 3, 1)
