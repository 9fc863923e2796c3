"""bench.py's N > 1 path, rehearsed on one GPU: two ranks launched by torch.distributed.run exactly as the
driver launches them, both on GPU 0 and over gloo (YAK_BENCH_SHARE_GPU=1; RCCL refuses two ranks on one
device).  Checks the contract of the one JSON line: whole-job value, max-over-ranks time, weak scaling.
And the RCCL branch itself (init_process_group("nccl", device_id=...), barrier, all_reduce of CUDA tensors) with
the one rank a one-GPU box allows."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_through_torch_distributed_run():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, YAK_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                                   # rank 0 prints, once
    d = json.loads(lines[0])
    assert d["metric"] == "voxel updates/sec" and d["unit"] == "Gvox/s" and d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["steps"] == 20 and d["warmup"] == 3 and d["vs_baseline"] is None
    assert d["config"]["voxels_updated_per_step"] == 2 * 512 ** 3              # both slabs, every voxel updates
    assert "1024x512x512" in d["config"]["workload"] and d["config"]["parallelism"] == "z-slab x2"
    # value = all ranks' voxels over the slowest rank's time
    assert abs(d["value"] - 2 * 512 ** 3 / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * d["value"]
    assert "cpu_baseline" not in d and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    # the same command carries north_star's scaling targets: config 5's sweep per rank ...
    c5 = d["c5"]
    assert "error" not in c5, c5
    assert c5["metric"] == "voxel updates/sec" and c5["n_gpus"] == 2 and c5["config"]["frames_per_step"] == 4
    assert "2048x2048x512" in c5["config"]["workload"] and c5["value"] > 0 and c5["marching_cubes"]["triangles"] > 0
    assert abs(c5["value"] - c5["config"]["voxel_updates_per_step"] / (c5["ms_per_step"] * 1e-3) / 1e9) < 0.01 * c5["value"]
    # ... and config 4, the sharded tracking loop: frames per second, stage times, the frame's collectives on their own
    lp = d["kinfu_loop_sharded"]
    assert "error" not in lp, lp
    assert "1024^3" in lp["workload"] and lp["frames"] == 60 and lp["frames_fused"] == 59 and lp["fused_frames_per_s"] > 1
    assert set(lp["stage_ms_slowest_rank"]) == {"preprocess", "icp", "integrate", "raycast"}
    assert all(v > 0 for v in lp["stage_ms_slowest_rank"].values())
    assert set(lp["collectives_of_a_frame_on_their_own_ms"]) == {"icp_19_x_33_floats_ms", "raycast_keys_min_ms", "raycast_payload_sum_ms",
                                                                  "raycast_settled_sum_ms", "sum"}
    assert lp["position_error_last_frame_mm"] < 30.0 and "gloo" in lp["transport"]


def test_one_rank_over_rccl():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, YAK_BENCH_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3",
           "--no-rows", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["voxels_updated_per_step"] == 512 ** 3
    assert abs(d["value"] - 512 ** 3 / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * d["value"]


def test_cpp_loop_row_of_the_bench_line():
    """bench.py's kinfu_loop row also reports the C++ host (examples/offline_fuse as a child process): the example must
    be built and its output must parse."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    row = bench.cpp_loop_rates(120)
    assert row is not None, "examples/offline_fuse missing or its output changed"
    for key in ("fused_frames_per_s", "fused_frames_per_s_supplied_poses", "fused_frames_per_s_after_the_first_50",
                "fused_frames_per_s_supplied_poses_after_the_first_50"):
        assert row[key] > 100, (key, row)
