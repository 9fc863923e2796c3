"""The C++ host layer (include/yak/..., yak_amd/host/): the reference's class names over the C ABI.
The cases live in tests/cpp/test_host.cpp (built by __graft_entry__.build()); this file runs them."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host")


def run(mode):
    assert os.path.exists(EXE), "tests/cpp/test_host missing: run __graft_entry__.build()"
    r = subprocess.run([EXE, mode], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    print(r.stderr)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout[-2000:]


def test_host_layer_cpu_cases():
    run("--cpu")


def test_closed_form_accumulation_equals_the_loop():
    """yak_amd/csrc/ffwd.hpp (n rounded fp32 additions in O(binades crossed)) against the loop, 2.5 M cases."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_ffwd")
    assert os.path.exists(exe), "tests/cpp/test_ffwd missing: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout[-3000:]


def test_frontend_library_exports_reference_class_names():
    lib = os.path.join(ROOT, "yak_amd", "host", "libyak_frontend.so")
    assert os.path.exists(lib)
    syms = subprocess.run(["nm", "-DC", "--defined-only", lib], capture_output=True, text=True).stdout
    for name in ["yak::FusionServer::fuse(", "yak::FusionServer::downloadTSDF()", "kfusion::KinFu::operator()(",
                 "kfusion::KinFuParams::default_params()", "kfusion::cuda::TsdfVolume::integrate(",
                 "kfusion::cuda::TsdfVolume::clear()", "kfusion::cuda::computeDists(",
                 "kfusion::cuda::DeviceMemory2D::upload(", "kfusion::cuda::getCudaEnabledDeviceCount()",
                 "yak::TSDFContainer::TSDFContainer(int, int, int)", "yak::marchingCubesCPU(",
                 "kfusion::cuda::TsdfVolume::raycast(", "kfusion::cuda::ProjectiveICP::estimateTransform(",
                 "kfusion::cuda::depthBilateralFilter(", "kfusion::cuda::computePointNormals(",
                 "kfusion::ShardedKinFu::ShardedKinFu(", "kfusion::ShardedKinFu::downloadVolume(",
                 "kfusion::cuda::setShard("]:
        assert name in syms, name


@pytest.mark.gpu
def test_host_layer_gpu_fusion_server_matches_oracle():
    run("--gpu")


@pytest.mark.gpu
def test_host_layer_gpu_sharded_kinfu_matches_the_single_gpu_loop():
    """kfusion::ShardedKinFu (C++ threads as ranks, collectives inside the C ABI) against kfusion::KinFu."""
    run("--gpu-sharded")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [[], ["--track"]])
def test_offline_fuse_example(tmp_path, mode):
    """examples/offline_fuse: FusionServer::fuse with supplied poses (what yak_ros does) or KinFu tracking from
    depth alone, then downloadTSDF -> marchingCubesCPU -> PLY."""
    exe = os.path.join(ROOT, "examples", "offline_fuse")
    assert os.path.exists(exe), "examples/offline_fuse missing: run __graft_entry__.build()"
    ply = str(tmp_path / "mesh.ply")
    r = subprocess.run([exe, "--n", "96", "--frames", "10", "--ply", ply] + mode, capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stderr[-2000:]
    head = open(ply).read(400).split("\n")
    n_vert, n_face = int(head[2].split()[-1]), int(head[6].split()[-1])
    assert n_vert > 1000 and n_face > 2000
    if mode:
        worst = float(r.stdout.split("true orbit:")[1].split("mm")[0])
        assert worst < 0.5 * 3000.0 / 96                                         # half a voxel
