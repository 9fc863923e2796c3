// offline_fuse -- the in-repo caller of the Yak API (what yak_ros does per depth image and per
// "extract mesh" service call): feed depth frames into yak::FusionServer::fuse with known camera poses (or
// let kfusion::KinFu track them from depth alone), downloadTSDF, marchingCubesCPU, write a PLY.
// Frames are rendered analytically (scene S2 of the benchmarks: a ground plane and three spheres seen from
// an orbit), so the program needs no data files.  Links only the product libraries: no oracle, no torch.
//
//   offline_fuse [--n 256] [--frames 60] [--track] [--ply out.ply]
#include <yak/kfusion/cuda/imgproc.hpp>
#include <yak/mc/marching_cubes.h>
#include <yak/yak_server.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

using kfusion::Affine3f;
using kfusion::Vec3f;

// camera -> volume pose looking from `eye` at `target`, +y down (the convention of depth cameras)
static Affine3f look_at(const double eye[3], const double target[3])
{
  double f[3] = { target[0] - eye[0], target[1] - eye[1], target[2] - eye[2] };
  const double fn = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  for (double& v : f)
    v /= fn;
  double r[3] = { f[2], 0.0, -f[0] };  // (0, 1, 0) x f
  const double rn = std::sqrt(r[0] * r[0] + r[2] * r[2]);
  for (double& v : r)
    v /= rn;
  const double d[3] = { f[1] * r[2] - f[2] * r[1], f[2] * r[0] - f[0] * r[2], f[0] * r[1] - f[1] * r[0] };
  kfusion::Mat3f R;
  for (int i = 0; i < 3; ++i)
    R(i, 0) = (float)r[i], R(i, 1) = (float)d[i], R(i, 2) = (float)f[i];
  return Affine3f(R, Vec3f((float)eye[0], (float)eye[1], (float)eye[2]));
}

static Affine3f orbit_pose(int frame)
{
  const double ang = 1.2 * frame * 3.14159265358979323846 / 180.0, centre[3] = { 1.5, 1.5, 1.5 };
  const double eye[3] = { 1.5 + 3.2 * std::sin(ang), 1.5, 1.5 - 3.2 * std::cos(ang) };
  return look_at(eye, centre);
}

// depth in millimetres of the ground plane y = 1.95 and three spheres
static std::vector<std::uint16_t> render(const Affine3f& cam, const kfusion::Intr& in, int rows, int cols)
{
  static const double sph[3][4] = { { 1.1, 1.45, 1.2, 0.5 }, { 2.0, 1.6, 1.4, 0.35 }, { 1.5, 1.7, 2.1, 0.25 } };
  std::vector<std::uint16_t> depth((size_t)rows * cols, 0);
  const kfusion::Mat3f R = cam.rotation();
  const Vec3f o = cam.translation();
  for (int v = 0; v < rows; ++v)
    for (int u = 0; u < cols; ++u)
    {
      const double dc[3] = { (u - in.cx) / in.fx, (v - in.cy) / in.fy, 1.0 };
      double d[3];
      for (int i = 0; i < 3; ++i)
        d[i] = R(i, 0) * dc[0] + R(i, 1) * dc[1] + R(i, 2) * dc[2];
      double best = 1e30;
      if (std::fabs(d[1]) > 1e-12 && (1.95 - o[1]) / d[1] > 1e-6)
        best = (1.95 - o[1]) / d[1];
      for (const auto& sp : sph)
      {
        const double oc[3] = { o[0] - sp[0], o[1] - sp[1], o[2] - sp[2] };
        const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], b = 2 * (d[0] * oc[0] + d[1] * oc[1] + d[2] * oc[2]),
                     c = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - sp[3] * sp[3], disc = b * b - 4 * a * c;
        if (disc >= 0)
        {
          const double s = (-b - std::sqrt(disc)) / (2 * a);
          if (s > 1e-6 && s < best)
            best = s;
        }
      }
      const double mm = std::floor(best * 1000.0 + 0.5);  // the ray parameter is the depth: dc.z = 1
      depth[(size_t)v * cols + u] = (best < 1e29 && mm > 0 && mm < 65535) ? (std::uint16_t)mm : 0;
    }
  return depth;
}

static double now()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
  int n = 256, frames = 60;
  bool track = false;
  std::string ply;
  for (int i = 1; i < argc; ++i)
  {
    if (!std::strcmp(argv[i], "--n") && i + 1 < argc)
      n = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--frames") && i + 1 < argc)
      frames = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--track"))
      track = true;
    else if (!std::strcmp(argv[i], "--ply") && i + 1 < argc)
      ply = argv[++i];
    else
    {
      std::fprintf(stderr, "usage: offline_fuse [--n 256] [--frames 60] [--track] [--ply out.ply]\n");
      return 2;
    }
  }
  try
  {
    kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
    p.volume_dims = kfusion::Vec3i::all(n);
    p.volume_resolution = 3.f / (float)n;
    p.volume_pose = orbit_pose(0);  // where frame 0 is integrated (kinfu.cpp:193)

    std::vector<std::vector<std::uint16_t>> depth;
    for (int f = 0; f < frames; ++f)
      depth.push_back(render(orbit_pose(f), p.intr, p.rows, p.cols));

    yak::MarchingCubesParameters mc;
    mc.scale = p.volume_resolution;
    mc.min_weight = 2;
    mc.clean = true;

    yak::TSDFContainer volume(1, 1, 1);
    yak::PolygonMesh direct;  // the mesh taken from the device volume, without the round trip through the host
    double t0 = 0, t_loop = 0, direct_ms = -1, t_down = 0, t_warm = 0;
    const int warm = frames > 100 ? 50 : 0;  // frames that carry the first launches and allocations
    double worst_mm = 0;
    if (track)
    {  // what a hand-held sensor does: no poses, KinFu tracks by ICP against the ray-cast model
      kfusion::KinFu kinfu(p);
      kfusion::cuda::Depth d;
      t0 = now();
      for (int f = 0; f < frames; ++f)
      {
        if (f == warm)
          t_warm = now();
        kinfu.beginFrame();
        d.upload(depth[f].data(), (size_t)p.cols * 2, p.rows, p.cols);
        const bool ok = kinfu(Affine3f::Identity(), Affine3f::Identity(), Affine3f::Identity(), d);
        if (!ok && f > 0)
          std::fprintf(stderr, "frame %d: tracking lost, volume reset\n", f);
        const Affine3f err = orbit_pose(f).inv() * kinfu.getCameraPose();
        worst_mm = std::fmax(worst_mm, 1e3 * yak::compat::norm(err.translation()));
      }
      kfusion::cuda::waitAllDefaultStream();  // (operator() returns before the device is done)
      t_loop = now();
      volume = yak::TSDFContainer(n, n, n);
      kinfu.tsdf().data().download(volume.data());
      t_down = now();
    }
    else
    {  // what yak_ros does: poses come from the robot (TF), FusionServer::fuse per depth image
      p.use_icp = false;
      p.use_pose_hints = true;
      p.update_via_sensor_motion = false;
      yak::FusionServer server(p, Eigen::Affine3f::Identity());
      t0 = now();
      for (int f = 0; f < frames; ++f)
      {
        if (f == warm)
          t_warm = now();
        const cv::Mat m(p.rows, p.cols, depth[f].data(), (size_t)p.cols * 2);
        server.fuse(m, Eigen::Affine3f(orbit_pose(f)));
      }
      kfusion::cuda::waitAllDefaultStream();
      t_loop = now();
      volume = server.downloadTSDF();
      t_down = now();
      direct = server.extractMesh(mc);
      direct_ms = (now() - t_down) * 1e3;
    }
    const double t1 = now();

    const yak::PolygonMesh mesh = yak::marchingCubesCPU(volume, mc);
    const double t2 = now();

    std::printf("%d^3 volume, %d frames (%s): %.1f ms = %.0f frames/s incl. the upload of every frame (%.0f frames/s after "
                "the first %d); download %.1f ms; mesh %.1f ms: %zu vertices, %zu triangles\n",
                n, frames, track ? "tracked by ICP" : "poses supplied", (t_loop - t0) * 1e3, frames / (t_loop - t0),
                (frames - warm) / (t_loop - t_warm), warm,
                (t_down - t_loop) * 1e3, (t2 - t1) * 1e3, mesh.cloud.size(), mesh.polygons.size());
    if (direct_ms >= 0)
    {
      const bool same = direct.cloud == mesh.cloud && direct.polygons == mesh.polygons;
      std::printf("FusionServer::extractMesh (from the device volume): %.1f ms, %s\n", direct_ms,
                  same ? "the same mesh" : "MESH DIFFERS");
      if (!same)
        return 2;
    }
    if (track)
      std::printf("worst position error against the true orbit: %.1f mm (voxel %.1f mm)\n", worst_mm, 3000.0 / n);
    if (!ply.empty())
    {
      FILE* f = std::fopen(ply.c_str(), "w");
      if (!f)
        throw std::runtime_error("cannot write " + ply);
      std::fprintf(f, "ply\nformat ascii 1.0\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\n"
                      "element face %zu\nproperty list uchar int vertex_indices\nend_header\n",
                   mesh.cloud.size(), mesh.polygons.size());
      for (const auto& v : mesh.cloud)
        std::fprintf(f, "%.6f %.6f %.6f\n", v[0], v[1], v[2]);
      for (const auto& t : mesh.polygons)
        std::fprintf(f, "3 %u %u %u\n", t[0], t[1], t[2]);
      std::fclose(f);
    }
    return mesh.polygons.empty() ? 1 : 0;
  }
  catch (const std::exception& e)
  {
    std::fprintf(stderr, "offline_fuse: %s\n", e.what());
    return 1;
  }
}
