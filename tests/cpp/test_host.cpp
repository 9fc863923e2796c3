// Host-layer tests, in the style the reference tests its own host code (gtest-like cases, no gtest
// in this image): `test_host --cpu` needs no GPU; `test_host --gpu` drives yak::FusionServer on a
// real device and checks the downloaded volume word for word against the CPU oracle.
#include <yak/kfusion/cuda/imgproc.hpp>
#include <yak/kfusion/kinfu_sharded.hpp>
#include <yak/mc/marching_cubes.h>
#include <yak/yak_server.h>

#include <yak_oracle.h>
#include <yakhip.h>

#include "../../yak_amd/csrc/solve6.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

static int g_failed = 0;
#define EXPECT(cond)                                                         \
  do                                                                         \
  {                                                                          \
    if (!(cond))                                                             \
    {                                                                        \
      std::printf("  FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);          \
      ++g_failed;                                                            \
    }                                                                        \
  } while (0)

using kfusion::Affine3f;
using kfusion::Vec3f;

static Affine3f rot_y_translate(float angle, float tx, float ty, float tz)
{
  kfusion::Mat3f R = kfusion::Mat3f::eye();
  R(0, 0) = std::cos(angle);
  R(0, 2) = std::sin(angle);
  R(2, 0) = -std::sin(angle);
  R(2, 2) = std::cos(angle);
  return Affine3f(R, Vec3f(tx, ty, tz));
}

// ---------------------------------------------------------------------------------------- CPU
static void test_params_defaults()  // reference kinfu.cpp:10-48
{
  std::printf("KinFuParams.default_params\n");
  const kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
  EXPECT(p.cols == 640 && p.rows == 480);
  EXPECT(p.intr.fx == 580.f && p.intr.fy == 580.f && p.intr.cx == 319.5f && p.intr.cy == 239.5f);
  EXPECT(p.volume_dims[0] == 512 && p.volume_dims[1] == 512 && p.volume_dims[2] == 512);
  EXPECT(p.volume_resolution == 0.005859375f);
  EXPECT(p.tsdf_trunc_dist == 0.04f && p.tsdf_max_weight == 64 && p.tsdf_min_camera_movement == 0.f);
  EXPECT(p.icp_iter_num.size() == 4 && p.icp_iter_num[0] == 10 && p.icp_iter_num[3] == 0);
  EXPECT(p.raycast_step_factor == 0.75f && p.gradient_delta_factor == 0.5f);
  EXPECT(p.use_icp && !p.use_pose_hints && p.update_via_sensor_motion);
  const kfusion::Intr l2 = p.intr(2);  // precomp.cpp:10-14
  EXPECT(l2.fx == 145.f && l2.cx == 319.5f / 4);
}

static void test_affine_math()
{
  std::printf("compat.Affine3f\n");
  const Affine3f a = rot_y_translate(0.3f, 0.5f, -1.25f, 2.f), b = rot_y_translate(-1.1f, 3.f, 0.25f, -0.5f);
  const Affine3f id = a * a.inv();
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c)
      EXPECT(std::fabs(id.matrix(r, c) - (r == c ? 1.f : 0.f)) < 1e-6f);
  const Vec3f v(0.2f, 0.4f, -0.7f);
  const Vec3f l = (a * b) * v, r2 = a * (b * v);
  EXPECT(std::fabs(l[0] - r2[0]) < 1e-5f && std::fabs(l[1] - r2[1]) < 1e-5f && std::fabs(l[2] - r2[2]) < 1e-5f);
  const Vec3f rv = a.rvec();  // rotation about +y by 0.3 rad
  EXPECT(std::fabs(rv[0]) < 1e-6f && std::fabs(rv[1] - 0.3f) < 1e-6f && std::fabs(rv[2]) < 1e-6f);
  const Affine3f back(rv, a.translation());  // Rodrigues round trip
  for (int i = 0; i < 9; ++i)
    EXPECT(std::fabs(back.rotation().val[i] - a.rotation().val[i]) < 1e-6f);
}

static void test_container_and_half()  // reference test/marching_cubes_tests.cpp:63-79, tsdf_container.h:21-49
{
  std::printf("TSDFContainer / Entry\n");
  yak::TSDFContainer c(3, 4, 5);
  EXPECT(c.size() == 60 && c.toIndex(1, 2, 3) == 3 * 4 * 3 + 3 * 2 + 1 && c.toIndex64(1, 2, 3) == 43u);
  int x, y, z;
  c.fromIndex(43, x, y, z);
  EXPECT(x == 1 && y == 2 && z == 3);
  c.data()[43] = 0x00133600u;  // half(0.375), weight 19
  std::uint16_t d, w;
  c.read(43, d, w);
  EXPECT(d == 0x3600 && w == 19 && c.distance(43) == 0.375f);
  typedef kfusion::cuda::TsdfVolume::Entry E;
  EXPECT(E::float2half(0.375f) == 0x3600 && E::half2float(0x3600) == 0.375f);
  for (unsigned h = 0; h < 65536; ++h)  // against the oracle's conversions
  {
    const float f = E::half2float((E::half)h);
    if (f == f)
    {
      const float o = yko_half2float((uint16_t)h);
      EXPECT(std::memcmp(&f, &o, 4) == 0);
      EXPECT(E::float2half(f) == h);
    }
  }
  kfusion::cuda::ProjectiveICP icp;  // levels with a non-zero iteration count: {10,5,4,0} -> 3
  EXPECT(icp.getUsedLevelsNum() == 3);
}

static void test_solve6()  // stands in for cv::determinant / cv::solve(DECOMP_SVD), projective_icp.cpp:215,225
{
  std::printf("solve6: determinant and SVD solve of the 6 x 6 normal equations\n");
  // A = M^T M + I for a fixed M: symmetric positive definite
  double M[36], A[36], xs[6], b[6], x[6];
  for (int i = 0; i < 36; ++i)
    M[i] = std::sin(1.0 + 0.7 * i) + (i % 7 == 0 ? 2.0 : 0.0);
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c)
    {
      double s = r == c ? 1.0 : 0.0;
      for (int k = 0; k < 6; ++k)
        s += M[k * 6 + r] * M[k * 6 + c];
      A[r * 6 + c] = s;
    }
  for (int i = 0; i < 6; ++i)
    xs[i] = 0.25 * (i - 2.5);
  for (int r = 0; r < 6; ++r)
  {
    b[r] = 0;
    for (int c = 0; c < 6; ++c)
      b[r] += A[r * 6 + c] * xs[c];
  }
  yak::host::solve6_svd_symmetric(A, b, x);
  for (int i = 0; i < 6; ++i)
    EXPECT(std::fabs(x[i] - xs[i]) < 1e-10);
  // determinant: diagonal, a row swap, and det(A) = prod(eigenvalues) > 0 for the SPD matrix
  double D[36] = { 0 };
  for (int i = 0; i < 6; ++i)
    D[i * 7] = i + 1.0;
  EXPECT(std::fabs(yak::host::det6(D) - 720.0) < 1e-9);
  for (int c = 0; c < 6; ++c)
  {
    const double t = D[c];
    D[c] = D[6 + c];
    D[6 + c] = t;
  }
  EXPECT(std::fabs(yak::host::det6(D) + 720.0) < 1e-9);
  EXPECT(yak::host::det6(A) > 1.0);
  // rank-deficient system: the SVD solve returns the minimum-norm solution (no component along the null vector)
  double P[36] = { 0 }, rhs[6] = { 1, 2, 3, 4, 5, 0 };
  for (int i = 0; i < 5; ++i)
    P[i * 7] = 2.0;  // last row / column zero
  yak::host::solve6_svd_symmetric(P, rhs, x);
  for (int i = 0; i < 5; ++i)
    EXPECT(std::fabs(x[i] - rhs[i] / 2.0) < 1e-12);
  EXPECT(x[5] == 0.0 && yak::host::det6(P) == 0.0);

  kfusion::cuda::ProjectiveICP icp;  // projective_icp.cpp:87-104
  icp.setIterationsNum(std::vector<int>{ 3, 2 });
  EXPECT(icp.getUsedLevelsNum() == 2);
  icp.setIterationsNum(std::vector<int>{ 1, 0, 0, 7, 9, 9 });  // cut to MAX_PYRAMID_LEVELS
  EXPECT(icp.getUsedLevelsNum() == 4);
  icp.setIterationsNum(std::vector<int>{});
  EXPECT(icp.getUsedLevelsNum() == 0);
}

static void test_no_gpu_is_an_exception()
{
  std::printf("no GPU -> std::runtime_error (never exit)\n");
  if (kfusion::cuda::getCudaEnabledDeviceCount() > 0)
  {
    std::printf("  (a GPU is present, skipped)\n");
    return;
  }
  bool threw = false;
  try
  {
    kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
    p.volume_dims = kfusion::Vec3i::all(8);
    kfusion::KinFu k(p);
  }
  catch (const std::runtime_error&)
  {
    threw = true;
  }
  EXPECT(threw);
}

// ---------------------------------------------------------------------------------------- GPU
// analytic depth frame: wall z_world = 3.4 and a sphere, seen from camera pose `cam` (camera -> volume)
static std::vector<std::uint16_t> render(const Affine3f& cam, const kfusion::Intr& in, int rows, int cols)
{
  std::vector<std::uint16_t> depth((size_t)rows * cols, 0);
  const kfusion::Mat3f R = cam.rotation();
  const Vec3f o = cam.translation();
  const double cx = 1.5, cy = 1.5, cz = 1.6, rad = 0.8;
  for (int v = 0; v < rows; ++v)
    for (int u = 0; u < cols; ++u)
    {
      const double dc[3] = { (u - in.cx) / in.fx, (v - in.cy) / in.fy, 1.0 };
      double d[3];
      for (int i = 0; i < 3; ++i)
        d[i] = R(i, 0) * dc[0] + R(i, 1) * dc[1] + R(i, 2) * dc[2];
      double best = 1e30;
      if (std::fabs(d[2]) > 1e-12)
      {
        const double s = (3.4 - o[2]) / d[2];
        if (s > 1e-6)
          best = s;
      }
      const double oc[3] = { o[0] - cx, o[1] - cy, o[2] - cz };
      const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], b = 2 * (d[0] * oc[0] + d[1] * oc[1] + d[2] * oc[2]),
                   c = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - rad * rad, disc = b * b - 4 * a * c;
      if (disc >= 0)
      {
        const double s = (-b - std::sqrt(disc)) / (2 * a);
        if (s > 1e-6 && s < best)
          best = s;
      }
      const double mm = std::floor(best * 1000.0 + 0.5);
      depth[(size_t)v * cols + u] = (best < 1e29 && mm > 0 && mm < 65535) ? (std::uint16_t)mm : 0;
    }
  return depth;
}

static void oracle_frame(std::vector<std::uint32_t>& vol, const kfusion::cuda::TsdfVolume& tv, const Affine3f& cam_pose,
                         const std::vector<std::uint16_t>& depth, const kfusion::KinFuParams& p)
{
  std::vector<std::uint16_t> dists(depth.size());
  yko_compute_dists(depth.data(), p.cols * 2, dists.data(), p.cols * 2, p.rows, p.cols, p.intr.fx, p.intr.fy, p.intr.cx,
                    p.intr.cy);
  const Affine3f vol2cam = cam_pose.inv() * tv.getPose();  // tsdf_volume.cpp:97, same host arithmetic
  const kfusion::Vec3i dims = tv.getDims();
  const Vec3f vs = tv.getVoxelSize();
  const int d[3] = { dims[0], dims[1], dims[2] };
  const float vsz[3] = { vs[0], vs[1], vs[2] }, t[3] = { vol2cam.translation()[0], vol2cam.translation()[1],
                                                          vol2cam.translation()[2] };
  const float intr[4] = { p.intr.fx, p.intr.fy, p.intr.cx, p.intr.cy };
  yko_integrate(vol.data(), d, 0, d[2], vsz, tv.getTruncDist(), tv.getMaxWeight(), dists.data(), p.cols * 2, p.rows,
                p.cols, vol2cam.rotation().val, t, intr, nullptr, 8);
}

static void test_fusion_server_against_oracle()
{
  std::printf("FusionServer.fuse / downloadTSDF vs oracle\n");
  kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
  p.volume_dims = kfusion::Vec3i(64, 48, 56);
  p.volume_resolution = 3.f / 64.f;
  p.use_icp = false;
  p.use_pose_hints = true;
  p.update_via_sensor_motion = false;
  yak::FusionServer server(p, Eigen::Affine3f::Identity());

  std::vector<std::uint32_t> want((size_t)64 * 48 * 56, 0u);
  kfusion::KinFu probe(p);  // only to read the volume parameters the host derives
  // trunc clamp (tsdf_volume.cpp:62-67) is sticky through the constructor order (kinfu.cpp:52-61): the
  // volume is born 3 m wide, so with 48 voxels in y the first clamp is 2.1 * 3/48, and the later
  // setSize() re-clamps that value, not the requested 0.04
  EXPECT(std::fabs(probe.tsdf().getTruncDist() - 2.1f * (3.f / 48.f)) < 1e-7f);
  EXPECT(probe.tsdf().getMaxWeight() == 64);

  const Affine3f poses[4] = { rot_y_translate(0.f, 1.5f, 1.2f, -2.0f), rot_y_translate(0.1f, 1.1f, 1.25f, -2.1f),
                              rot_y_translate(-0.15f, 1.9f, 1.1f, -1.9f), rot_y_translate(0.25f, 0.8f, 1.3f, -2.2f) };
  for (int f = 0; f < 4; ++f)
  {
    std::vector<std::uint16_t> depth = render(poses[f], p.intr, p.rows, p.cols);
    const cv::Mat m(p.rows, p.cols, depth.data(), (size_t)p.cols * 2);
    const bool ok = server.fuse(m, Eigen::Affine3f(poses[f]));
    EXPECT(ok == (f > 0));  // frame 0 integrates but reports false (kinfu.cpp:200)
    // frame 0 integrates at params.volume_pose, not at the pose passed in (kinfu.cpp:193)
    oracle_frame(want, probe.tsdf(), f == 0 ? p.volume_pose : poses[f], depth, p);
    const yak::TSDFContainer got = server.downloadTSDF();
    EXPECT(got.size64() == want.size());
    size_t diff = 0, touched = 0;
    for (size_t i = 0; i < want.size(); ++i)
    {
      diff += got.data()[i] != want[i];
      touched += want[i] != 0;
    }
    std::printf("  frame %d: %zu voxels set, %zu differ\n", f, touched, diff);
    EXPECT(diff == 0);
    EXPECT(f == 0 || touched > 1000);
  }
  const yak::TSDFContainer saved = server.downloadTSDF();
  EXPECT(server.reset());
  const yak::TSDFContainer cleared = server.downloadTSDF();
  size_t nz = 0;
  for (size_t i = 0; i < cleared.size64(); ++i)
    nz += cleared.data()[i] != 0;
  EXPECT(nz == 0);
  // uploadTSDF puts a saved volume back, word for word; a container of another size is refused
  EXPECT(server.uploadTSDF(saved));
  const yak::TSDFContainer back = server.downloadTSDF();
  EXPECT(back.size64() == saved.size64() && std::memcmp(back.data(), saved.data(), saved.size64() * 4) == 0);
  EXPECT(!server.uploadTSDF(yak::TSDFContainer(8, 8, 8)));
  // A volume written behind the class's back (the non-const data() handle marks its surface map stale): fetchCloud --
  // a const member, it cannot recompute the map -- must read the voxels, not trust the map of the empty volume.
  {
    probe.tsdf().data().upload(saved.data(), saved.size64() * sizeof(std::uint32_t));
    kfusion::cuda::DeviceArray<kfusion::Point> buffer;
    const kfusion::cuda::TsdfVolume& tv = probe.tsdf();
    const std::size_t got = tv.fetchCloud(buffer).size();
    // tsdf_volume.cu:524-612: a point per voxel pair (x, y, z) / +x, +y, +z with weights != 0, tsdf != 1, opposite strict signs
    const int X = 64, Y = 48, Z = 56;
    std::size_t pairs = 0;
    auto at = [&](int x, int y, int z, float& F) {
      std::uint16_t h, w;
      const std::size_t i = (std::size_t)x + (std::size_t)X * ((std::size_t)y + (std::size_t)Y * (std::size_t)z);
      saved.read(i, h, w);
      F = saved.distance(i);
      return w != 0 && F != 1.f;
    };
    for (int z = 0; z < Z - 1; ++z)
      for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x)
        {
          float F, Fn;
          if (!at(x, y, z, F))
            continue;
          const int nb[3][3] = { { x + 1, y, z }, { x, y + 1, z }, { x, y, z + 1 } };
          for (const auto& n : nb)
            if (n[0] < X && n[1] < Y && at(n[0], n[1], n[2], Fn))
              pairs += (F > 0.f && Fn < 0.f) || (F < 0.f && Fn > 0.f);
        }
    std::printf("  cloud of the uploaded volume: %zu points, %zu sign changes in the container\n", got, pairs);
    EXPECT(pairs > 100 && got == pairs);
  }
  // DeviceMemory::copyTo is a device-to-device copy into memory of its own
  kfusion::cuda::CudaData copy;
  probe.tsdf().data().copyTo(copy);
  EXPECT(copy.sizeBytes() == probe.tsdf().data().sizeBytes() && copy.ptr<void>() != probe.tsdf().data().ptr<void>());
}

// camera -> volume pose looking from `eye` at `target`, +y down (the convention of the depth frames)
static Affine3f look_at(const double eye[3], const double target[3])
{
  double f[3] = { target[0] - eye[0], target[1] - eye[1], target[2] - eye[2] };
  const double fn = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  for (double& v : f)
    v /= fn;
  const double down0[3] = { 0, 1, 0 };
  double r[3] = { down0[1] * f[2] - down0[2] * f[1], down0[2] * f[0] - down0[0] * f[2], down0[0] * f[1] - down0[1] * f[0] };
  const double rn = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  for (double& v : r)
    v /= rn;
  const double d[3] = { f[1] * r[2] - f[2] * r[1], f[2] * r[0] - f[0] * r[2], f[0] * r[1] - f[1] * r[0] };
  kfusion::Mat3f R;
  for (int i = 0; i < 3; ++i)
  {
    R(i, 0) = (float)r[i];
    R(i, 1) = (float)d[i];
    R(i, 2) = (float)f[i];
  }
  return Affine3f(R, Vec3f((float)eye[0], (float)eye[1], (float)eye[2]));
}

// ground plane y = 1.95 and three spheres (SURVEY.md 8d scene S2: a lone sphere leaves ICP rank-deficient)
static std::vector<std::uint16_t> render_s2(const Affine3f& cam, const kfusion::Intr& in, int rows, int cols)
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
      if (std::fabs(d[1]) > 1e-12)
      {
        const double s = (1.95 - o[1]) / d[1];
        if (s > 1e-6)
          best = s;
      }
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
      const double mm = std::floor(best * 1000.0 + 0.5);
      depth[(size_t)v * cols + u] = (best < 1e29 && mm > 0 && mm < 65535) ? (std::uint16_t)mm : 0;
    }
  return depth;
}

static Affine3f orbit_pose(int frame)
{
  const double ang = 1.2 * frame * 3.14159265358979323846 / 180.0, centre[3] = { 1.5, 1.5, 1.5 };
  const double eye[3] = { 1.5 + 3.2 * std::sin(ang), 1.5, 1.5 - 3.2 * std::cos(ang) };
  return look_at(eye, centre);
}

// The complete KinFu loop (kinfu.cpp:162-309) with ICP on: bilateral / pyramid / point maps, ICP against
// the ray-cast model, integrate, ray-cast.  No pose is supplied after frame 0: the tracked poses must
// follow the true orbit.
static void test_kinfu_icp_tracks_orbit()
{
  std::printf("KinFu with use_icp: tracks the S2 orbit from depth alone\n");
  kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
  p.volume_dims = kfusion::Vec3i::all(128);
  p.volume_resolution = 3.f / 128.f;
  p.volume_pose = orbit_pose(0);  // frame 0 is integrated at this pose (kinfu.cpp:193)
  kfusion::KinFu kinfu(p);
  const int frames = 12;
  double worst_t = 0, worst_r = 0;
  for (int f = 0; f < frames; ++f)
  {
    const Affine3f truth = orbit_pose(f);
    std::vector<std::uint16_t> depth = render_s2(truth, p.intr, p.rows, p.cols);
    kfusion::cuda::Depth d;
    d.upload(depth.data(), (size_t)p.cols * 2, p.rows, p.cols);
    // the supplied poses are ignored when use_icp && update_via_sensor_motion (kinfu.cpp:229-262)
    const bool ok = kinfu(Affine3f::Identity(), Affine3f::Identity(), Affine3f::Identity(), d);
    EXPECT(ok == (f > 0));
    const Affine3f est = kinfu.getCameraPose();
    const Affine3f err = truth.inv() * est;
    const double terr = yak::compat::norm(err.translation()), rerr = yak::compat::norm(err.rvec()) * 180.0 / 3.14159265;
    if (terr > worst_t)
      worst_t = terr;
    if (rerr > worst_r)
      worst_r = rerr;
  }
  std::printf("  %d frames, worst pose error: %.2f mm, %.3f deg (voxel %.1f mm, step 1.2 deg = 67 mm)\n", frames,
              worst_t * 1e3, worst_r, 3000.0 / 128);
  // drift bound: half a voxel and half a degree over the run (a lost track is off by the 67 mm / 1.2 deg steps)
  EXPECT(worst_t < 0.5 * 3.0 / 128 && worst_r < 0.5);

  // raycastPyramid (what operator() calls) against the reference's sequence, raycast + resizePointsNormals per
  // level (kinfu.cpp:294-306), four levels: the same bytes in every map
  {
    const Affine3f view = kinfu.getCameraPose();
    const int levels = 4;
    std::vector<kfusion::cuda::Cloud> pa(levels), pb(levels);
    std::vector<kfusion::cuda::Normals> na(levels), nb(levels);
    pa[0].create(p.rows, p.cols), na[0].create(p.rows, p.cols), pb[0].create(p.rows, p.cols), nb[0].create(p.rows, p.cols);
    kinfu.tsdf().raycastPyramid(view, p.intr, pa, na, levels);
    kinfu.tsdf().raycast(view, p.intr, pb[0], nb[0]);
    for (int i = 1; i < levels; ++i)
      kfusion::cuda::resizePointsNormals(pb[i - 1], nb[i - 1], pb[i], nb[i]);
    size_t surface = 0;
    for (int i = 0; i < levels; ++i)
    {
      EXPECT(pa[i].rows() == p.rows >> i && pa[i].cols() == p.cols >> i && na[i].rows() == pb[i].rows() && na[i].cols() == pb[i].cols());
      const size_t n = (size_t)pa[i].rows() * pa[i].cols() * 4;
      std::vector<float> a(n), b(n), c(n), d(n);
      pa[i].download(a.data(), (size_t)pa[i].cols() * 16), pb[i].download(b.data(), (size_t)pb[i].cols() * 16);
      na[i].download(c.data(), (size_t)na[i].cols() * 16), nb[i].download(d.data(), (size_t)nb[i].cols() * 16);
      EXPECT(std::memcmp(a.data(), b.data(), n * 4) == 0 && std::memcmp(c.data(), d.data(), n * 4) == 0);
      for (size_t k = 0; k < n; k += 4)
        surface += a[k] == a[k];
    }
    EXPECT(surface > 10000);
  }

  // a featureless frame makes the normal matrix singular: the loop reports failure and starts over
  // (kinfu.cpp:247-252)
  std::vector<std::uint16_t> blank((size_t)p.rows * p.cols, 0);
  kfusion::cuda::Depth d;
  d.upload(blank.data(), (size_t)p.cols * 2, p.rows, p.cols);
  EXPECT(!kinfu(Affine3f::Identity(), Affine3f::Identity(), Affine3f::Identity(), d));
  const Affine3f after = kinfu.getCameraPose();
  for (int i = 0; i < 16; ++i)
    EXPECT(after.matrix.val[i] == p.volume_pose.matrix.val[i]);
}

// The multi-GPU host path on the one GPU of the test box: kfusion::ShardedKinFu with several ranks on device 0 (its
// ranks are threads with their own contexts, slabs and halos, talking through the library's in-process transport;
// on a multi-GPU node the same code runs one rank per GPU over RCCL).  Bars as tests/test_sharded_kinfu_gpu.py:
// supplied poses -> the slabs concatenate to the one-GPU volume and every rank's model maps equal the one-GPU ray
// cast, bit for bit, on every frame; tracking -> identical poses on all ranks, within rounding of the one-GPU track.
static void test_sharded_kinfu_against_single()
{
  std::printf("ShardedKinFu (3 ranks on one GPU): supplied poses, bit for bit against KinFu\n");
  kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
  p.volume_dims = kfusion::Vec3i::all(96);
  p.volume_resolution = 3.f / 96.f;
  p.volume_pose = orbit_pose(0);
  p.use_icp = false;
  p.update_via_sensor_motion = false;
  const size_t voxels = (size_t)96 * 96 * 96, pixels = (size_t)p.rows * p.cols;
  const int frames = 6;
  {
    kfusion::KinFu single(p);
    kfusion::ShardedKinFu sharded(p, std::vector<int>{ 0, 0, 0 });
    EXPECT(sharded.world() == 3);
    std::vector<float> sp(pixels * 4), sn(pixels * 4), rp(pixels * 4), rn(pixels * 4);
    for (int f = 0; f < frames; ++f)
    {
      const Affine3f pose = orbit_pose(3 * f), prev = orbit_pose(f ? 3 * (f - 1) : 0);
      std::vector<std::uint16_t> depth = render_s2(pose, p.intr, p.rows, p.cols);
      kfusion::cuda::Depth d;
      d.upload(depth.data(), (size_t)p.cols * 2, p.rows, p.cols);
      const bool a = single(Affine3f::Identity(), pose, prev, d);
      const bool b = sharded(Affine3f::Identity(), pose, prev, depth.data(), (size_t)p.cols * 2, p.rows, p.cols);
      EXPECT(a == b && a == (f > 0));
      if (f == 0)
        continue;
      single.downloadModelMaps(sp.data(), sn.data());
      for (int r = 0; r < 3; ++r)
      {
        sharded.downloadModelMaps(r, rp.data(), rn.data());
        EXPECT(std::memcmp(sp.data(), rp.data(), pixels * 16) == 0);
        EXPECT(std::memcmp(sn.data(), rn.data(), pixels * 16) == 0);
      }
    }
    size_t hits = 0;
    for (size_t i = 0; i < pixels; ++i)
      hits += sp[4 * i] == sp[4 * i];
    EXPECT(hits > 50000);
    std::vector<std::uint32_t> one(voxels), many(voxels, 0xdeadbeefu);
    single.tsdf().data().download(one.data());
    sharded.downloadVolume(many.data());
    EXPECT(std::memcmp(one.data(), many.data(), voxels * 4) == 0);
    size_t nonzero = 0;
    for (size_t i = 0; i < voxels; ++i)
      nonzero += one[i] != 0;
    EXPECT(nonzero > voxels / 20);
  }

  std::printf("ShardedKinFu (2 ranks on one GPU): tracking from depth alone, lost track\n");
  p.use_icp = true;
  p.update_via_sensor_motion = true;
  {
    kfusion::KinFu single(p);
    kfusion::ShardedKinFu sharded(p, std::vector<int>{ 0, 0 });
    double worst = 0, worst_diff = 0;
    for (int f = 0; f < 8; ++f)
    {
      const Affine3f truth = orbit_pose(f);
      std::vector<std::uint16_t> depth = render_s2(truth, p.intr, p.rows, p.cols);
      kfusion::cuda::Depth d;
      d.upload(depth.data(), (size_t)p.cols * 2, p.rows, p.cols);
      const bool a = single(Affine3f::Identity(), Affine3f::Identity(), Affine3f::Identity(), d);
      const bool b = sharded(Affine3f::Identity(), Affine3f::Identity(), Affine3f::Identity(), depth.data(),
                             (size_t)p.cols * 2, p.rows, p.cols);
      EXPECT(a == b);
      const Affine3f e1 = single.getCameraPose(), e2 = sharded.getCameraPose();
      for (int i = 0; i < 16; ++i)
        worst_diff = std::max(worst_diff, (double)std::fabs(e1.matrix.val[i] - e2.matrix.val[i]));
      worst = std::max(worst, yak::compat::norm((truth.inv() * e2).translation()));
    }
    std::printf("  worst position error %.2f mm, worst difference to the one-GPU track %.2e\n", worst * 1e3, worst_diff);
    EXPECT(worst < 0.5 * 3.0 / 96 && worst_diff < 5e-4);
    // a frame without returns: every rank reports the failure and starts over (kinfu.cpp:247-252)
    std::vector<std::uint16_t> blank((size_t)p.rows * p.cols, 0);
    EXPECT(!sharded(Affine3f::Identity(), Affine3f::Identity(), Affine3f::Identity(), blank.data(), (size_t)p.cols * 2, p.rows,
                    p.cols));
    const Affine3f after = sharded.getCameraPose();
    for (int i = 0; i < 16; ++i)
      EXPECT(after.matrix.val[i] == p.volume_pose.matrix.val[i]);
    std::vector<std::uint32_t> vol((size_t)96 * 96 * 96, 1u);
    sharded.downloadVolume(vol.data());
    size_t nonzero = 0;
    for (std::uint32_t w : vol)
      nonzero += w != 0;
    EXPECT(nonzero == 0);
  }
  EXPECT(kfusion::ShardedKinFu::devicesFromEnvironment().empty() || std::getenv("YAK_HIP_DEVICES"));
}


// A rank that throws mid-frame must not leave its peers waiting in an exchange for ever:
// the worker aborts its communicator, the dispatching thread aborts the peers', everybody comes back with an error.
static void test_sharded_failure_wakes_peers()
{
  std::printf("ShardedKinFu: a rank that throws wakes the ranks waiting for it\n");
  kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
  p.volume_dims = kfusion::Vec3i::all(64);
  p.volume_resolution = 3.f / 64.f;
  kfusion::ShardedKinFu sharded(p, std::vector<int>{ 0, 0, 0 });
  auto exchange = [](int rank) {
    const kfusion::cuda::Shard& sh = kfusion::cuda::currentShard();
    kfusion::cuda::DeviceArray<int> buf(4);
    const int ones[4] = { 1, 1, 1, 1 };
    buf.upload(ones, 4);
    if (yakhip_comm_allreduce(kfusion::cuda::context(), sh.comm, buf.ptr(), 4, YAKHIP_SUM_I32) != 0)
      throw std::runtime_error(yakhip_last_error());
    int got[4] = { 0, 0, 0, 0 };
    buf.download(got);
    if (got[0] != sh.world || got[3] != sh.world)
      throw std::runtime_error("rank " + std::to_string(rank) + ": wrong sum");
  };
  bool ok = true;
  try
  {
    sharded.forEachRank([&](int rank, kfusion::KinFu&) { exchange(rank); });  // all three: works
  }
  catch (const std::exception& e)
  {
    std::printf("  unexpected: %s\n", e.what());
    ok = false;
  }
  EXPECT(ok);
  std::string what;
  try
  {
    sharded.forEachRank([&](int rank, kfusion::KinFu&) {
      if (rank == 1)
        throw std::runtime_error("rank 1 gives up");
      exchange(rank);  // ranks 0 and 2 wait for rank 1, which never comes
    });
  }
  catch (const std::exception& e)
  {
    what = e.what();
  }
  std::printf("  forEachRank threw: %s\n", what.c_str());
  EXPECT(what.find("rank 1 gives up") != std::string::npos);  // the cause, not a peer's "left the group"
  // the communicators are dead: the next exchange fails at once on every rank instead of hanging
  what.clear();
  try
  {
    sharded.forEachRank([&](int rank, kfusion::KinFu&) { exchange(rank); });
  }
  catch (const std::exception& e)
  {
    what = e.what();
  }
  EXPECT(what.find("abort") != std::string::npos || what.find("left the group") != std::string::npos);
}  // (and the destructor joins the workers)

// The halo of a rank's slab follows trunc_dist / voxel size / the ray-cast factors; changing one of them after frames
// have been fused keeps the fused model (the reference's setters never touch the voxels, tsdf_volume.cpp:54-78).
static void test_slab_range_change_keeps_model()
{
  std::printf("ShardedKinFu: a setter that moves the halo keeps the fused model\n");
  kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
  p.volume_dims = kfusion::Vec3i::all(96);
  p.volume_resolution = 3.f / 96.f;
  p.volume_pose = orbit_pose(0);
  p.use_icp = false;
  p.update_via_sensor_motion = false;
  const size_t voxels = (size_t)96 * 96 * 96;
  kfusion::KinFu single(p);
  kfusion::ShardedKinFu sharded(p, std::vector<int>{ 0, 0, 0 });
  auto frame = [&](int f) {
    const Affine3f pose = orbit_pose(3 * f), prev = orbit_pose(f ? 3 * (f - 1) : 0);
    std::vector<std::uint16_t> depth = render_s2(pose, p.intr, p.rows, p.cols);
    kfusion::cuda::Depth d;
    d.upload(depth.data(), (size_t)p.cols * 2, p.rows, p.cols);
    single(Affine3f::Identity(), pose, prev, d);
    sharded(Affine3f::Identity(), pose, prev, depth.data(), (size_t)p.cols * 2, p.rows, p.cols);
  };
  std::vector<std::uint32_t> one(voxels), many(voxels);
  auto same_volume = [&]() {
    single.tsdf().data().download(one.data());
    std::fill(many.begin(), many.end(), 0xdeadbeefu);
    sharded.downloadVolume(many.data());
    size_t nonzero = 0;
    for (size_t i = 0; i < voxels; ++i)
      nonzero += one[i] != 0;
    return nonzero > voxels / 20 && std::memcmp(one.data(), many.data(), voxels * 4) == 0;
  };
  for (int f = 0; f < 3; ++f)
    frame(f);
  EXPECT(same_volume());
  std::vector<int> held_before(6), held_after(6);
  auto held = [&](std::vector<int>& out) {
    sharded.forEachRank([&](int rank, kfusion::KinFu& kf) { kf.tsdf().heldLayers(out[2 * rank], out[2 * rank + 1]); });
  };
  held(held_before);
  const float factors[2] = { 4.f, 0.25f };  // the halo grows, then shrinks below where it began
  for (float factor : factors)
  {
    single.tsdf().setGradientDeltaFactor(factor);
    sharded.forEachRank([&](int, kfusion::KinFu& kf) { kf.tsdf().setGradientDeltaFactor(factor); });
    held(held_after);
    EXPECT(held_after != held_before);
    held_before = held_after;
    frame(3 + (factor < 1.f));  // the next frame reallocates the slab: the model so far must come along
    EXPECT(same_volume());
  }
}

// TsdfVolume::data() hands out a handle that aliases the voxels; writes through it after a ray cast has already
// rebuilt the surface map must still be seen by the next ray cast (staleness is sticky while a handle lives).
static void test_writable_handle_keeps_map_honest()
{
  std::printf("TsdfVolume::data(): writes through a kept handle after a ray cast\n");
  kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
  p.volume_dims = kfusion::Vec3i::all(96);
  p.volume_resolution = 3.f / 96.f;
  p.volume_pose = orbit_pose(0);
  p.use_icp = false;
  p.update_via_sensor_motion = false;
  const size_t voxels = (size_t)96 * 96 * 96;
  const int rows = p.rows, cols = p.cols;
  kfusion::KinFu full(p), sparse(p);
  for (int f = 0; f < 4; ++f)
  {
    const Affine3f pose = orbit_pose(3 * f);
    std::vector<std::uint16_t> depth = render_s2(pose, p.intr, rows, cols);
    kfusion::cuda::Depth d;
    d.upload(depth.data(), (size_t)cols * 2, rows, cols);
    full(Affine3f::Identity(), pose, pose, d);
    if (f == 0)
    {
      for (int y = 0; y < rows; ++y)  // `sparse` sees the left third of one frame only: few surface bricks
        for (int x = cols / 3; x < cols; ++x)
          depth[(size_t)y * cols + x] = 0;
      d.upload(depth.data(), (size_t)cols * 2, rows, cols);
      sparse(Affine3f::Identity(), pose, pose, d);
    }
  }
  std::vector<std::uint32_t> model(voxels);
  { const kfusion::cuda::TsdfVolume& cv = full.tsdf(); cv.data().download(model.data()); }
  const Affine3f view = orbit_pose(4);
  auto cast = [&](kfusion::cuda::TsdfVolume& vol, std::vector<float>& hp, std::vector<float>& hn) {
    kfusion::cuda::Cloud points(rows, cols);
    kfusion::cuda::Normals normals(rows, cols);
    vol.raycast(view, p.intr, points, normals);
    hp.resize((size_t)rows * cols * 4), hn.resize(hp.size());
    points.download(hp.data(), (size_t)cols * 16);
    normals.download(hn.data(), (size_t)cols * 16);
  };
  std::vector<float> want_p, want_n, got_p, got_n;
  cast(full.tsdf(), want_p, want_n);
  kfusion::cuda::CudaData handle = sparse.tsdf().data();  // writable, kept
  cast(sparse.tsdf(), got_p, got_n);                      // rebuilds the map of the sparse model
  size_t lit_sparse = 0, lit_full = 0;
  for (size_t i = 0; i < (size_t)rows * cols; ++i)
    lit_sparse += got_p[4 * i] == got_p[4 * i], lit_full += want_p[4 * i] == want_p[4 * i];
  EXPECT(lit_sparse * 2 < lit_full);
  handle.upload(model.data(), voxels * 4);                // the full model arrives behind the class's back
  cast(sparse.tsdf(), got_p, got_n);
  EXPECT(std::memcmp(got_p.data(), want_p.data(), want_p.size() * 4) == 0);
  EXPECT(std::memcmp(got_n.data(), want_n.data(), want_n.size() * 4) == 0);
  handle.release();                                        // no outside handle left: one last rebuild, then the map is trusted
  cast(sparse.tsdf(), got_p, got_n);
  EXPECT(std::memcmp(got_p.data(), want_p.data(), want_p.size() * 4) == 0);
}

// yak::FusionServer spread over ranks by YAK_HIP_DEVICES: fuse / downloadTSDF / reset / uploadTSDF / display
// give what the one-GPU server gives (the volume word for word: supplied poses).
static void test_fusion_server_over_ranks()
{
  std::printf("FusionServer with YAK_HIP_DEVICES=0,0,0 against the one-GPU server\n");
  kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
  p.volume_dims = kfusion::Vec3i(64, 48, 56);
  p.volume_resolution = 3.f / 64.f;
  p.use_icp = false;
  p.use_pose_hints = true;
  p.update_via_sensor_motion = false;
  unsetenv("YAK_HIP_DEVICES");
  yak::FusionServer one(p, Eigen::Affine3f::Identity());
  setenv("YAK_HIP_DEVICES", "0,0,0", 1);
  yak::FusionServer many(p, Eigen::Affine3f::Identity());
  unsetenv("YAK_HIP_DEVICES");
  const Affine3f poses[4] = { rot_y_translate(0.f, 1.5f, 1.2f, -2.0f), rot_y_translate(0.1f, 1.1f, 1.25f, -2.1f),
                              rot_y_translate(-0.15f, 1.9f, 1.1f, -1.9f), rot_y_translate(0.25f, 0.8f, 1.3f, -2.2f) };
  for (int f = 0; f < 4; ++f)
  {
    std::vector<std::uint16_t> depth = render(poses[f], p.intr, p.rows, p.cols);
    const cv::Mat m(p.rows, p.cols, depth.data(), (size_t)p.cols * 2);
    EXPECT(one.fuse(m, Eigen::Affine3f(poses[f])) == many.fuse(m, Eigen::Affine3f(poses[f])));
  }
  const yak::TSDFContainer a = one.downloadTSDF(), b = many.downloadTSDF();
  EXPECT(a.size64() == b.size64() && std::memcmp(a.data(), b.data(), a.size64() * 4) == 0);
  size_t touched = 0;
  for (size_t i = 0; i < a.size64(); ++i)
    touched += a.data()[i] != 0;
  EXPECT(touched > 1000);
  // the tracker's model view: the same pixels from rank 0's composited maps
  one.display();
  many.display();
  int r1, c1, r2, c2;
  const std::vector<unsigned char>& v1 = one.view(r1, c1);
  const std::vector<unsigned char>& v2 = many.view(r2, c2);
  EXPECT(r1 == r2 && c1 == c2 && v1.size() == v2.size() && !v1.empty() && v1 == v2);
  EXPECT(many.reset());
  const yak::TSDFContainer cleared = many.downloadTSDF();
  size_t nz = 0;
  for (size_t i = 0; i < cleared.size64(); ++i)
    nz += cleared.data()[i] != 0;
  EXPECT(nz == 0);
  EXPECT(many.uploadTSDF(a));  // every rank takes the layers it holds, halo included
  const yak::TSDFContainer back = many.downloadTSDF();
  EXPECT(std::memcmp(back.data(), a.data(), a.size64() * 4) == 0);
  EXPECT(!many.uploadTSDF(yak::TSDFContainer(8, 8, 8)));
  // and fusion goes on from the uploaded volume like the one-GPU server's (reset() restarts the frame count on both)
  EXPECT(one.reset() && one.uploadTSDF(a));
  std::vector<std::uint16_t> depth = render(poses[1], p.intr, p.rows, p.cols);
  const cv::Mat m(p.rows, p.cols, depth.data(), (size_t)p.cols * 2);
  one.fuse(m, Eigen::Affine3f(poses[1]));
  many.fuse(m, Eigen::Affine3f(poses[1]));
  const yak::TSDFContainer a2 = one.downloadTSDF(), b2 = many.downloadTSDF();
  EXPECT(std::memcmp(a2.data(), b2.data(), a2.size64() * 4) == 0);
}

// The reference's own marching-cubes test, case for case (yak/test/marching_cubes_tests.cpp:86-150):
// a 3 x 3 x 3 volume of half(1.0), weight 1, with half(-1.0) in the middle, scale 0.01, min_weight 1,
// clean -> 8 faces over 6 distinct vertices at the positions of the test's PLY text.
static void test_marching_cubes_dipyramid()
{
  std::printf("marchingCubesCPU: the reference's dipyramid known answer\n");
  yak::TSDFContainer volume(3, 3, 3);
  for (int i = 0; i < 27; ++i)
    volume.data()[i] = 0x00013C00u;                      // pack(half(1.0), 1)
  volume.data()[volume.toIndex(1, 1, 1)] = 0x0001BC00u;  // pack(half(-1.0), 1)
  yak::MarchingCubesParameters mc;
  mc.scale = 0.01;
  mc.min_weight = 1;
  mc.clean = true;
  const yak::PolygonMesh mesh = yak::marchingCubesCPU(volume, mc);
  EXPECT(mesh.polygons.size() == 8);  // :121
  EXPECT(mesh.cloud.size() == 6);     // :133
  const double want[6][3] = { { 0.01, 0.005, 0.01 }, { 0.005, 0.01, 0.01 }, { 0.01, 0.01, 0.005 },
                              { 0.015, 0.01, 0.01 }, { 0.01, 0.01, 0.015 }, { 0.01, 0.015, 0.01 } };  // :24-29
  for (const auto& w : want)  // :136-149: every baseline vertex has a mesh vertex in its 0.001 octree cell
  {
    bool found = false;
    for (const auto& p : mesh.cloud)
      found = found || (std::fabs(p[0] - w[0]) < 1e-6 && std::fabs(p[1] - w[1]) < 1e-6 && std::fabs(p[2] - w[2]) < 1e-6);
    EXPECT(found);
  }
  for (const auto& f : mesh.polygons)
    EXPECT(f[0] < 6 && f[1] < 6 && f[2] < 6 && f[0] != f[1] && f[1] != f[2] && f[0] != f[2]);
  mc.clean = false;  // marching_cubes.cpp:217-220: three fresh points per triangle
  const yak::PolygonMesh raw = yak::marchingCubesCPU(volume, mc);
  EXPECT(raw.polygons.size() == 8 && raw.cloud.size() == 24);
  mc.min_weight = 2;  // every weight is 1: all cubes dropped (:51-74)
  EXPECT(yak::marchingCubesCPU(volume, mc).polygons.empty());
}

// KinFu::downloadCloud / TsdfVolume::fetchCloud / fetchNormals (reference kinfu.cpp:368-377,
// tsdf_volume.cpp:147-183) against the oracle on the same fused volume: the same point set, and for the
// library's own point order the same normals, bit for bit.
static void test_cloud_and_normals_against_oracle()
{
  std::printf("KinFu::downloadCloud / TsdfVolume::fetchNormals vs oracle\n");
  kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
  p.volume_dims = kfusion::Vec3i(96, 80, 72);
  p.volume_resolution = 3.f / 96.f;
  p.volume_pose = rot_y_translate(0.2f, -1.4f, -1.2f, 0.6f);
  p.use_icp = false;
  p.use_pose_hints = true;
  p.update_via_sensor_motion = false;
  kfusion::KinFu kinfu(p);
  EXPECT(kinfu.downloadCloud().empty());  // an empty volume has no zero crossing
  for (int f = 0; f < 3; ++f)
  {
    const Affine3f pose = orbit_pose(f * 4);
    std::vector<std::uint16_t> depth = render_s2(pose, p.intr, p.rows, p.cols);
    kfusion::cuda::Depth d;
    d.upload(depth.data(), (size_t)p.cols * 2, p.rows, p.cols);
    kinfu(Affine3f::Identity(), pose, pose, d);
  }
  const kfusion::cuda::TsdfVolume& tv = kinfu.tsdf();
  std::vector<std::uint32_t> vol((size_t)96 * 80 * 72);
  tv.data().download(vol.data());
  const int dims[3] = { 96, 80, 72 };
  const Vec3f vs = tv.getVoxelSize();
  const float vsz[3] = { vs[0], vs[1], vs[2] };
  const Affine3f aff = tv.getPose();
  const auto R = aff.rotation();
  const auto Rinv = aff.rotation().inv(cv::DECOMP_SVD);
  const float t[3] = { aff.translation()[0], aff.translation()[1], aff.translation()[2] };

  std::vector<kfusion::Point> got = kinfu.downloadCloud();
  const std::uint64_t n = yko_extract_cloud(vol.data(), dims, vsz, R.val, t, nullptr, 0);
  std::vector<float> want((size_t)n * 4);
  yko_extract_cloud(vol.data(), dims, vsz, R.val, t, want.data(), n);
  std::printf("  %zu points (oracle %llu)\n", got.size(), (unsigned long long)n);
  EXPECT(got.size() == n && n > 1000);
  typedef std::array<std::uint32_t, 4> Bits;
  std::vector<Bits> a(got.size()), b((size_t)n);
  for (size_t i = 0; i < got.size(); ++i)
    std::memcpy(a[i].data(), got[i].data, 16);
  for (size_t i = 0; i < (size_t)n; ++i)
    std::memcpy(b[i].data(), &want[i * 4], 16);
  std::sort(a.begin(), a.end());
  std::sort(b.begin(), b.end());
  EXPECT(a == b);

  // fetchCloud into a caller's small buffer stops at its size (tsdf_volume.cu:639-642)
  kfusion::cuda::DeviceArray<kfusion::Point> small(100), buffer;
  EXPECT(tv.fetchCloud(small).size() == 100);
  kfusion::cuda::DeviceArray<kfusion::Point> cloud = tv.fetchCloud(buffer);
  EXPECT(buffer.size() == 10u * 1000 * 1000 && cloud.size() == n && cloud.ptr() == buffer.ptr());
  kfusion::cuda::DeviceArray<kfusion::Normal> normals;
  tv.fetchNormals(cloud, normals);
  EXPECT(normals.size() == cloud.size());
  std::vector<kfusion::Point> pts(cloud.size()), nrm(cloud.size());
  cloud.download(pts.data());
  normals.download(nrm.data());
  std::vector<float> want_n(pts.size() * 4);
  yko_extract_normals(vol.data(), dims, vsz, pts[0].data, pts.size(), R.val, t, Rinv.val, p.gradient_delta_factor,
                      want_n.data());
  size_t diff = 0, nan = 0;
  for (size_t i = 0; i < pts.size(); ++i)
  {
    diff += std::memcmp(nrm[i].data, &want_n[i * 4], 12) != 0 && !(std::isnan(nrm[i].x) && std::isnan(want_n[i * 4]));
    nan += std::isnan(nrm[i].x);
  }
  std::printf("  %zu normals differ, %zu NaN\n", diff, nan);
  EXPECT(diff == 0 && nan < pts.size() / 2);
}

// KinFu::renderImage (both overloads, flags 0 / 2 / 3) and FusionServer::display (reference kinfu.cpp:311-366,
// yak_server.cpp:90-110) against the oracle's restatement of imgproc.cu:400-561 applied to the same
// ray-cast maps: every pixel the same four bytes.
static void test_render_views_against_oracle()
{
  std::printf("KinFu::renderImage / FusionServer::display vs oracle\n");
  kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
  p.volume_dims = kfusion::Vec3i::all(128);
  p.volume_resolution = 3.f / 128.f;
  p.volume_pose = orbit_pose(0);
  p.use_icp = false;
  p.use_pose_hints = true;
  p.update_via_sensor_motion = false;
  p.light_pose = Vec3f(0.5f, -0.3f, -1.f);
  kfusion::KinFu kinfu(p);
  Affine3f pose;
  for (int f = 0; f < 3; ++f)
  {
    pose = orbit_pose(f * 3);
    std::vector<std::uint16_t> depth = render_s2(pose, p.intr, p.rows, p.cols);
    kfusion::cuda::Depth d;
    d.upload(depth.data(), (size_t)p.cols * 2, p.rows, p.cols);
    kinfu(Affine3f::Identity(), pose, pose, d);
  }
  const int rows = p.rows, cols = p.cols;
  kfusion::cuda::Cloud points(rows, cols);  // the caller allocates (tsdf_volume.cpp:128-145)
  kfusion::cuda::Normals normals(rows, cols);
  kinfu.tsdf().raycast(pose, p.intr, points, normals);
  std::vector<float> hp((size_t)rows * cols * 4), hn(hp.size());
  points.download(hp.data(), (size_t)cols * 16);
  normals.download(hn.data(), (size_t)cols * 16);
  std::vector<std::uint8_t> want_phong((size_t)rows * cols * 4), want_tangent(want_phong.size());
  yko_render_image_points(hp.data(), (size_t)cols * 16, hn.data(), (size_t)cols * 16, rows, cols, p.light_pose.val,
                          want_phong.data(), (size_t)cols * 4);
  yko_render_tangent_colors(hn.data(), (size_t)cols * 16, rows, cols, want_tangent.data(), (size_t)cols * 4);
  size_t lit = 0;
  for (size_t i = 0; i < (size_t)rows * cols; ++i)
    lit += !std::isnan(hp[i * 4]);
  std::printf("  %zu of %d pixels hit the model\n", lit, rows * cols);
  EXPECT(lit > (size_t)rows * cols / 16);

  auto download = [](const kfusion::cuda::Image& img) {
    std::vector<std::uint8_t> h((size_t)img.rows() * img.cols() * 4);
    img.download(h.data(), (size_t)img.cols() * 4);
    return h;
  };
  auto half = [&](const std::vector<std::uint8_t>& wide, int which) {  // left / right half of a rows x 2 cols image
    std::vector<std::uint8_t> out((size_t)rows * cols * 4);
    for (int y = 0; y < rows; ++y)
      std::memcpy(&out[(size_t)y * cols * 4], &wide[((size_t)y * 2 + which) * cols * 4], (size_t)cols * 4);
    return out;
  };
  for (int from_pose = 0; from_pose < 2; ++from_pose)  // the tracker's own maps are the ray cast of the last pose
  {
    kfusion::cuda::Image image;
    auto render = [&](int flag) { from_pose ? kinfu.renderImage(image, pose, flag) : kinfu.renderImage(image, flag); };
    render(0);
    EXPECT(image.rows() == rows && image.cols() == cols && download(image) == want_phong);
    render(2);
    EXPECT(image.rows() == rows && image.cols() == cols && download(image) == want_tangent);
    render(3);
    EXPECT(image.rows() == rows && image.cols() == 2 * cols);
    const std::vector<std::uint8_t> wide = download(image);
    EXPECT(half(wide, 0) == want_phong && half(wide, 1) == want_tangent);
  }

  // the Depth overload re-projects with (fx, fy, cx, fy) as imgproc.cpp:92 does
  kfusion::cuda::Depth depth_map(rows, cols);
  kfusion::cuda::Normals normals_d(rows, cols);
  kinfu.tsdf().raycast(pose, p.intr, depth_map, normals_d);
  std::vector<std::uint16_t> hd((size_t)rows * cols);
  depth_map.download(hd.data(), (size_t)cols * 2);
  normals_d.download(hn.data(), (size_t)cols * 16);
  const float quirk[4] = { p.intr.fx, p.intr.fy, p.intr.cx, p.intr.fy };
  yko_render_image_depth(hd.data(), (size_t)cols * 2, hn.data(), (size_t)cols * 16, rows, cols, quirk, p.light_pose.val,
                         want_phong.data(), (size_t)cols * 4);
  kfusion::cuda::Image image;
  kfusion::cuda::renderImage(depth_map, normals_d, p.intr, p.light_pose, image);
  EXPECT(download(image) == want_phong);

  // FusionServer: headless display() keeps the view the reference would show
  yak::FusionServer server(p, Eigen::Affine3f::Identity());
  int vr = -1, vc = -1;
  EXPECT(server.view(vr, vc).empty() && vr == 0 && vc == 0);
  std::vector<std::uint16_t> depth = render_s2(orbit_pose(0), p.intr, rows, cols);
  server.fuse(cv::Mat(rows, cols, depth.data(), (size_t)cols * 2), Eigen::Affine3f(orbit_pose(0)));
  EXPECT(server.view(vr, vc).empty());  // YAK_HIP_DISPLAY is not set: fuse() does not render
  server.display(Eigen::Affine3f(orbit_pose(2)));
  const std::vector<std::uint8_t>& view = server.view(vr, vc);
  EXPECT(vr == rows && vc == 2 * cols && view.size() == (size_t)rows * cols * 8);
  size_t grey = 0;
  for (size_t i = 0; i < view.size(); i += 4)
    grey += view[i] == view[i + 1] && view[i + 1] == view[i + 2] && view[i] > 70;
  std::printf("  %zu shaded pixels in the server's view\n", grey);
  EXPECT(grey > (size_t)rows * cols / 64);  // Phong pixels are grey, at least the ambient 0.3 * 255
}

static void test_device_arrays()
{
  std::printf("DeviceArray / DeviceArray2D\n");
  std::vector<unsigned short> host(37 * 53);
  for (size_t i = 0; i < host.size(); ++i)
    host[i] = (unsigned short)(i * 7u);
  kfusion::cuda::Depth d;
  d.upload(host.data(), 53 * 2, 37, 53);
  EXPECT(d.rows() == 37 && d.cols() == 53 && d.step() >= 106 && d.step() % 256 == 0);
  kfusion::cuda::Depth alias = d, copy;
  d.copyTo(copy);
  std::vector<unsigned short> back(host.size(), 0);
  copy.download(back.data(), 53 * 2);
  EXPECT(back == host);
  EXPECT(alias.ptr() == d.ptr() && copy.ptr() != d.ptr());
  d.release();
  EXPECT(d.empty() && !alias.empty());
  kfusion::cuda::DeviceArray<int> a(100);
  EXPECT(a.size() == 100 && a.sizeBytes() == 400);
}

int main(int argc, char** argv)
{
  const std::string mode = argc > 1 ? argv[1] : "--cpu";
  try
  {
    if (mode == "--cpu")
    {
      test_params_defaults();
      test_affine_math();
      test_container_and_half();
      test_solve6();
      test_no_gpu_is_an_exception();
    }
    else if (mode == "--gpu-sharded")
    {
      test_sharded_kinfu_against_single();
      test_fusion_server_over_ranks();
      test_sharded_failure_wakes_peers();
      test_slab_range_change_keeps_model();
    }
    else
    {
      std::printf("device: %s\n", kfusion::cuda::getDeviceName(0).c_str());
      test_device_arrays();
      test_fusion_server_against_oracle();
      test_kinfu_icp_tracks_orbit();
      test_marching_cubes_dipyramid();
      test_cloud_and_normals_against_oracle();
      test_render_views_against_oracle();
      test_writable_handle_keeps_map_honest();
    }
  }
  catch (const std::exception& e)
  {
    std::printf("  FAIL uncaught exception: %s\n", e.what());
    ++g_failed;
  }
  std::printf(g_failed ? "FAILED (%d)\n" : "PASSED\n", g_failed);
  return g_failed ? 1 : 0;
}
