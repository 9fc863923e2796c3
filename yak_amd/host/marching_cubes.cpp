// yak::marchingCubesCPU (reference yak/src/mc/marching_cubes.cpp:158-230): container -> triangles ->
// mesh.  The triple loop over cubes runs on the GPU; the optional cleaning is done here.
#include <yak/mc/marching_cubes.h>

#include <yak/kfusion/cuda/device_memory.hpp>
#include <yakhip.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace yak
{
PolygonMesh marchingCubesDevice(const void* volume_dev, int dim_x, int dim_y, int dim_z,
                                const MarchingCubesParameters& params)
{
  using kfusion::cuda::context;
  using kfusion::cuda::error;
  yakhip_volume v = {};
  v.data = const_cast<void*>(volume_dev);
  v.dims[0] = dim_x, v.dims[1] = dim_y, v.dims[2] = dim_z;
  v.z0 = 0, v.z1 = dim_z;
  v.voxel_size[0] = v.voxel_size[1] = v.voxel_size[2] = 1.f;  // vertices are in voxel units times params.scale
  v.trunc_dist = 1.f;
  v.max_weight = 1;

  // YAK_MC_TIMING=1: where the call's time goes, on stderr
  const bool timing = std::getenv("YAK_MC_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  const auto lap = [&](const char* what) {
    if (!timing)
      return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "marchingCubesDevice: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };

  unsigned long long n = 0;
  if (yakhip_marching_cubes(context(), &v, params.min_weight, params.scale, nullptr, 0, &n) != 0)
    error(yakhip_last_error(), __FILE__, __LINE__, "yakhip_marching_cubes(count)");
  lap("count pass");
  PolygonMesh mesh;
  if (n == 0)
    return mesh;
  kfusion::cuda::DeviceMemory out((size_t)n * 9 * sizeof(float));
  unsigned long long m = 0;
  if (yakhip_marching_cubes(context(), &v, params.min_weight, params.scale, out.ptr<void>(), n, &m) != 0 || m != n)
    error(yakhip_last_error(), __FILE__, __LINE__, "yakhip_marching_cubes(emit)");
  lap("allocate + emit pass");

  if (!params.clean)  // marching_cubes.cpp:217-220: three fresh points per triangle
  {
    std::vector<float> tri((size_t)n * 9);
    out.download(tri.data());
    mesh.cloud.resize((size_t)n * 3);
    mesh.polygons.resize((size_t)n);
    for (size_t i = 0; i < (size_t)n * 3; ++i)
      mesh.cloud[i] = { tri[3 * i], tri[3 * i + 1], tri[3 * i + 2] };
    for (size_t i = 0; i < (size_t)n; ++i)
      mesh.polygons[i] = { (std::uint32_t)(3 * i), (std::uint32_t)(3 * i + 1), (std::uint32_t)(3 * i + 2) };
    return mesh;
  }

  // :204-216 -- exact point merging (bit patterns) on the device: distinct points in the order of their first
  // appearance in the triangle list, as vtkCleanPolyData's point locator numbers them, and every corner's point
  // (yakhip_weld_vertices: a hash table and a scan instead of the 120 ms a four-thread host sort of 2.4 M corners
  // took); then triangles with a repeated index are dropped.
  const size_t corners = (size_t)n * 3;
  kfusion::cuda::DeviceMemory verts(corners * 3 * sizeof(float)), index_dev(corners * sizeof(std::uint32_t));
  unsigned long long nv = 0;
  if (yakhip_weld_vertices(context(), out.ptr<void>(), n, verts.ptr<void>(), index_dev.ptr<void>(), &nv) != 0)
    error(yakhip_last_error(), __FILE__, __LINE__, "yakhip_weld_vertices");
  lap("allocate + weld");
  mesh.cloud.resize((size_t)nv);  // std::array<float, 3>: 12 contiguous bytes a point, as the device wrote them
  // the corner indices land in the polygon list itself (std::array<uint32_t, 3>: a triangle's three indices in a row)
  // and the triangles with a repeated index are squeezed out in place: no second 12 MB buffer to fault in and fill
  mesh.polygons.resize((size_t)n);
  static_assert(sizeof(mesh.polygons[0]) == 3 * sizeof(std::uint32_t), "a polygon is three packed indices");
  if (yakhip_download(context(), mesh.cloud.data(), verts.ptr<void>(), (size_t)nv * 3 * sizeof(float)) != 0 ||
      yakhip_download(context(), mesh.polygons.data(), index_dev.ptr<void>(), corners * sizeof(std::uint32_t)) != 0)
    error(yakhip_last_error(), __FILE__, __LINE__, "yakhip_download(weld)");
  lap("download vertices + indices");
  size_t kept = 0;
  for (size_t i = 0; i < (size_t)n; ++i)
  {
    const std::array<std::uint32_t, 3> tri = mesh.polygons[i];
    if (tri[0] != tri[1] && tri[1] != tri[2] && tri[0] != tri[2])
      mesh.polygons[kept++] = tri;
  }
  mesh.polygons.resize(kept);
  lap("drop degenerate triangles");
  return mesh;
}

PolygonMesh marchingCubesCPU(const TSDFContainer& tsdf, const MarchingCubesParameters& params)
{
  const Vector3i d = tsdf.dims();
  kfusion::cuda::DeviceMemory vol;
  vol.upload(tsdf.data(), tsdf.size64() * sizeof(std::uint32_t));
  return marchingCubesDevice(vol.ptr<void>(), d.x(), d.y(), d.z(), params);
}
}  // namespace yak
