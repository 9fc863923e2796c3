// yak::marchingCubesCPU (reference yak/src/mc/marching_cubes.cpp:158-230): container -> triangles ->
// mesh.  The triple loop over cubes runs on the GPU; the optional cleaning is done here.
#include <yak/mc/marching_cubes.h>

#include <yak/kfusion/cuda/device_memory.hpp>
#include <yakhip.h>

#include <algorithm>
#include <cstring>
#include <thread>

namespace yak
{
PolygonMesh marchingCubesCPU(const TSDFContainer& tsdf, const MarchingCubesParameters& params)
{
  using kfusion::cuda::context;
  using kfusion::cuda::error;
  const Vector3i d = tsdf.dims();
  kfusion::cuda::DeviceMemory vol;
  vol.upload(tsdf.data(), tsdf.size64() * sizeof(std::uint32_t));

  yakhip_volume v;
  v.data = vol.ptr<void>();
  v.dims[0] = d.x(), v.dims[1] = d.y(), v.dims[2] = d.z();
  v.z0 = 0, v.z1 = d.z();
  v.voxel_size[0] = v.voxel_size[1] = v.voxel_size[2] = 1.f;  // vertices are in voxel units times params.scale
  v.trunc_dist = 1.f;
  v.max_weight = 1;

  unsigned long long n = 0;
  if (yakhip_marching_cubes(context(), &v, params.min_weight, params.scale, nullptr, 0, &n) != 0)
    error(yakhip_last_error(), __FILE__, __LINE__, "yakhip_marching_cubes(count)");
  std::vector<float> tri((size_t)n * 9);
  if (n)
  {
    kfusion::cuda::DeviceMemory out((size_t)n * 9 * sizeof(float));
    unsigned long long m = 0;
    if (yakhip_marching_cubes(context(), &v, params.min_weight, params.scale, out.ptr<void>(), n, &m) != 0 || m != n)
      error(yakhip_last_error(), __FILE__, __LINE__, "yakhip_marching_cubes(emit)");
    out.download(tri.data());
  }

  PolygonMesh mesh;
  if (!params.clean)  // marching_cubes.cpp:217-220: three fresh points per triangle
  {
    mesh.cloud.resize((size_t)n * 3);
    mesh.polygons.resize((size_t)n);
    for (size_t i = 0; i < (size_t)n * 3; ++i)
      mesh.cloud[i] = { tri[3 * i], tri[3 * i + 1], tri[3 * i + 2] };
    for (size_t i = 0; i < (size_t)n; ++i)
      mesh.polygons[i] = { (std::uint32_t)(3 * i), (std::uint32_t)(3 * i + 1), (std::uint32_t)(3 * i + 2) };
    return mesh;
  }

  // :204-216 -- exact point merging (bit patterns), then triangles with a repeated index are dropped.
  // Corners are sorted as packed 16-byte records {x, y, z bits, corner} (contiguous: a sort through an index
  // array spends its time on cache misses); any strict order on the bit patterns groups equal points.
  const size_t corners = (size_t)n * 3;
  struct Corner
  {
    std::uint32_t x, y, z, c;
  };
  std::vector<Corner> order(corners);
  for (size_t i = 0; i < corners; ++i)
  {
    std::memcpy(&order[i], &tri[i * 3], 12);
    order[i].c = (std::uint32_t)i;
  }
  const auto less = [](const Corner& a, const Corner& b) {
    if (a.x != b.x)
      return a.x < b.x;
    if (a.y != b.y)
      return a.y < b.y;
    if (a.z != b.z)
      return a.z < b.z;
    return a.c < b.c;
  };
  // four sorted quarters on four threads, merged pairwise: the sort is most of this routine's time
  const size_t parts = corners >= (1u << 16) ? 4 : 1;
  if (parts == 1)
    std::sort(order.begin(), order.end(), less);
  else
  {
    size_t cut[5];
    for (size_t k = 0; k <= 4; ++k)
      cut[k] = corners * k / 4;
    std::vector<std::thread> pool;
    for (size_t k = 0; k < 4; ++k)
      pool.emplace_back([&, k] { std::sort(order.begin() + cut[k], order.begin() + cut[k + 1], less); });
    for (auto& t : pool)
      t.join();
    std::thread left([&] { std::inplace_merge(order.begin(), order.begin() + cut[1], order.begin() + cut[2], less); });
    std::inplace_merge(order.begin() + cut[2], order.begin() + cut[3], order.end(), less);
    left.join();
    std::inplace_merge(order.begin(), order.begin() + cut[2], order.end(), less);
  }
  std::vector<std::uint32_t> index(corners);
  mesh.cloud.reserve(corners / 5);
  for (size_t i = 0; i < corners; ++i)
  {
    if (i == 0 || order[i].x != order[i - 1].x || order[i].y != order[i - 1].y || order[i].z != order[i - 1].z)
    {
      const float* p = &tri[(size_t)order[i].c * 3];
      mesh.cloud.push_back({ p[0], p[1], p[2] });
    }
    index[order[i].c] = (std::uint32_t)(mesh.cloud.size() - 1);
  }
  mesh.polygons.reserve((size_t)n);
  for (size_t i = 0; i < (size_t)n; ++i)
  {
    const std::uint32_t a = index[3 * i], b = index[3 * i + 1], c = index[3 * i + 2];
    if (a != b && b != c && a != c)
      mesh.polygons.push_back({ a, b, c });
  }
  return mesh;
}
}  // namespace yak
