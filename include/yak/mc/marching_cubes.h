// yak/mc/marching_cubes.h -- surface extraction from a TSDFContainer
// (reference yak/include/yak/mc/marching_cubes.h:7-22; implementation yak/src/mc/marching_cubes.cpp).
//
// Same entry point, same parameters.  The reference returns a pcl::PolygonMesh; PCL is not a dependency
// here, so the mesh comes back as yak::PolygonMesh -- vertex positions plus index triples, which is what
// the reference's callers read out of the PCL type (cloud + polygons).  The name says CPU because the
// reference's does; the triangles are produced on the GPU (yakhip_marching_cubes), there is no CPU
// fallback.
#ifndef YAK_MARCHING_CUBES_H
#define YAK_MARCHING_CUBES_H

#include <array>
#include <cstdint>
#include <vector>

#include <yak/kfusion/exports.hpp>
#include <yak/kfusion/tsdf_container.h>

namespace yak
{
struct MarchingCubesParameters
{
  // Scale factor applied to vertices.  Usually the physical size of a single voxel.
  double scale{ 1.0 };
  // Minimum weight of a voxel to not be dropped in the meshing process.
  int min_weight{ 4 };
  // Merge identical vertices and drop triangles that collapse (what the reference asks of
  // vtkCleanPolyData, marching_cubes.cpp:204-216).
  bool clean{ false };
};

struct PolygonMesh
{
  std::vector<std::array<float, 3>> cloud;             // vertices
  std::vector<std::array<std::uint32_t, 3>> polygons;  // triangles, indices into cloud
};

KF_EXPORTS PolygonMesh marchingCubesCPU(const TSDFContainer& tsdf, const MarchingCubesParameters& params = {});
/** The same mesh from a volume that already lies in device memory (dim_x * dim_y * dim_z packed voxels, x fastest),
 *  e.g. kfusion::cuda::TsdfVolume::data(): no counterpart in the reference, whose only way to a mesh is
 *  downloadTSDF() + marchingCubesCPU() -- 512 MiB down and up again at 512^3. */
KF_EXPORTS PolygonMesh marchingCubesDevice(const void* volume_dev, int dim_x, int dim_y, int dim_z,
                                           const MarchingCubesParameters& params = {});
}  // namespace yak

#endif
