// What a yak_ros node does with the package: KinFuParams, a FusionServer, the meshing parameters.
#include <yak/mc/marching_cubes.h>
#include <yak/yak_server.h>

#include <cstdio>

int main()
{
  kfusion::KinFuParams p = kfusion::KinFuParams::default_params();
  yak::MarchingCubesParameters mc;
  std::printf("yak package: %d x %d x %d voxels of %.6f m, mc scale %.1f\n", p.volume_dims[0], p.volume_dims[1], p.volume_dims[2],
              p.volume_resolution, mc.scale);
  if (kfusion::cuda::getCudaEnabledDeviceCount() == 0)
    return 0;  // built and linked; nothing to run without a GPU
  yak::FusionServer server(p, Eigen::Affine3f::Identity());
  return server.reset() ? 0 : 1;
}
