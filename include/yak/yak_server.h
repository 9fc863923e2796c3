// yak/yak_server.h -- yak::FusionServer: feeds depth frames with externally known poses into KinFu
// and hands back the fused volume (reference yak/include/yak/yak_server.h:24-60,
// yak/src/yak_server.cpp:4-110).  Headless: the reference opens an OpenCV window in the constructor and
// shows a rendered view after every frame (yak_server.cpp:10-12,36,90-98).  Here display() renders the
// same view on the GPU and keeps it in host memory (view()); fuse() does so after every frame only when
// the environment variable YAK_HIP_DISPLAY is set to a non-zero value.
#ifndef YAK_OFFLINE_FUSION_SERVER_H_
#define YAK_OFFLINE_FUSION_SERVER_H_

#include "yak/kfusion/kinfu.hpp"
#include "yak/kfusion/kinfu_sharded.hpp"
#include "yak/kfusion/tsdf_container.h"
#include "yak/mc/marching_cubes.h"

#if defined(YAK_USE_EIGEN) && __has_include(<Eigen/Geometry>)
#include <Eigen/Geometry>
#else
namespace Eigen
{
typedef yak::compat::EigenAffine3f Affine3f;  // stand-in with inverse(), operator*, Identity(), (r,c)
}
#endif

namespace yak
{
/** Not thread safe (same contract as the reference, yak_server.h:22).
 *  Multi-GPU: when the environment variable YAK_HIP_DEVICES lists more than one device ("0,1,2,3,4,5,6,7") the
 *  volume is spread over them as z-slabs (kfusion::ShardedKinFu: one rank per listed GPU, RCCL over xGMI between
 *  them); fuse / reset / downloadTSDF / uploadTSDF keep their meaning, results are those of one GPU (bit for bit with
 *  supplied poses).  extractMesh and display() are one-GPU features: with several devices extractMesh goes through
 *  downloadTSDF + the device routine, display() shows rank 0's model maps. */
class FusionServer
{
public:
  FusionServer(const kfusion::KinFuParams& params, const Eigen::Affine3f& world_to_volume);

  /** depth_data: 16-bit unsigned millimetres, rows x cols as in params (cv::Mat 16UC1 in the reference). */
  bool fuse(const cv::Mat& depth_data, const Eigen::Affine3f& world_to_camera);
  bool reset();
  bool resetWithNewParams(const kfusion::KinFuParams& params);
  yak::TSDFContainer downloadTSDF();
  /** The inverse of downloadTSDF (no counterpart in the reference, whose only way in is fuse()): replaces the
   *  device volume by `volume`, e.g. to resume a fusion that was saved.  Returns false, touching nothing,
   *  if the dimensions differ from the server's volume.  Tracking continues from the poses it had. */
  bool uploadTSDF(const yak::TSDFContainer& volume);
  /** marchingCubesCPU(downloadTSDF(), params) without the round trip through host memory: the mesh is extracted
   *  (and, with params.clean, welded) from the volume where it lies (no counterpart in the reference). */
  yak::PolygonMesh extractMesh(const yak::MarchingCubesParameters& params = {});
  /** Renders the tracker's current model view (Phong | normal colours, rows x 2 cols) and downloads it
   *  (yak_server.cpp:100-104); the reference then shows it in a window. */
  void display();
  /** The same, ray-cast from `pose` (yak_server.cpp:106-110). */
  void display(const Eigen::Affine3f& pose);
  /** The last view display() produced: rows x cols pixels of 4 bytes {b, g, r, 0}, tightly packed
   *  (what the reference hands to cv::imshow, yak_server.cpp:92-96).  Empty before the first display(). */
  const std::vector<unsigned char>& view(int& rows, int& cols) const;

private:
  bool step(const Eigen::Affine3f& current_pose, const Eigen::Affine3f& last_pose, const cv::Mat& depth);

  kfusion::KinFu::Ptr kinfu_;            // one GPU
  kfusion::ShardedKinFu::Ptr sharded_;   // several (YAK_HIP_DEVICES); exactly one of the two is set
  void create(const kfusion::KinFuParams& params);
  void downloadView();

  kfusion::cuda::Depth depthDevice_;
  Eigen::Affine3f volume_to_world_;
  Eigen::Affine3f last_camera_pose_;
  kfusion::cuda::Image viewDevice_;
  std::vector<unsigned char> viewHost_;
  int viewRows_ = 0, viewCols_ = 0;  // of viewHost_ when the view came from a sharded server
  bool display_every_frame_;
};
}  // namespace yak
#endif
