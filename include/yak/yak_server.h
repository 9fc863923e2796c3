// yak/yak_server.h -- yak::FusionServer: feeds depth frames with externally known poses into KinFu
// and hands back the fused volume (reference yak/include/yak/yak_server.h:24-60,
// yak/src/yak_server.cpp:4-110).  Headless: the reference opens an OpenCV window in the constructor
// and renders after every frame (yak_server.cpp:10-12,36); display() is a no-op here.
#ifndef YAK_OFFLINE_FUSION_SERVER_H_
#define YAK_OFFLINE_FUSION_SERVER_H_

#include "yak/kfusion/kinfu.hpp"
#include "yak/kfusion/tsdf_container.h"

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
/** Not thread safe (same contract as the reference, yak_server.h:22). */
class FusionServer
{
public:
  FusionServer(const kfusion::KinFuParams& params, const Eigen::Affine3f& world_to_volume);

  /** depth_data: 16-bit unsigned millimetres, rows x cols as in params (cv::Mat 16UC1 in the reference). */
  bool fuse(const cv::Mat& depth_data, const Eigen::Affine3f& world_to_camera);
  bool reset();
  bool resetWithNewParams(const kfusion::KinFuParams& params);
  yak::TSDFContainer downloadTSDF();
  void display();
  void display(const Eigen::Affine3f& pose);

private:
  bool step(const Eigen::Affine3f& current_pose, const Eigen::Affine3f& last_pose, const cv::Mat& depth);

  kfusion::KinFu::Ptr kinfu_;
  kfusion::cuda::Depth depthDevice_;
  Eigen::Affine3f volume_to_world_;
  Eigen::Affine3f last_camera_pose_;
};
}  // namespace yak
#endif
