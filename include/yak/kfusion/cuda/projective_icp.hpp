// yak/kfusion/cuda/projective_icp.hpp -- projective point-to-plane ICP front
// (reference yak/include/yak/kfusion/cuda/projective_icp.hpp:11-62).
#pragma once
#include <vector>

#include <yak/kfusion/types.hpp>

namespace kfusion
{
namespace cuda
{
class KF_EXPORTS ProjectiveICP
{
public:
  enum
  {
    MAX_PYRAMID_LEVELS = 4
  };
  typedef std::vector<Depth> DepthPyr;
  typedef std::vector<Cloud> PointsPyr;
  typedef std::vector<Normals> NormalsPyr;

  ProjectiveICP();
  virtual ~ProjectiveICP();

  float getDistThreshold() const;
  void setDistThreshold(float distance);
  float getAngleThreshold() const;
  void setAngleThreshold(float angle);
  void setIterationsNum(const std::vector<int>& iters);
  int getUsedLevelsNum() const;

  /** Points variant (reference projective_icp.cpp:175-232): refines `affine` (current -> previous
   *  camera) coarse to fine against the previous frame's maps; false when the normal matrix is
   *  (numerically) singular.  The Depth variant belongs to the reference's disabled USE_DEPTH build
   *  (internal.hpp:8) and the Frame overload is a CV_Assert(!"Not implemented") there (:106-123). */
  virtual bool estimateTransform(const Affine3f& affine, Affine3f& correctedAffine, const Intr& intr,
                                 const PointsPyr& vcurr, const NormalsPyr ncurr, const PointsPyr vprev,
                                 const NormalsPyr nprev);

private:
  std::vector<int> iters_;
  float angle_thres_;
  float dist_thres_;
};
}  // namespace cuda
}  // namespace kfusion
