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

  /** Points variant (reference projective_icp.cpp:175-232).  Not available yet in this build:
   *  throws std::logic_error. */
  virtual bool estimateTransform(Affine3f& affine, Affine3f& corrected, const Intr& intr, const PointsPyr& vcurr,
                                 const NormalsPyr ncurr, const PointsPyr vprev, const NormalsPyr nprev);

private:
  std::vector<int> iters_;
  float angle_thres_;
  float dist_thres_;
};
}  // namespace cuda
}  // namespace kfusion
