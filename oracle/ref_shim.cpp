// ref_shim.cpp -- C entry points onto the two pieces of the reference that compile stand-alone.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/libyakref.so, from the
// reference sources WHERE THEY LIE under /root/reference (nothing is copied into this repo):
//   yak/include/yak/kfusion/half.hpp            (vendored half_float 1.12, host half type)
//   yak/src/mc/marching_cubes_tables.cpp        (edgeFlags / triangleTable / edgeConnections)
// Everything else on the hot path is CUDA-only or needs OpenCV/Eigen/PCL/VTK, none of which
// exist in this image, so it is unbuildable here ("Oracle").
//
// Used to (1) pin yko_half2float against the reference's own half type for all 65536 codes,
// (2) pin the voxel wire format the way test/marching_cubes_tests.cpp:46-79 does it,
// (3) give the marching-cubes oracle the reference's own tables.
#include <cstdint>
#include <cstring>

#include "yak/kfusion/half.hpp"
#include "marching_cubes_tables.h"

extern "C" {

float yakref_half2float(uint16_t bits)
{
  half_float::half h;
  std::memcpy(&h, &bits, 2);
  return float(h);
}

// half.hpp's float->half constructor (truncating by default, half.hpp:214-215).
uint16_t yakref_float2half_ctor(float f)
{
  half_float::half h(f);
  uint16_t bits;
  std::memcpy(&bits, &h, 2);
  return bits;
}

// The test's pack(): test/marching_cubes_tests.cpp:46-50
uint32_t yakref_pack(float dist, uint16_t weight)
{
  half_float::half h(dist);
  uint32_t data = 0;
  std::memcpy(&data, &h, 2);
  std::memcpy(((char*)(&data)) + 2, &weight, 2);
  return data;
}

// TSDFContainer::unpack: include/yak/kfusion/tsdf_container.h:45-49
void yakref_unpack(uint32_t data, float* dist, uint16_t* weight)
{
  half_float::half h;
  std::memcpy(&h, &data, 2);
  std::memcpy(weight, ((char*)(&data)) + 2, 2);
  *dist = float(h);
}

const int* yakref_edge_flags() { return yak::edgeFlags; }
const int* yakref_triangle_table() { return &yak::triangleTable[0][0]; }
const int* yakref_edge_connections() { return &yak::edgeConnections[0][0]; }
}
