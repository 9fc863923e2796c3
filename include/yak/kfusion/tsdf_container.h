// yak/kfusion/tsdf_container.h -- host copy of a dense TSDF grid, 32 bits per voxel:
// bytes 0-1 IEEE half distance, bytes 2-3 uint16 weight (reference tsdf_container.h:16-56).
// Indexing is 64-bit here (the reference's int index overflows at 2048^3); the int
// accessors of the reference are kept for volumes that fit.
#ifndef YAK_TSDF_CONTAINER_H
#define YAK_TSDF_CONTAINER_H

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace yak
{
struct Vector3i
{
  int v[3];
  int x() const { return v[0]; }
  int y() const { return v[1]; }
  int z() const { return v[2]; }
  int operator()(int i) const { return v[i]; }
};

class TSDFContainer
{
public:
  TSDFContainer(int x, int y, int z);

  /** Distance as IEEE half bits (the reference hands out half_float::half, which has this layout). */
  void read(std::size_t index, std::uint16_t& dist_half_bits, std::uint16_t& weight) const
  {
    unpack((*data_)[index], dist_half_bits, weight);
  }
  /** Distance widened to float (half -> float is exact). */
  float distance(std::size_t index) const;

  int toIndex(int x, int y, int z) const { return dims_.v[0] * dims_.v[1] * z + dims_.v[0] * y + x; }
  std::size_t toIndex64(int x, int y, int z) const
  {
    return ((std::size_t)z * (std::size_t)dims_.v[1] + (std::size_t)y) * (std::size_t)dims_.v[0] + (std::size_t)x;
  }
  void fromIndex(int index, int& x, int& y, int& z) const
  {
    z = index / (dims_.v[0] * dims_.v[1]);
    index -= z * dims_.v[0] * dims_.v[1];
    y = index / dims_.v[0];
    x = index % dims_.v[0];
  }

  std::uint32_t* data() { return data_->data(); }
  const std::uint32_t* data() const { return data_->data(); }

  int size() const { return dims_.v[0] * dims_.v[1] * dims_.v[2]; }
  std::size_t size64() const { return data_->size(); }
  Vector3i dims() const { return dims_; }

private:
  static void unpack(std::uint32_t word, std::uint16_t& dist, std::uint16_t& weight)
  {
    std::memcpy(&dist, &word, 2);
    std::memcpy(&weight, ((const char*)(&word)) + 2, 2);
  }
  std::shared_ptr<std::vector<std::uint32_t>> data_;
  Vector3i dims_;
};
}  // namespace yak
#endif
