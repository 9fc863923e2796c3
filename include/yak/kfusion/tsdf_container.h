// yak/kfusion/tsdf_container.h -- host copy of a dense TSDF grid, 32 bits per voxel:
// bytes 0-1 IEEE half distance, bytes 2-3 uint16 weight (reference tsdf_container.h:16-56).
// Indexing is 64-bit here (the reference's int index overflows at 2048^3); the int
// accessors of the reference are kept for volumes that fit.
#ifndef YAK_TSDF_CONTAINER_H
#define YAK_TSDF_CONTAINER_H

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <new>
#include <vector>

namespace yak
{
class FusionServer;

namespace detail
{
// std::allocator, except that (a) value-less construction leaves the element alone, so vector(n) does not
// touch its memory, and (b) large blocks are 2 MiB aligned (the .cpp asks the kernel for huge pages on them).
// FusionServer::downloadTSDF fills a container it is about to overwrite anyway: without this the 512 MiB of
// a 512^3 volume are written twice and faulted in 4 KiB at a time (120 ms against 19 ms for the copy itself).
template <typename T>
struct LazyAllocator
{
  typedef T value_type;
  LazyAllocator() {}
  template <typename U>
  LazyAllocator(const LazyAllocator<U>&)
  {
  }
  T* allocate(std::size_t n);
  void deallocate(T* p, std::size_t) { std::free(p); }
  template <typename U>
  void construct(U*)
  {
  }
  template <typename U, typename A, typename... Rest>
  void construct(U* p, A&& a, Rest&&... rest)
  {
    ::new ((void*)p) U(static_cast<A&&>(a), static_cast<Rest&&>(rest)...);
  }
  template <typename U>
  bool operator==(const LazyAllocator<U>&) const
  {
    return true;
  }
  template <typename U>
  bool operator!=(const LazyAllocator<U>&) const
  {
    return false;
  }
};
void* lazy_allocate(std::size_t bytes);  // tsdf_container.cpp
template <typename T>
T* LazyAllocator<T>::allocate(std::size_t n)
{
  return static_cast<T*>(lazy_allocate(n * sizeof(T)));
}
}  // namespace detail

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
  friend class FusionServer;
  struct Uninitialized
  {
  };
  /** Contents unspecified: for a container that is overwritten at once (FusionServer::downloadTSDF). */
  TSDFContainer(int x, int y, int z, Uninitialized);
  static void unpack(std::uint32_t word, std::uint16_t& dist, std::uint16_t& weight)
  {
    std::memcpy(&dist, &word, 2);
    std::memcpy(&weight, ((const char*)(&word)) + 2, 2);
  }
  std::shared_ptr<std::vector<std::uint32_t, detail::LazyAllocator<std::uint32_t>>> data_;
  Vector3i dims_;
};
}  // namespace yak
#endif
