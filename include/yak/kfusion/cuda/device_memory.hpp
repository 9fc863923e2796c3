// yak/kfusion/cuda/device_memory.hpp -- reference-counted device buffers with the reference's class
// and method names (reference yak/include/yak/kfusion/cuda/device_memory.hpp:23-223), backed by HIP
// through the C ABI (yakhip_malloc / yakhip_upload / ...).  Ownership is a std::shared_ptr here
// instead of the reference's hand-rolled CV_XADD counter (device_memory.cpp:13-48); copies alias,
// release() drops one reference, the last owner frees.
#pragma once
#include <cstddef>
#include <memory>

#include <yak/kfusion/exports.hpp>

struct yakhip_ctx;
struct yakhip_comm;

namespace kfusion
{
namespace cuda
{
/** Error sink of the GPU layer (reference device_memory.hpp:13).  The reference prints and calls
 *  exit(0) (device_memory.cpp:7-11); this one throws std::runtime_error. */
KF_EXPORTS void error(const char* error_string, const char* file, const int line, const char* func = "");

/** The context (device + stream) the host classes submit to: one per device, created on first use.
 *  setDevice() (kinfu.hpp) selects which one is current for the calling process. */
KF_EXPORTS yakhip_ctx* context();
/** Makes `ctx` the context of the CALLING THREAD (nullptr: back to the device's shared one).  No counterpart in the
 *  reference (one implicit CUDA device per process): a multi-GPU process drives each GPU from its own thread. */
KF_EXPORTS void setContext(yakhip_ctx* ctx);
/** The calling thread's place in a z-slab sharded volume: its communicator, rank and world size.
 *  world == 1 (the default) is the reference's one-GPU behaviour.  TsdfVolume then holds only its own layers plus
 *  the ray-cast halo, raycast() composites over the ranks, ProjectiveICP sums its normal equations over them. */
struct Shard
{
  yakhip_comm* comm;
  int rank, world;
};
KF_EXPORTS void setShard(const Shard& shard);
KF_EXPORTS const Shard& currentShard();

class KF_EXPORTS DeviceMemory
{
public:
  DeviceMemory();
  ~DeviceMemory();
  explicit DeviceMemory(size_t sizeBytes_arg);
  DeviceMemory(void* ptr_arg, size_t sizeBytes_arg);  // wraps user memory, never freed here
  DeviceMemory(const DeviceMemory& other_arg);
  DeviceMemory& operator=(const DeviceMemory& other_arg);

  void create(size_t sizeBytes_arg);
  void release();
  void copyTo(DeviceMemory& other) const;
  void upload(const void* host_ptr_arg, size_t sizeBytes_arg);
  void download(void* host_ptr_arg) const;
  void swap(DeviceMemory& other_arg);

  template <class T>
  T* ptr()
  {
    return (T*)data_;
  }
  template <class T>
  const T* ptr() const
  {
    return (const T*)data_;
  }
  bool empty() const;
  size_t sizeBytes() const;
  /** True if somebody else can reach this storage too: another DeviceMemory aliases it, or it wraps user memory.
   *  No counterpart in the reference (its counter is private, device_memory.hpp:93). */
  bool shared() const;

private:
  void* data_;
  size_t sizeBytes_;
  std::shared_ptr<void> owner_;  // null for wrapped user memory
};

class KF_EXPORTS DeviceMemory2D
{
public:
  DeviceMemory2D();
  ~DeviceMemory2D();
  DeviceMemory2D(int rows_arg, int colsBytes_arg);
  DeviceMemory2D(int rows_arg, int colsBytes_arg, void* data_arg, size_t step_arg);  // wraps user memory
  DeviceMemory2D(const DeviceMemory2D& other_arg);
  DeviceMemory2D& operator=(const DeviceMemory2D& other_arg);

  void create(int rows_arg, int colsBytes_arg);
  void release();
  void copyTo(DeviceMemory2D& other) const;
  void upload(const void* host_ptr_arg, size_t host_step_arg, int rows_arg, int colsBytes_arg);
  void download(void* host_ptr_arg, size_t host_step_arg) const;
  void swap(DeviceMemory2D& other_arg);

  template <class T>
  T* ptr(int y_arg = 0)
  {
    return (T*)((char*)data_ + (size_t)y_arg * step_);
  }
  template <class T>
  const T* ptr(int y_arg = 0) const
  {
    return (const T*)((const char*)data_ + (size_t)y_arg * step_);
  }
  bool empty() const;
  int colsBytes() const;
  int rows() const;
  size_t step() const;  // bytes, always

private:
  void* data_;
  size_t step_;
  int colsBytes_;
  int rows_;
  std::shared_ptr<void> owner_;
};
}  // namespace cuda
}  // namespace kfusion
