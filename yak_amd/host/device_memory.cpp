// Device buffers + context plumbing of the host layer (reference yak/src/kfusion/device_memory.cpp,
// yak/src/kfusion/core.cpp:8-32), over the C ABI.
#include <yak/kfusion/cuda/device_memory.hpp>
#include <yak/kfusion/kinfu.hpp>

#include <yakhip.h>

#include <cstdio>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>

namespace kfusion
{
namespace cuda
{
// reference device_memory.cpp:7-11 prints and exits the process; a library must not
void error(const char* error_string, const char* file, const int line, const char* func)
{
  std::ostringstream os;
  os << "KinFu/HIP error: " << error_string << " (" << file << ":" << line << (func && *func ? ", " : "")
     << (func ? func : "") << ")";
  throw std::runtime_error(os.str());
}

#define YAK_CHECK(call)                                                   \
  do                                                                      \
  {                                                                       \
    if ((call) != 0)                                                      \
      ::kfusion::cuda::error(yakhip_last_error(), __FILE__, __LINE__, #call); \
  } while (0)

namespace
{
// per thread: a multi-GPU process drives every GPU (rank) from its own thread (kfusion::ShardedKinFu)
thread_local int g_device = 0;
thread_local yakhip_ctx* g_context = nullptr;  // setContext(): this thread's own context instead of the device's shared one
thread_local Shard g_shard = { nullptr, 0, 1 };
std::mutex g_contexts_mutex;
std::map<int, yakhip_ctx*>& contexts()
{
  static std::map<int, yakhip_ctx*> m;
  return m;
}
}  // namespace

void setContext(yakhip_ctx* ctx) { g_context = ctx; }
void setShard(const Shard& shard) { g_shard = shard; }
const Shard& currentShard() { return g_shard; }

yakhip_ctx* context()
{
  if (g_context)
    return g_context;
  std::lock_guard<std::mutex> lock(g_contexts_mutex);
  auto& m = contexts();
  auto it = m.find(g_device);
  if (it != m.end())
    return it->second;
  yakhip_ctx* c = nullptr;
  YAK_CHECK(yakhip_ctx_create(&c, g_device));
  m[g_device] = c;
  return c;
}

int getCudaEnabledDeviceCount()
{
  int n = 0;
  if (yakhip_device_count(&n) != 0)
    return 0;
  return n;
}
void setDevice(int device) { g_device = device; }
std::string getDeviceName(int device)
{
  char buf[256];
  YAK_CHECK(yakhip_device_name(device, buf, sizeof(buf)));
  return buf;
}
bool checkIfPreFermiGPU(int) { return false; }
void printShortCudaDeviceInfo(int device) { std::printf("[yak] device %d: %s\n", device, getDeviceName(device).c_str()); }
void printCudaDeviceInfo(int device) { printShortCudaDeviceInfo(device); }

// ---- DeviceMemory ------------------------------------------------------------------------------
namespace
{
std::shared_ptr<void> device_alloc(size_t bytes, void** out)
{
  yakhip_ctx* c = context();
  void* p = nullptr;
  YAK_CHECK(yakhip_malloc(c, &p, bytes));
  *out = p;
  return std::shared_ptr<void>(p, [c](void* q) { yakhip_free(c, q); });
}
}  // namespace

DeviceMemory::DeviceMemory() : data_(nullptr), sizeBytes_(0) {}
DeviceMemory::~DeviceMemory() {}
DeviceMemory::DeviceMemory(size_t sizeBytes_arg) : data_(nullptr), sizeBytes_(0) { create(sizeBytes_arg); }
DeviceMemory::DeviceMemory(void* ptr_arg, size_t sizeBytes_arg) : data_(ptr_arg), sizeBytes_(sizeBytes_arg) {}
DeviceMemory::DeviceMemory(const DeviceMemory& o) : data_(o.data_), sizeBytes_(o.sizeBytes_), owner_(o.owner_) {}
DeviceMemory& DeviceMemory::operator=(const DeviceMemory& o)
{
  data_ = o.data_;
  sizeBytes_ = o.sizeBytes_;
  owner_ = o.owner_;
  return *this;
}
void DeviceMemory::create(size_t sizeBytes_arg)
{
  if (sizeBytes_arg == sizeBytes_ && data_)  // reference device_memory.cpp:87-88: same size, keep
    return;
  release();
  if (sizeBytes_arg > 0)
  {
    owner_ = device_alloc(sizeBytes_arg, &data_);
    sizeBytes_ = sizeBytes_arg;
  }
}
void DeviceMemory::release()
{
  owner_.reset();
  data_ = nullptr;
  sizeBytes_ = 0;
}
void DeviceMemory::copyTo(DeviceMemory& other) const
{
  if (empty())
  {
    other.release();
    return;
  }
  other.create(sizeBytes_);
  YAK_CHECK(yakhip_copy(context(), other.data_, data_, sizeBytes_));  // device_memory.cpp:112
}
void DeviceMemory::upload(const void* host_ptr_arg, size_t sizeBytes_arg)
{
  create(sizeBytes_arg);
  YAK_CHECK(yakhip_upload(context(), data_, host_ptr_arg, sizeBytes_arg));
}
void DeviceMemory::download(void* host_ptr_arg) const
{
  YAK_CHECK(yakhip_download(context(), host_ptr_arg, data_, sizeBytes_));
}
void DeviceMemory::swap(DeviceMemory& o)
{
  std::swap(data_, o.data_);
  std::swap(sizeBytes_, o.sizeBytes_);
  owner_.swap(o.owner_);
}
bool DeviceMemory::empty() const { return !data_; }
size_t DeviceMemory::sizeBytes() const { return sizeBytes_; }
bool DeviceMemory::shared() const { return data_ && owner_.use_count() != 1; }

// ---- DeviceMemory2D ----------------------------------------------------------------------------
DeviceMemory2D::DeviceMemory2D() : data_(nullptr), step_(0), colsBytes_(0), rows_(0) {}
DeviceMemory2D::~DeviceMemory2D() {}
DeviceMemory2D::DeviceMemory2D(int rows_arg, int colsBytes_arg) : data_(nullptr), step_(0), colsBytes_(0), rows_(0)
{
  create(rows_arg, colsBytes_arg);
}
DeviceMemory2D::DeviceMemory2D(int rows_arg, int colsBytes_arg, void* data_arg, size_t step_arg)
  : data_(data_arg), step_(step_arg), colsBytes_(colsBytes_arg), rows_(rows_arg)
{
}
DeviceMemory2D::DeviceMemory2D(const DeviceMemory2D& o)
  : data_(o.data_), step_(o.step_), colsBytes_(o.colsBytes_), rows_(o.rows_), owner_(o.owner_)
{
}
DeviceMemory2D& DeviceMemory2D::operator=(const DeviceMemory2D& o)
{
  data_ = o.data_;
  step_ = o.step_;
  colsBytes_ = o.colsBytes_;
  rows_ = o.rows_;
  owner_ = o.owner_;
  return *this;
}
void DeviceMemory2D::create(int rows_arg, int colsBytes_arg)
{
  if (colsBytes_ == colsBytes_arg && rows_ == rows_arg && data_)
    return;
  release();
  if (rows_arg > 0 && colsBytes_arg > 0)
  {
    yakhip_ctx* c = context();
    void* p = nullptr;
    size_t pitch = 0;
    YAK_CHECK(yakhip_malloc_pitch(c, &p, &pitch, (size_t)colsBytes_arg, (size_t)rows_arg));
    owner_ = std::shared_ptr<void>(p, [c](void* q) { yakhip_free(c, q); });
    data_ = p;
    step_ = pitch;
    colsBytes_ = colsBytes_arg;
    rows_ = rows_arg;
  }
}
void DeviceMemory2D::release()
{
  owner_.reset();
  data_ = nullptr;
  step_ = 0;
  colsBytes_ = 0;
  rows_ = 0;
}
void DeviceMemory2D::copyTo(DeviceMemory2D& other) const
{
  if (empty())
  {
    other.release();
    return;
  }
  other.create(rows_, colsBytes_);
  YAK_CHECK(yakhip_copy_2d(context(), other.data_, other.step_, data_, step_, (size_t)colsBytes_, (size_t)rows_));  // :240
}
void DeviceMemory2D::upload(const void* host_ptr_arg, size_t host_step_arg, int rows_arg, int colsBytes_arg)
{
  create(rows_arg, colsBytes_arg);
  YAK_CHECK(yakhip_upload_2d(context(), data_, step_, host_ptr_arg, host_step_arg, (size_t)colsBytes_arg,
                             (size_t)rows_arg));
}
void DeviceMemory2D::download(void* host_ptr_arg, size_t host_step_arg) const
{
  YAK_CHECK(yakhip_download_2d(context(), host_ptr_arg, host_step_arg, data_, step_, (size_t)colsBytes_,
                               (size_t)rows_));
}
void DeviceMemory2D::swap(DeviceMemory2D& o)
{
  std::swap(data_, o.data_);
  std::swap(step_, o.step_);
  std::swap(colsBytes_, o.colsBytes_);
  std::swap(rows_, o.rows_);
  owner_.swap(o.owner_);
}
bool DeviceMemory2D::empty() const { return !data_; }
int DeviceMemory2D::colsBytes() const { return colsBytes_; }
int DeviceMemory2D::rows() const { return rows_; }
size_t DeviceMemory2D::step() const { return step_; }
}  // namespace cuda
}  // namespace kfusion
