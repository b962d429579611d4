// Host-side plumbing over the C ABI: a per-thread context and RAII device buffers.
// No HIP headers here: device memory goes through ksh_malloc / ksh_ctx_memcpy_* (copies ordered after
// the calling thread's own context: the writer threads of KmerSetSet::Dump do not wait for each other).
#ifndef KSC_CORE_DEVICE_H_
#define KSC_CORE_DEVICE_H_

#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "kmersets_hip.h"

namespace ksc {

// The reference's compute functions cannot fail (asserts only); a failing device
// call here is an exception, never a silent CPU fallback.
inline void Check(int rc) {
  if (rc != KSH_OK) throw std::runtime_error(std::string("kmersets_hip: ") + ksh_last_error());
}

inline int DeviceIndex() {
  const char* e = std::getenv("KSH_DEVICE");
  return e ? std::atoi(e) : 0;
}

// One context per host thread (the reference runs its set operations from pool threads).
inline ksh_ctx* Ctx() {
  struct Holder {
    ksh_ctx* ctx = nullptr;
    Holder() { Check(ksh_ctx_create(DeviceIndex(), nullptr, &ctx)); }
    ~Holder() { ksh_ctx_destroy(ctx); }
  };
  static thread_local Holder holder;
  return holder.ctx;
}

class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  explicit DeviceBuffer(std::size_t bytes) : bytes_(bytes) {
    Check(ksh_malloc(DeviceIndex(), bytes ? bytes : 16, &ptr_));
  }
  DeviceBuffer(const DeviceBuffer& o) : DeviceBuffer(o.bytes_) {
    if (bytes_) Check(ksh_ctx_memcpy_d2d(Ctx(), ptr_, o.ptr_, bytes_));
  }
  DeviceBuffer(DeviceBuffer&& o) noexcept : ptr_(o.ptr_), bytes_(o.bytes_) {
    o.ptr_ = nullptr;
    o.bytes_ = 0;
  }
  DeviceBuffer& operator=(DeviceBuffer o) noexcept {
    std::swap(ptr_, o.ptr_);
    std::swap(bytes_, o.bytes_);
    return *this;
  }
  ~DeviceBuffer() {
    if (ptr_) ksh_free(DeviceIndex(), ptr_);
  }

  void* get() const { return ptr_; }
  std::size_t bytes() const { return bytes_; }

  template <typename T>
  static DeviceBuffer FromHost(const std::vector<T>& v) {
    DeviceBuffer b(v.size() * sizeof(T));
    if (!v.empty()) Check(ksh_ctx_memcpy_h2d(Ctx(), b.ptr_, v.data(), v.size() * sizeof(T)));
    return b;
  }

  template <typename T>
  std::vector<T> ToHost(std::size_t count) const {
    std::vector<T> v(count);
    if (count) Check(ksh_ctx_memcpy_d2h(Ctx(), v.data(), ptr_, count * sizeof(T)));
    return v;
  }

 private:
  void* ptr_ = nullptr;
  std::size_t bytes_ = 0;
};

}  // namespace ksc

#endif
