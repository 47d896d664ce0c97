#pragma once
#include "openmm/KernelImpl.h"
namespace OpenMM {
// reference-counted handle to a KernelImpl, as OpenMM's
class Kernel {
 public:
  Kernel() : impl(nullptr) {}
  explicit Kernel(KernelImpl* impl) : impl(impl) {}
  Kernel(const Kernel& other) : impl(other.impl) {
    if (impl) impl->referenceCount++;
  }
  Kernel& operator=(const Kernel& other) {
    if (other.impl) other.impl->referenceCount++;
    release();
    impl = other.impl;
    return *this;
  }
  ~Kernel() { release(); }
  std::string getName() const { return impl->getName(); }
  KernelImpl& getImpl() { return *impl; }
  template <class T>
  T& getAs() {
    return dynamic_cast<T&>(*impl);
  }

 private:
  void release() {
    if (impl && --impl->referenceCount == 0) delete impl;
    impl = nullptr;
  }
  KernelImpl* impl;
};
}  // namespace OpenMM
