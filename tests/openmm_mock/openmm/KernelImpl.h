#pragma once
#include <string>
namespace OpenMM {
class Platform;
class KernelImpl {
 public:
  KernelImpl(std::string name, const Platform& platform) : name(name), platform(&platform), referenceCount(1) {}
  virtual ~KernelImpl() {}
  std::string getName() const { return name; }
  const Platform& getPlatform() const { return *platform; }

 private:
  friend class Kernel;
  std::string name;
  const Platform* platform;
  int referenceCount;
};
}  // namespace OpenMM
