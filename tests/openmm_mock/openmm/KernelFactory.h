#pragma once
#include <string>
#include "openmm/KernelImpl.h"
namespace OpenMM {
class ContextImpl;
class KernelFactory {
 public:
  virtual ~KernelFactory() {}
  virtual KernelImpl* createKernelImpl(std::string name, const Platform& platform, ContextImpl& context) const = 0;
};
}  // namespace OpenMM
