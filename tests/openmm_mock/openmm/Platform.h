#pragma once
#include <map>
#include <string>
#include <vector>
#include "openmm/Kernel.h"
#include "openmm/KernelFactory.h"
#include "openmm/OpenMMException.h"
namespace OpenMM {
class Platform {
 public:
  virtual ~Platform() {
    for (auto& f : kernelFactories) delete f.second;
  }
  virtual const std::string& getName() const = 0;
  virtual double getSpeed() const { return 1.0; }
  void registerKernelFactory(const std::string& name, KernelFactory* factory) {
    auto old = kernelFactories.find(name);
    if (old != kernelFactories.end()) delete old->second;
    kernelFactories[name] = factory;
  }
  bool supportsKernels(const std::vector<std::string>& kernelNames) const {
    for (auto& k : kernelNames)
      if (kernelFactories.find(k) == kernelFactories.end()) return false;
    return true;
  }
  Kernel createKernel(const std::string& name, ContextImpl& context) const {
    auto f = kernelFactories.find(name);
    if (f == kernelFactories.end()) throw OpenMMException("Called createKernel() on a Platform which does not support the requested kernel");
    return Kernel(f->second->createKernelImpl(name, *this, context));
  }
  static void registerPlatform(Platform* platform) { registry().push_back(platform); }
  static int getNumPlatforms() { return (int)registry().size(); }
  static Platform& getPlatform(int index) { return *registry().at(index); }
  static Platform& getPlatformByName(const std::string& name) {
    for (Platform* p : registry())
      if (p->getName() == name) return *p;
    throw OpenMMException("There is no registered Platform called \"" + name + "\"");
  }

 private:
  static std::vector<Platform*>& registry() {
    static std::vector<Platform*> platforms;
    return platforms;
  }
  std::map<std::string, KernelFactory*> kernelFactories;
};
}  // namespace OpenMM
