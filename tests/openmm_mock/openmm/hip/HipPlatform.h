#pragma once
#include <string>
#include <vector>
#include "openmm/Platform.h"
#include "openmm/hip/HipContext.h"
namespace OpenMM {
class HipPlatform : public Platform {
 public:
  class PlatformData {
   public:
    ~PlatformData() {
      for (HipContext* c : contexts) delete c;
    }
    std::vector<HipContext*> contexts;  // one per device; a force plugin that supports one device uses contexts[0]
  };
  const std::string& getName() const override {
    static const std::string name = "HIP";
    return name;
  }
  double getSpeed() const override { return 100.0; }
};
}  // namespace OpenMM
