#pragma once
#include <vector>
#include "openmm/Platform.h"
#include "openmm/System.h"
#include "openmm/internal/ForceImpl.h"
namespace OpenMM {
class Context;
// what a platform kernel sees of a Context: the System, the Platform and the platform's opaque per-context data
class ContextImpl {
 public:
  ContextImpl(Context& owner, const System& system, Platform& platform, void* platformData)
      : owner(owner), system(system), platform(&platform), platformData(platformData) {}
  ~ContextImpl() {
    for (ForceImpl* f : forceImpls) delete f;
  }
  Context& getOwner() { return owner; }
  const System& getSystem() const { return system; }
  Platform& getPlatform() { return *platform; }
  void* getPlatformData() { return platformData; }
  const void* getPlatformData() const { return platformData; }
  void setPlatformData(void* data) { platformData = data; }
  std::vector<ForceImpl*>& getForceImpls() { return forceImpls; }
  // Context creation: every Force of the System gets its ForceImpl, which creates and initialises its kernels
  void initialize() {
    for (int i = 0; i < system.getNumForces(); i++) {
      forceImpls.push_back(system.getForce(i).createImpl());
      forceImpls.back()->initialize(*this);
    }
  }
  double calcForcesAndEnergy(bool includeForces, bool includeEnergy, int groups = 0xFFFFFFFF) {
    double e = 0.0;
    for (ForceImpl* f : forceImpls) e += f->calcForcesAndEnergy(*this, includeForces, includeEnergy, groups);
    return e;
  }

 private:
  Context& owner;
  const System& system;
  Platform* platform;
  void* platformData;
  std::vector<ForceImpl*> forceImpls;
};
}  // namespace OpenMM
