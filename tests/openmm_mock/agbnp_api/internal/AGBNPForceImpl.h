// Restatement of the plugin's ForceImpl (openmmapi/include/internal/AGBNPForceImpl.h:23-43,
// openmmapi/src/AGBNPForceImpl.cpp:21-46): creates the platform's kernel, forwards evaluation and parameter updates,
// honours the force-group mask.
#pragma once
#include <map>
#include <string>
#include <vector>
#include "openmm/Kernel.h"
#include "openmm/internal/ContextImpl.h"
#include "openmm/internal/ForceImpl.h"

namespace AGBNPPlugin {
class AGBNPForce;
class AGBNPForceImpl : public OpenMM::ForceImpl {
 public:
  explicit AGBNPForceImpl(const AGBNPForce& owner) : owner(owner) {}
  inline void initialize(OpenMM::ContextImpl& context) override;
  const AGBNPForce& getOwner() const override { return owner; }
  void updateContextState(OpenMM::ContextImpl&, bool&) override {}
  inline double calcForcesAndEnergy(OpenMM::ContextImpl& context, bool includeForces, bool includeEnergy, int groups) override;
  std::map<std::string, double> getDefaultParameters() override { return {}; }
  inline std::vector<std::string> getKernelNames() override;
  inline void updateParametersInContext(OpenMM::ContextImpl& context);

 private:
  const AGBNPForce& owner;
  OpenMM::Kernel kernel;
};
}  // namespace AGBNPPlugin

