// Restatement of the plugin's kernel interface (openmmapi/include/AGBNPKernels.h:19-47): the abstract class every
// platform implements.
#pragma once
#include <string>
#include "AGBNPForce.h"
#include "openmm/KernelImpl.h"
#include "openmm/Platform.h"
#include "openmm/System.h"

namespace AGBNPPlugin {

class CalcAGBNPForceKernel : public OpenMM::KernelImpl {
 public:
  static std::string Name() { return "CalcAGBNPForce"; }
  CalcAGBNPForceKernel(std::string name, const OpenMM::Platform& platform) : OpenMM::KernelImpl(name, platform) {}
  virtual void initialize(const OpenMM::System& system, const AGBNPForce& force) = 0;
  virtual double execute(OpenMM::ContextImpl& context, bool includeForces, bool includeEnergy) = 0;
  virtual void copyParametersToContext(OpenMM::ContextImpl& context, const AGBNPForce& force) = 0;
};

}  // namespace AGBNPPlugin

#include "internal/AGBNPInline.h"
