// OpenMM "HIP" platform implementation of the AGBNP plugin's kernel, on top of libagbnp_hip.so (include/agbnp_hip.h).
// Counterpart of platforms/opencl/src/OpenCLAGBNPKernels.h:20-60 (class OpenCLCalcAGBNPForceKernel) for versions 0 / 1.
#ifndef HIP_AGBNP_KERNELS_H_
#define HIP_AGBNP_KERNELS_H_

#include <string>
#include <vector>

#include "AGBNPKernels.h"
#include "agbnp_hip.h"
#include "openmm/hip/HipContext.h"

namespace AGBNPPlugin {

class HipCalcAGBNPForceKernel : public CalcAGBNPForceKernel {
 public:
  HipCalcAGBNPForceKernel(std::string name, const OpenMM::Platform& platform, OpenMM::HipContext& cu);
  ~HipCalcAGBNPForceKernel() override;
  void initialize(const OpenMM::System& system, const AGBNPForce& force) override;
  double execute(OpenMM::ContextImpl& context, bool includeForces, bool includeEnergy) override;
  void copyParametersToContext(OpenMM::ContextImpl& context, const AGBNPForce& force) override;

  // How often execute() synchronises to read the engine's overflow log.  1 (default) = every evaluation, the protocol of
  // the reference's OpenCL platform (a blocking PanicButton read per step, OpenCLAGBNPKernels.cpp:3599-3634): an
  // evaluation whose trees outgrew their store is repeated before execute() returns.  k > 1 = every k-th evaluation:
  // no host synchronisation in between; a withheld evaluation found then cannot be repeated in place (the integrator has
  // moved on), so execute() throws.  Environment variable AGBNP_HIP_CHECK_INTERVAL overrides the default.
  void setCheckInterval(int evaluations);
  // Poll mode (AGBNP_HIP_CHECK_MODE=poll): execute() never synchronises in the steady state.  After every enqueue it looks
  // at the engine's pinned status words (agbnp_hip_poll: no device call) and only calls the blocking agbnp_hip_finish()
  // when they report a withheld evaluation (or every 1024 evaluations, to keep the log bounded).  What the poll sees is at
  // least one evaluation old, so a withheld evaluation is found AFTER the integrator has used the step's forces without
  // the AGBNP term: the engine adapts at once (capacity / packing), the step is counted in getLateWithheld() and reported
  // on stderr once.  The first eight evaluations of a context, and the eight after every adaptation, are checked the
  // strict way (the capacity negotiation of a new system happens there).  The default stays the strict protocol above
  // (exact, one synchronisation per step).
  void setPollMode(bool on);
  int getLateWithheld() const { return lateWithheld; }
  agbnp_hip_context* getEngine() { return engine; }

 private:
  void enqueue();
  OpenMM::HipContext& cu;
  agbnp_hip_context* engine;
  int checkInterval, sinceCheck;
  bool pollMode = false;
  int lateWithheld = 0;
  int strictLeft = 0;  // poll mode: evaluations that are still checked the strict way (a fresh context, or one that has just adapted)
};

}  // namespace AGBNPPlugin

// plugin entry points (the shape of platforms/reference/src/ReferenceAGBNPKernelFactory.cpp:14-36)
extern "C" void registerPlatforms();
extern "C" void registerKernelFactories();
extern "C" void registerAGBNPHipKernelFactories();

#endif
