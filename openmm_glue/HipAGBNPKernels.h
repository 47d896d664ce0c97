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

  // How execute() learns whether the evaluation it has just enqueued was withheld (a tree outgrew its store).
  // Default: the VERDICT protocol.  execute() waits -- on the host, for a word in pinned memory, not for the stream -- until
  // the device has judged THIS evaluation (agbnp_hip_wait_verdict: the verdict is final when the tree stage has ended, three
  // quarters into the evaluation; the forces follow on the stream, gated on the device by the same words).  A withheld
  // evaluation is repeated before execute() returns, exactly as under the reference's protocol (a blocking PanicButton read
  // per step, OpenCLAGBNPKernels.cpp:3599-3634), but the stream is never drained in the steady state.
  // AGBNP_HIP_CHECK_MODE=finish restores the reference's own protocol (agbnp_hip_finish after every evaluation: stream
  // synchronisation + two device reads per step).
  // Check interval k > 1 (setCheckInterval / AGBNP_HIP_CHECK_INTERVAL; finish mode): a blocking check every k-th evaluation
  // only; a withheld evaluation found then cannot be repeated in place (the integrator has moved on), so execute() throws.
  void setCheckInterval(int evaluations);
  void setVerdictMode(bool on);
  // Poll mode (AGBNP_HIP_CHECK_MODE=poll): execute() never synchronises in the steady state.  After every enqueue it looks
  // at the engine's pinned status words (agbnp_hip_poll: no device call) and only calls the blocking agbnp_hip_finish()
  // when they report a withheld evaluation (or every 1024 evaluations, to keep the log bounded).  What the poll sees is at
  // least one evaluation old, so a withheld evaluation is found AFTER the integrator has used the step's forces without
  // the AGBNP term: the engine adapts at once (capacity / packing), the step is counted in getLateWithheld() and reported
  // on stderr once.  The first eight evaluations of a context, and the eight after every adaptation, are checked the
  // strict way (the capacity negotiation of a new system happens there).  Not the default: the verdict protocol above is
  // exact and costs about as little.
  void setPollMode(bool on);
  int getLateWithheld() const { return lateWithheld; }
  agbnp_hip_context* getEngine() { return engine; }

 private:
  void enqueue();
  OpenMM::HipContext& cu;
  agbnp_hip_context* engine;
  int checkInterval, sinceCheck;
  bool pollMode = false;
  std::vector<int> lastOrder;  // the context's atom order at the last enqueue (host copy)
  bool verdictMode = true;
  int sinceFinish = 0;  // verdict mode: evaluations since the last agbnp_hip_finish (one every 1024 keeps the device's log bounded)
  int lateWithheld = 0;
  int strictLeft = 0;  // poll mode: evaluations that are still checked the strict way (a fresh context, or one that has just adapted)
};

}  // namespace AGBNPPlugin

// plugin entry points (the shape of platforms/reference/src/ReferenceAGBNPKernelFactory.cpp:14-36)
extern "C" void registerPlatforms();
extern "C" void registerKernelFactories();
extern "C" void registerAGBNPHipKernelFactories();

#endif
