// OpenMM-side glue of the gfx950 engine: the platform plugin a maintainer adds to Gallicchio-Lab/openmm_agbnp_plugin next
// to platforms/reference and platforms/opencl.  Written against
//   openmmapi/include/AGBNPKernels.h:19-47                          (CalcAGBNPForceKernel: what it implements)
//   platforms/reference/src/ReferenceAGBNPKernelFactory.cpp:14-36   (plugin entry points, factory)
//   platforms/opencl/src/OpenCLAGBNPKernels.cpp:541-556             (GPU data conventions: device posq in, forces and
//                                                                    energy added on the device, execute returns 0.0)
// and OpenMM's HIP platform (HipPlatform::PlatformData::contexts, HipContext, HipArray).  It registers on the platform
// named "HIP" only -- not on the Reference platform, whose "CalcAGBNPForce" factory belongs to AGBNPPluginReference.
//
// Compiled and run by tests/test_openmm_glue.py against tests/openmm_mock (OpenMM is not in the build image) and, in the
// build container, also against the reference's own openmmapi/include headers taken in place.
#include "HipAGBNPKernels.h"

#include <cstdio>
#include <cstdlib>
#include <string>

#include "openmm/OpenMMException.h"
#include "openmm/hip/HipPlatform.h"
#include "openmm/internal/ContextImpl.h"

using namespace AGBNPPlugin;
using namespace OpenMM;

namespace {

void gatherParameters(const AGBNPForce& force, std::vector<double>& r, std::vector<double>& g, std::vector<double>& a,
                      std::vector<double>& q, std::vector<int>& h) {
  const int n = force.getNumParticles();
  r.resize(n), g.resize(n), a.resize(n), q.resize(n), h.resize(n);
  for (int i = 0; i < n; i++) {
    bool ish;
    force.getParticleParameters(i, r[i], g[i], a[i], q[i], ish);
    h[i] = ish ? 1 : 0;
  }
}

class HipAGBNPKernelFactory : public KernelFactory {
 public:
  KernelImpl* createKernelImpl(std::string name, const Platform& platform, ContextImpl& context) const override {
    HipPlatform::PlatformData& data = *static_cast<HipPlatform::PlatformData*>(context.getPlatformData());
    if (data.contexts.size() > 1)  // as the reference's OpenCL platform (OpenCLAGBNPKernels.cpp:411-413)
      throw OpenMMException("AGBNPForce does not support using multiple HIP devices");
    if (name == CalcAGBNPForceKernel::Name()) return new HipCalcAGBNPForceKernel(name, platform, *data.contexts[0]);
    throw OpenMMException((std::string("Tried to create kernel with illegal kernel name '") + name + "'").c_str());
  }
};

}  // namespace

HipCalcAGBNPForceKernel::HipCalcAGBNPForceKernel(std::string name, const Platform& platform, HipContext& cu)
    : CalcAGBNPForceKernel(name, platform), cu(cu), engine(nullptr), checkInterval(1), sinceCheck(0) {
  if (const char* env = getenv("AGBNP_HIP_CHECK_MODE")) {
    setPollMode(std::string(env) == "poll");
    setVerdictMode(std::string(env) == "verdict");  // ("finish": the reference's own protocol)
  }
  if (const char* env = getenv("AGBNP_HIP_CHECK_INTERVAL")) setCheckInterval(atoi(env));
}

void HipCalcAGBNPForceKernel::setVerdictMode(bool on) {
  verdictMode = on;
  if (on) pollMode = false;
}

void HipCalcAGBNPForceKernel::setPollMode(bool on) {
  pollMode = on;
  if (on) verdictMode = false;
  if (on && checkInterval == 1) checkInterval = 1024;  // the full check only bounds the log in this mode
}

HipCalcAGBNPForceKernel::~HipCalcAGBNPForceKernel() { agbnp_hip_destroy(engine); }

void HipCalcAGBNPForceKernel::setCheckInterval(int evaluations) {
  checkInterval = evaluations < 1 ? 1 : evaluations;
  if (checkInterval > 1) verdictMode = false;  // (an interval only means something for the blocking check)
}

void HipCalcAGBNPForceKernel::initialize(const System& system, const AGBNPForce& force) {
  if (force.getNumParticles() != system.getNumParticles())
    throw OpenMMException("AGBNPForce must have exactly as many particles as the System it belongs to.");
  if (force.getVersion() == 2)
    throw OpenMMException("HipCalcAGBNPForceKernel: AGBNP version 2 is not implemented on the HIP platform (versions 0 and 1 are)");
  std::vector<double> r, g, a, q;
  std::vector<int> h;
  gatherParameters(force, r, g, a, q, h);
  agbnp_hip_destroy(engine);
  engine = nullptr;
  if (agbnp_hip_create(&engine, (int)r.size(), r.data(), g.data(), a.data(), q.data(), h.data(), (int)force.getVersion(),
                       (int)force.getNonbondedMethod(), force.getCutoffDistance(), cu.getDeviceIndex()) != AGBNP_HIP_OK)
    throw OpenMMException(agbnp_hip_last_error(nullptr));
  strictLeft = 8;
  sinceCheck = 0;
  sinceFinish = 0;
}

void HipCalcAGBNPForceKernel::enqueue() {
  const bool dbl = cu.getUseDoublePrecision(), mixed = cu.getUseMixedPrecision();
  // OpenMM reorders its atoms now and then (ComputeContext::reorderAtoms: same arrays, new contents).  The engine would find
  // out by itself -- and repeat one evaluation; the host copy of the order is at hand, so it is told beforehand.
  const std::vector<int>& order = cu.getAtomIndex();
  if (order != lastOrder) {
    if (!lastOrder.empty()) agbnp_hip_atom_order_changed(engine);
    lastOrder = order;
  }
  if (agbnp_hip_execute_openmm(engine, cu.getPosq().getDevicePointer(), dbl ? 1 : 0,
                               mixed ? cu.getPosqCorrection().getDevicePointer() : nullptr,
                               static_cast<const int*>(cu.getAtomIndexArray().getDevicePointer()), cu.getPaddedNumAtoms(),
                               static_cast<long long*>(cu.getLongForceBuffer().getDevicePointer()),
                               cu.getEnergyBuffer().getDevicePointer(), (dbl || mixed) ? 1 : 0, /*energy slot*/ 0,
                               cu.getCurrentStream()) != AGBNP_HIP_OK)
    throw OpenMMException(agbnp_hip_last_error(engine));
}

double HipCalcAGBNPForceKernel::execute(ContextImpl& context, bool includeForces, bool includeEnergy) {
  if (!engine) throw OpenMMException("HipCalcAGBNPForceKernel: initialize() has not been called");
  // both are always computed, as in the reference (ReferenceAGBNPKernels.cpp:139-149 ignores the two flags)
  enqueue();
  if (verdictMode) {
    // the host waits for the device's word on this evaluation, not for the stream; the blocking path below is only taken
    // for a withheld evaluation (repeat it, as the reference does), without pinned memory, and once in 1024 evaluations
    int done = 0, bad = 0;
    const int rc = agbnp_hip_wait_verdict(engine, 0, 10.0, &done, &bad);
    if (rc == AGBNP_HIP_OK && bad == 0 && ++sinceFinish < 1024) return 0.0;
    sinceFinish = 0;
    sinceCheck = 0;  // (the blocking check below judges exactly this evaluation)
  }
  ++sinceCheck;
  bool check = sinceCheck >= checkInterval || verdictMode;
  if (pollMode && strictLeft > 0) {
    strictLeft--;
    check = true;
  }
  if (!check && pollMode) {  // a look at the pinned status words: no device call, nothing is waited for
    int done = 0, bad = 0;
    check = agbnp_hip_poll(engine, &done, &bad) != AGBNP_HIP_OK || bad > 0;
  }
  if (check) {
    const bool only_this_one = sinceCheck == 1;
    sinceCheck = 0;
    for (int attempt = 0;; attempt++) {
      int withheld = 0;
      if (agbnp_hip_finish(engine, cu.getCurrentStream(), &withheld) != AGBNP_HIP_OK) throw OpenMMException(agbnp_hip_last_error(engine));
      if (withheld == 0) break;
      strictLeft = 8;
      // a withheld evaluation added nothing to the context's buffers.  With a check after every evaluation it is this
      // one: run it again on the capacity the engine has just switched to (forces invalidated and recomputed in the
      // reference's words, OpenCLAGBNPKernels.cpp:3613-3634).
      if (pollMode && !only_this_one) {  // found late: the engine has adapted; the steps in question ran without the AGBNP term
        if (lateWithheld == 0)
          fprintf(stderr, "AGBNPForce (HIP): %d evaluation(s) outgrew the overlap-tree capacity and were withheld before the poll saw it; "
                          "capacity raised, run continues (AGBNP_HIP_CHECK_MODE unset = strict per-step check)\n", withheld);
        lateWithheld += withheld;
        break;
      }
      if (!only_this_one)
        throw OpenMMException("AGBNPForce (HIP): an overlap tree outgrew its capacity in an earlier step of this check interval; "
                              "its forces were not applied.  Restart from the last checkpoint (the capacity has been raised) "
                              "or run with AGBNP_HIP_CHECK_INTERVAL=1");
      if (attempt >= 8) throw OpenMMException("AGBNPForce (HIP): capacity negotiation did not converge");
      enqueue();
    }
  }
  return 0.0;  // the energy went into the context's energy buffer (OpenCLAGBNPKernels.cpp:555)
}

void HipCalcAGBNPForceKernel::copyParametersToContext(ContextImpl& context, const AGBNPForce& force) {
  if (!engine) throw OpenMMException("HipCalcAGBNPForceKernel: initialize() has not been called");
  std::vector<double> r, g, a, q;
  std::vector<int> h;
  gatherParameters(force, r, g, a, q, h);
  if (agbnp_hip_update_parameters(engine, (int)r.size(), r.data(), g.data(), a.data(), q.data(), h.data()) != AGBNP_HIP_OK)
    throw OpenMMException(agbnp_hip_last_error(engine));
}

extern "C" void registerPlatforms() {}

extern "C" void registerKernelFactories() {
  for (int i = 0; i < Platform::getNumPlatforms(); i++) {
    Platform& platform = Platform::getPlatform(i);
    if (platform.getName() == "HIP") platform.registerKernelFactory(CalcAGBNPForceKernel::Name(), new HipAGBNPKernelFactory());
  }
}

extern "C" void registerAGBNPHipKernelFactories() { registerKernelFactories(); }
