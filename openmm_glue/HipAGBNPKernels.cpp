// OpenMM-side glue for the gfx950 engine: a CalcAGBNPForceKernel implementation and its kernel factory,
// written against the reference's own interfaces
//   openmmapi/include/AGBNPKernels.h:19-47           (CalcAGBNPForceKernel)
//   platforms/reference/src/ReferenceAGBNPKernelFactory.cpp:14-36  (plugin entry points, factory)
// It needs the OpenMM headers and the reference's openmmapi headers, neither of which exists in the
// build image, so it is NOT compiled by build(); a maintainer adds it to the plugin's CMake as a third
// platform library next to platforms/reference and platforms/opencl.
//
// Data convention used here: the CPU-platform one (positions/forces as std::vector<Vec3>, energy
// returned), so the kernel can be registered on the Reference or CPU platform of any OpenMM build
// and still run its arithmetic on the MI355X.  For a GPU platform whose context owns FP64 device
// buffers, call agbnp_hip_execute_device with those pointers instead.
#include <string>
#include <vector>

#include "AGBNPKernels.h"
#include "agbnp_hip.h"
#include "openmm/OpenMMException.h"
#include "openmm/internal/ContextImpl.h"
#include "openmm/reference/ReferencePlatform.h"

using namespace AGBNPPlugin;
using namespace OpenMM;

namespace {

class HipCalcAGBNPForceKernel : public CalcAGBNPForceKernel {
 public:
  HipCalcAGBNPForceKernel(std::string name, const Platform& platform) : CalcAGBNPForceKernel(name, platform), ctx(nullptr) {}
  ~HipCalcAGBNPForceKernel() { agbnp_hip_destroy(ctx); }

  void initialize(const System& system, const AGBNPForce& force) override {
    std::vector<double> r, g, a, q;
    std::vector<int> h;
    gather(force, r, g, a, q, h);
    int device = 0;  // one context <-> one device, as the reference's OpenCL platform (OpenCLAGBNPKernels.cpp:411-413)
    if (agbnp_hip_create(&ctx, (int)r.size(), r.data(), g.data(), a.data(), q.data(), h.data(), (int)force.getVersion(),
                         (int)force.getNonbondedMethod(), force.getCutoffDistance(), device) != AGBNP_HIP_OK)
      throw OpenMMException(agbnp_hip_last_error(nullptr));
  }

  double execute(ContextImpl& context, bool includeForces, bool includeEnergy) override {
    ReferencePlatform::PlatformData* data = reinterpret_cast<ReferencePlatform::PlatformData*>(context.getPlatformData());
    std::vector<Vec3>& pos = *((std::vector<Vec3>*)data->positions);
    std::vector<Vec3>& frc = *((std::vector<Vec3>*)data->forces);
    double energy = 0.0;  // Vec3 is three contiguous doubles: the vectors are [3N] arrays
    if (agbnp_hip_execute_host(ctx, &pos[0][0], &frc[0][0], &energy) != AGBNP_HIP_OK) throw OpenMMException(agbnp_hip_last_error(ctx));
    return energy;
  }

  void copyParametersToContext(ContextImpl& context, const AGBNPForce& force) override {
    std::vector<double> r, g, a, q;
    std::vector<int> h;
    gather(force, r, g, a, q, h);
    if (agbnp_hip_update_parameters(ctx, (int)r.size(), r.data(), g.data(), a.data(), q.data(), h.data()) != AGBNP_HIP_OK)
      throw OpenMMException(agbnp_hip_last_error(ctx));
  }

 private:
  static void gather(const AGBNPForce& force, std::vector<double>& r, std::vector<double>& g, std::vector<double>& a,
                     std::vector<double>& q, std::vector<int>& h) {
    const int n = force.getNumParticles();
    r.resize(n), g.resize(n), a.resize(n), q.resize(n), h.resize(n);
    for (int i = 0; i < n; i++) {
      bool ish;
      force.getParticleParameters(i, r[i], g[i], a[i], q[i], ish);
      h[i] = ish ? 1 : 0;
    }
  }
  agbnp_hip_context* ctx;
};

class HipAGBNPKernelFactory : public KernelFactory {
 public:
  KernelImpl* createKernelImpl(std::string name, const Platform& platform, ContextImpl& context) const override {
    if (name == CalcAGBNPForceKernel::Name()) return new HipCalcAGBNPForceKernel(name, platform);
    throw OpenMMException((std::string("Tried to create kernel with illegal kernel name '") + name + "'").c_str());
  }
};

}  // namespace

extern "C" void registerPlatforms() {}

extern "C" void registerKernelFactories() {
  for (int i = 0; i < Platform::getNumPlatforms(); i++) {
    Platform& platform = Platform::getPlatform(i);
    if (dynamic_cast<ReferencePlatform*>(&platform) != NULL)
      platform.registerKernelFactory(CalcAGBNPForceKernel::Name(), new HipAGBNPKernelFactory());
  }
}

extern "C" void registerAGBNPHipKernelFactories() { registerKernelFactories(); }
