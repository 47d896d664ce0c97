// C++ host-side mirror of the reference's API class and kernel interface, on top of the C ABI
// (include/agbnp_hip.h).  Header-only; link against openmm_agbnp_plugin_amd/libagbnp_hip.so.
//
//   AGBNPPlugin::AGBNPForce            <-> openmmapi/include/AGBNPForce.h:39-155, openmmapi/src/AGBNPForce.cpp:15-78
//   AGBNPPlugin::HipCalcAGBNPForceKernel <-> CalcAGBNPForceKernel, openmmapi/include/AGBNPKernels.h:19-47
//
// Same method names, argument order and meaning, defaults and error behaviour (exceptions with the
// reference's messages).  OpenMM types are replaced by plain containers because OpenMM is not part of this
// path: positions / forces are std::vector<double> of 3N values (x0,y0,z0,x1,...), nm and kJ/mol/nm.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/agbnp_hip.h"

namespace AGBNPPlugin {

// stands in for OpenMM::OpenMMException
class OpenMMException : public std::runtime_error {
 public:
  explicit OpenMMException(const std::string& msg) : std::runtime_error(msg) {}
};

class AGBNPForce {
 public:
  enum NonbondedMethod { NoCutoff = 0, CutoffNonPeriodic = 1, CutoffPeriodic = 2 };

  // defaults of the reference constructor (AGBNPForce.cpp:15)
  AGBNPForce() : nonbondedMethod(NoCutoff), cutoffDistance(1.0), version(1), solvent_radius(1.0 * (0.1f)) {}

  int addParticle(double radius, double gamma, double vdw_alpha, double charge, bool ishydrogen) {
    particles.push_back(ParticleInfo{radius, gamma, vdw_alpha, charge, ishydrogen});
    return (int)particles.size() - 1;
  }
  void setParticleParameters(int index, double radius, double gamma, double vdw_alpha, double charge, bool ishydrogen) {
    checkIndex(index);
    particles[index] = ParticleInfo{radius, gamma, vdw_alpha, charge, ishydrogen};
  }
  void getParticleParameters(int index, double& radius, double& gamma, double& vdw_alpha, double& charge, bool& ishydrogen) const {
    checkIndex(index);
    const ParticleInfo& p = particles[index];
    radius = p.radius;
    gamma = p.gamma;
    vdw_alpha = p.vdw_alpha;
    charge = p.charge;
    ishydrogen = p.ishydrogen;
  }
  int getNumParticles() const { return (int)particles.size(); }
  NonbondedMethod getNonbondedMethod() const { return nonbondedMethod; }
  void setNonbondedMethod(NonbondedMethod method) { nonbondedMethod = method; }
  double getCutoffDistance() const { return cutoffDistance; }
  void setCutoffDistance(double distance) { cutoffDistance = distance; }
  double getSolventRadius() const { return solvent_radius; }
  void setVersion(int agbnp_version) {  // 0 = GVolSA, 1 = AGBNP1, 2 = AGBNP2 (AGBNPForce.cpp:52-59)
    if (agbnp_version >= 0 && agbnp_version <= 2)
      version = (unsigned)agbnp_version;
    else
      throw OpenMMException("AGBNPForce::setVersion(): illegal version number");
  }
  unsigned int getVersion() const { return version; }

 private:
  struct ParticleInfo {
    double radius, gamma, vdw_alpha, charge;
    bool ishydrogen;
  };
  void checkIndex(int index) const {
    if (index < 0 || index >= (int)particles.size()) throw OpenMMException("Assertion failure: Index out of range");
  }
  std::vector<ParticleInfo> particles;
  NonbondedMethod nonbondedMethod;
  double cutoffDistance;
  unsigned int version;
  double solvent_radius;
};

// The "HIP platform" implementation of the plugin's kernel interface.
class HipCalcAGBNPForceKernel {
 public:
  static std::string Name() { return "CalcAGBNPForce"; }
  explicit HipCalcAGBNPForceKernel(int device = 0) : ctx(nullptr), device(device), numParticles(0) {}
  ~HipCalcAGBNPForceKernel() { agbnp_hip_destroy(ctx); }
  HipCalcAGBNPForceKernel(const HipCalcAGBNPForceKernel&) = delete;
  HipCalcAGBNPForceKernel& operator=(const HipCalcAGBNPForceKernel&) = delete;

  void initialize(const AGBNPForce& force) {
    agbnp_hip_destroy(ctx);
    ctx = nullptr;
    std::vector<double> r, g, a, q;
    std::vector<int> h;
    gather(force, r, g, a, q, h);
    numParticles = (int)r.size();
    if (agbnp_hip_create(&ctx, numParticles, r.data(), g.data(), a.data(), q.data(), h.data(), (int)force.getVersion(),
                         (int)force.getNonbondedMethod(), force.getCutoffDistance(), device) != AGBNP_HIP_OK)
      throw OpenMMException(agbnp_hip_last_error(nullptr));
  }

  // CPU-platform data convention of the reference (ReferenceAGBNPKernels.cpp:27-35,197,794): forces are
  // accumulated into `forces`, the energy is returned; the two flags are accepted and ignored.
  double execute(const std::vector<double>& positions, std::vector<double>& forces, bool includeForces = true,
                 bool includeEnergy = true) {
    (void)includeForces;
    (void)includeEnergy;
    if (!ctx) throw OpenMMException("HipCalcAGBNPForceKernel: initialize() has not been called");
    if ((int)positions.size() != 3 * numParticles || (int)forces.size() != 3 * numParticles)
      throw OpenMMException("execute(): positions and forces must hold 3N values");
    double energy = 0.0;
    if (agbnp_hip_execute_host(ctx, positions.data(), forces.data(), &energy) != AGBNP_HIP_OK)
      throw OpenMMException(agbnp_hip_last_error(ctx));
    return energy;
  }

  void copyParametersToContext(const AGBNPForce& force) {
    if (!ctx) throw OpenMMException("HipCalcAGBNPForceKernel: initialize() has not been called");
    std::vector<double> r, g, a, q;
    std::vector<int> h;
    gather(force, r, g, a, q, h);
    if (agbnp_hip_update_parameters(ctx, (int)r.size(), r.data(), g.data(), a.data(), q.data(), h.data()) != AGBNP_HIP_OK)
      throw OpenMMException(agbnp_hip_last_error(ctx));
  }

  agbnp_hip_context* handle() const { return ctx; }

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
  int device;
  int numParticles;
};

}  // namespace AGBNPPlugin
