// Restatement of the plugin's API class for builds without /root/reference (header-only): same class name, namespace,
// methods, defaults and error text as openmmapi/include/AGBNPForce.h:39-155 + openmmapi/src/AGBNPForce.cpp:15-78.
#pragma once
#include <vector>
#include "openmm/Context.h"
#include "openmm/Force.h"
#include "openmm/OpenMMException.h"

namespace AGBNPPlugin {

class AGBNPForce : public OpenMM::Force {
 public:
  enum NonbondedMethod { NoCutoff = 0, CutoffNonPeriodic = 1, CutoffPeriodic = 2 };
  AGBNPForce() : nonbondedMethod(NoCutoff), cutoffDistance(1.0), version(1), solvent_radius(1.0 * (0.1f)) {}
  int addParticle(double radius, double gamma, double vdw_alpha, double charge, bool ishydrogen) {
    particles.push_back({radius, gamma, vdw_alpha, charge, ishydrogen});
    return (int)particles.size() - 1;
  }
  void setParticleParameters(int index, double radius, double gamma, double vdw_alpha, double charge, bool ishydrogen) {
    check(index);
    particles[index] = {radius, gamma, vdw_alpha, charge, ishydrogen};
  }
  void getParticleParameters(int index, double& radius, double& gamma, double& vdw_alpha, double& charge, bool& ishydrogen) const {
    check(index);
    const Particle& p = particles[index];
    radius = p.radius, gamma = p.gamma, vdw_alpha = p.vdw_alpha, charge = p.charge, ishydrogen = p.ishydrogen;
  }
  int getNumParticles() const { return (int)particles.size(); }
  NonbondedMethod getNonbondedMethod() const { return nonbondedMethod; }
  void setNonbondedMethod(NonbondedMethod method) { nonbondedMethod = method; }
  double getCutoffDistance() const { return cutoffDistance; }
  void setCutoffDistance(double distance) { cutoffDistance = distance; }
  double getSolventRadius() const { return solvent_radius; }
  void setVersion(int agbnp_version) {
    if (agbnp_version < 0 || agbnp_version > 2) throw OpenMM::OpenMMException("AGBNPForce::setVersion(): illegal version number");
    version = (unsigned)agbnp_version;
  }
  unsigned int getVersion() const { return version; }
  inline void updateParametersInContext(OpenMM::Context& context);

 protected:
  inline OpenMM::ForceImpl* createImpl() const override;

 private:
  struct Particle {
    double radius, gamma, vdw_alpha, charge;
    bool ishydrogen;
  };
  void check(int index) const {
    if (index < 0 || index >= (int)particles.size()) throw OpenMM::OpenMMException("Assertion failure: Index out of range");
  }
  std::vector<Particle> particles;
  NonbondedMethod nonbondedMethod;
  double cutoffDistance;
  unsigned int version;
  double solvent_radius;
};

}  // namespace AGBNPPlugin

#include "internal/AGBNPForceImpl.h"
#include "AGBNPKernels.h"  // (completes the inline members: see internal/AGBNPInline.h)
