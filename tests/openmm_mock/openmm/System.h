#pragma once
#include <vector>
#include "openmm/Force.h"
namespace OpenMM {
class System {
 public:
  ~System() {
    for (Force* f : forces) delete f;
  }
  int addParticle(double mass) {
    masses.push_back(mass);
    return (int)masses.size() - 1;
  }
  int getNumParticles() const { return (int)masses.size(); }
  double getParticleMass(int index) const { return masses.at(index); }
  int addForce(Force* force) {  // the System takes ownership, as in OpenMM
    forces.push_back(force);
    return (int)forces.size() - 1;
  }
  int getNumForces() const { return (int)forces.size(); }
  Force& getForce(int index) { return *forces.at(index); }
  const Force& getForce(int index) const { return *forces.at(index); }

 private:
  std::vector<double> masses;
  std::vector<Force*> forces;
};
}  // namespace OpenMM
