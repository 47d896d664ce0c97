// C++ test of the HIP path through the host-side mirror (cpp/AGBNPForce.h), written to read like the
// reference's own test program platforms/reference/tests/TestReferenceAGBNPForce.cpp: it reads a structure
// from stdin (`id x y z radius charge gamma ishydrogen`, Angstrom / e / kcal/mol/A^2), applies the test's
// parameterisation (:47-70), evaluates once and prints "Energy: ...".  Unlike the reference's test it also
// ASSERTS: the moved-atom probe of :117-127 and, for the bundled 264-atom fixture, the known answers of
// v0.reference / v1.reference.  Usage:  TestHipAGBNPForce <version> [expect_energy expect_moved] < gaussvol.dat
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "../../cpp/AGBNPForce.h"

using namespace AGBNPPlugin;

static double sig6(double x) {
  char buf[64];
  snprintf(buf, sizeof buf, "%.6g", x);
  return atof(buf);
}

int main(int argc, char** argv) {
  try {
    const int version = argc > 1 ? atoi(argv[1]) : 1;
    AGBNPForce force;
    force.setVersion(version);
    int numParticles = 0;
    std::cin >> numParticles;
    const double ang2nm = 0.1, kcalmol2kjmol = 4.184;
    const double sigmaw = 3.15365 * ang2nm, epsilonw = 0.155 * kcalmol2kjmol, rho = 0.033428 / pow(ang2nm, 3);
    const double epsilon_LJ = 0.155 * kcalmol2kjmol;
    std::vector<double> positions;
    for (int i = 0; i < numParticles; i++) {
      double id, x, y, z, radius, charge, gamma;
      int ih;
      std::cin >> id >> x >> y >> z >> radius >> charge >> gamma >> ih;
      positions.push_back(x * ang2nm);
      positions.push_back(y * ang2nm);
      positions.push_back(z * ang2nm);
      radius *= ang2nm;
      gamma *= kcalmol2kjmol / (ang2nm * ang2nm);
      const double sij = sqrt(sigmaw * 2. * radius), eij = sqrt(epsilonw * epsilon_LJ);
      const double alpha = -16.0 * M_PI * rho * eij * pow(sij, 6) / 3.0;
      force.addParticle(radius, gamma, alpha, charge, ih > 0);
    }
    HipCalcAGBNPForceKernel kernel;
    kernel.initialize(force);
    std::vector<double> forces(3 * numParticles, 0.0);
    const double energy1 = kernel.execute(positions, forces, true, true);
    std::cout << "Energy: " << energy1 << std::endl;

    // the probe of TestReferenceAGBNPForce.cpp:117-127
    const int pmove = 121 < numParticles ? 121 : 0, direction = 1;
    const double offset = 2.e-3;
    positions[3 * pmove + direction] += offset;
    std::vector<double> scratch(3 * numParticles, 0.0);
    const double energy2 = kernel.execute(positions, scratch, true, true);
    const double de = -forces[3 * pmove + direction] * offset;
    std::cout << "Energy: " << energy2 << std::endl;
    std::cout << "Energy Change: " << energy2 - energy1 << std::endl;
    std::cout << "Energy Change from Gradient: " << de << std::endl;
    int bad = 0;
    if (std::fabs((energy2 - energy1) - de) > 0.05 * std::fabs(de) + 1e-6) {
      std::cout << "FAIL: finite-difference probe disagrees with the force" << std::endl;
      bad = 1;
    }
    if (argc > 3) {
      if (sig6(energy1) != atof(argv[2]) || sig6(energy2) != atof(argv[3])) {
        std::cout << "FAIL: known answers " << argv[2] << " / " << argv[3] << " not reproduced" << std::endl;
        bad = 1;
      }
    }
    // error behaviour of the API class
    try {
      force.setVersion(7);
      std::cout << "FAIL: illegal version accepted" << std::endl;
      bad = 1;
    } catch (const OpenMMException&) {
    }
    if (!bad) std::cout << "PASS" << std::endl;
    return bad;
  } catch (const std::exception& e) {
    std::cout << "exception: " << e.what() << std::endl;
    return 1;
  }
}
