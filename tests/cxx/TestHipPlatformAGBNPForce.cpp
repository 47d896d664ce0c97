// The plugin path end to end, shaped like platforms/reference/tests/TestReferenceAGBNPForce.cpp but on the "HIP"
// platform: System + AGBNPForce -> Context (ForceImpl -> Platform::createKernel("CalcAGBNPForce") -> initialize) ->
// calcForcesAndEnergy -> forces from the context's fixed-point buffer, energy from its energy buffer.  The context holds
// its atoms in a SHUFFLED order with padding, like a real OpenMM GPU context.  Reads the reference test's structure
// format on stdin; usage:  TestHipPlatformAGBNPForce <version> <double|mixed|single>
// Prints "Energy: ..." lines in the reference test's format (compared with v0.reference / v1.reference by the caller)
// and checks the moved-atom probe of TestReferenceAGBNPForce.cpp:117-127.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>
#include <vector>

#include "AGBNPForce.h"
#include "HipAGBNPKernels.h"
#include "openmm/Context.h"
#include "openmm/System.h"
#include "openmm/Vec3.h"
#include "openmm/hip/HipPlatform.h"

using namespace AGBNPPlugin;
using namespace OpenMM;

struct Double4 { double x, y, z, w; };
struct Float4 { float x, y, z, w; };

static void uploadPositions(HipContext& cu, const std::vector<Vec3>& pos, const std::vector<double>& charge) {
  const int padded = cu.getPaddedNumAtoms();
  const std::vector<int>& index = cu.getAtomIndex();
  if (cu.getUseDoublePrecision()) {
    std::vector<Double4> posq(padded, Double4{0, 0, 0, 0});
    for (int s = 0; s < cu.getNumAtoms(); s++) posq[s] = Double4{pos[index[s]][0], pos[index[s]][1], pos[index[s]][2], charge[index[s]]};
    cu.getPosq().upload(posq);
    return;
  }
  std::vector<Float4> posq(padded, Float4{0, 0, 0, 0}), corr(padded, Float4{0, 0, 0, 0});
  for (int s = 0; s < cu.getNumAtoms(); s++) {
    const Vec3& p = pos[index[s]];
    posq[s] = Float4{(float)p[0], (float)p[1], (float)p[2], (float)charge[index[s]]};
    corr[s] = Float4{(float)(p[0] - (double)posq[s].x), (float)(p[1] - (double)posq[s].y), (float)(p[2] - (double)posq[s].z), 0.f};
  }
  cu.getPosq().upload(posq);
  if (cu.getUseMixedPrecision()) cu.getPosqCorrection().upload(corr);
}

// energy and forces (particle order) of one evaluation, read the way OpenMM reads them from a GPU context
static double evaluate(Context& context, HipContext& cu, std::vector<Vec3>& forces) {
  std::vector<long long> zeros(3 * (size_t)cu.getPaddedNumAtoms(), 0);
  cu.getLongForceBuffer().upload(zeros);
  double energy = 0.0;
  if (cu.getUseDoublePrecision() || cu.getUseMixedPrecision()) {
    std::vector<double> e(cu.getEnergyBuffer().getSize(), 0.0);
    cu.getEnergyBuffer().upload(e);
    energy += context.getImpl().calcForcesAndEnergy(true, true);
    (void)hipStreamSynchronize(cu.getCurrentStream());
    cu.getEnergyBuffer().download(e);
    for (double v : e) energy += v;
  } else {
    std::vector<float> e(cu.getEnergyBuffer().getSize(), 0.f);
    cu.getEnergyBuffer().upload(e);
    energy += context.getImpl().calcForcesAndEnergy(true, true);
    (void)hipStreamSynchronize(cu.getCurrentStream());
    cu.getEnergyBuffer().download(e);
    for (float v : e) energy += v;
  }
  std::vector<long long> fixed;
  cu.getLongForceBuffer().download(fixed);
  const int padded = cu.getPaddedNumAtoms();
  const double scale = 1.0 / (double)0x100000000LL;
  forces.assign(cu.getNumAtoms(), Vec3());
  for (int s = 0; s < cu.getNumAtoms(); s++)
    forces[cu.getAtomIndex()[s]] = Vec3(scale * fixed[s], scale * fixed[s + padded], scale * fixed[s + 2 * padded]);
  return energy;
}

int main(int argc, char** argv) {
  try {
    const int version = argc > 1 ? atoi(argv[1]) : 1;
    const std::string precision = argc > 2 ? argv[2] : "double";
    int numParticles = 0;
    std::cin >> numParticles;
    System system;
    AGBNPForce* force = new AGBNPForce();
    force->setNonbondedMethod(AGBNPForce::NoCutoff);
    force->setCutoffDistance(1.0);
    force->setVersion(version);
    system.addForce(force);
    const double ang2nm = 0.1, kcalmol2kjmol = 4.184;
    const double sigmaw = 3.15365 * ang2nm, epsilonw = 0.155 * kcalmol2kjmol, rho = 0.033428 / pow(ang2nm, 3);
    const double epsilon_LJ = 0.155 * kcalmol2kjmol;
    std::vector<Vec3> positions;
    std::vector<double> charges;
    for (int i = 0; i < numParticles; i++) {
      double id, x, y, z, radius, charge, gamma;
      int ih;
      std::cin >> id >> x >> y >> z >> radius >> charge >> gamma >> ih;
      system.addParticle(1.0);
      positions.push_back(Vec3(x * ang2nm, y * ang2nm, z * ang2nm));
      charges.push_back(charge);
      radius *= ang2nm;
      gamma *= kcalmol2kjmol / (ang2nm * ang2nm);
      const double sij = sqrt(sigmaw * 2. * radius), eij = sqrt(epsilonw * epsilon_LJ);
      force->addParticle(radius, gamma, -16.0 * M_PI * rho * eij * pow(sij, 6) / 3.0, charge, ih > 0);
    }

    // the platform side: a "HIP" platform with one device context whose atoms are shuffled
    HipPlatform* platform = new HipPlatform();
    Platform::registerPlatform(platform);
    registerAGBNPHipKernelFactories();
    if (!platform->supportsKernels({CalcAGBNPForceKernel::Name()})) {
      std::cout << "FAIL: the HIP platform has no CalcAGBNPForce factory" << std::endl;
      return 1;
    }
    HipPlatform::PlatformData data;
    data.contexts.push_back(new HipContext(numParticles, 0, precision == "double", precision == "mixed"));
    HipContext& cu = *data.contexts[0];
    std::vector<int> order(cu.getPaddedNumAtoms());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
    std::mt19937 rng(20261004);
    std::shuffle(order.begin(), order.begin() + numParticles, rng);
    cu.setAtomIndex(order);

    Context context(system, *platform, &data);  // ForceImpl::initialize -> createKernel -> HipCalcAGBNPForceKernel::initialize
    std::vector<Vec3> forces, scratch;
    uploadPositions(cu, positions, charges);
    const double energy1 = evaluate(context, cu, forces);
    std::cout << "Energy: " << energy1 << std::endl;

    const int pmove = 121 < numParticles ? 121 : 0, direction = 1;
    const double offset = 2.e-3;
    positions[pmove][direction] += offset;
    uploadPositions(cu, positions, charges);
    const double energy2 = evaluate(context, cu, scratch);
    const double de = -forces[pmove][direction] * offset;
    std::cout << "Energy: " << energy2 << std::endl;
    std::cout << "Energy Change: " << energy2 - energy1 << std::endl;
    std::cout << "Energy Change from Gradient: " << de << std::endl;
    double fsum[3] = {0, 0, 0};
    for (const Vec3& f : forces)
      for (int d = 0; d < 3; d++) fsum[d] += f[d];
    std::cout << "Net force: " << fsum[0] << " " << fsum[1] << " " << fsum[2] << std::endl;

    // updateParametersInContext through the API class (copyParametersToContext): scale the charges, energy must change
    for (int i = 0; i < numParticles; i++) {
      double r, g, a, q;
      bool h;
      force->getParticleParameters(i, r, g, a, q, h);
      force->setParticleParameters(i, r, g, a, 0.5 * q, h);
    }
    force->updateParametersInContext(context);
    const double energy3 = evaluate(context, cu, scratch);
    std::cout << "Energy after halving the charges: " << energy3 << std::endl;
    const bool gradient_ok = fabs((energy2 - energy1) - de) < 0.05 * fabs(de) + 1e-3;
    const bool net_ok = fabs(fsum[0]) + fabs(fsum[1]) + fabs(fsum[2]) < 1e-3;
    const bool update_ok = version == 0 ? fabs(energy3 - energy2) < 1e-6 : fabs(energy3 - energy2) > 1.0;  // GVolSA has no charges
    std::cout << ((gradient_ok && net_ok && update_ok) ? "PASS" : "FAIL") << std::endl;
    return (gradient_ok && net_ok && update_ok) ? 0 : 1;
  } catch (const std::exception& e) {
    std::cout << "exception: " << e.what() << std::endl;
    return 2;
  }
}
