#pragma once
#include <map>
#include <string>
#include <vector>
namespace OpenMM {
class ContextImpl;
class Force;
class ForceImpl {
 public:
  virtual ~ForceImpl() {}
  virtual void initialize(ContextImpl& context) = 0;
  virtual const Force& getOwner() const = 0;
  virtual void updateContextState(ContextImpl& context, bool& forcesInvalid) = 0;
  virtual double calcForcesAndEnergy(ContextImpl& context, bool includeForces, bool includeEnergy, int groups) = 0;
  virtual std::map<std::string, double> getDefaultParameters() = 0;
  virtual std::vector<std::string> getKernelNames() = 0;
};
}  // namespace OpenMM
