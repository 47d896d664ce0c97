// Inline bodies of the restated API classes (need AGBNPForce, AGBNPForceImpl and CalcAGBNPForceKernel complete).
#pragma once
namespace AGBNPPlugin {
inline void AGBNPForceImpl::initialize(OpenMM::ContextImpl& context) {
  kernel = context.getPlatform().createKernel(CalcAGBNPForceKernel::Name(), context);
  kernel.getAs<CalcAGBNPForceKernel>().initialize(context.getSystem(), owner);
}
inline double AGBNPForceImpl::calcForcesAndEnergy(OpenMM::ContextImpl& context, bool includeForces, bool includeEnergy, int groups) {
  if ((groups & (1 << owner.getForceGroup())) == 0) return 0.0;
  return kernel.getAs<CalcAGBNPForceKernel>().execute(context, includeForces, includeEnergy);
}
inline std::vector<std::string> AGBNPForceImpl::getKernelNames() { return {CalcAGBNPForceKernel::Name()}; }
inline void AGBNPForceImpl::updateParametersInContext(OpenMM::ContextImpl& context) {
  kernel.getAs<CalcAGBNPForceKernel>().copyParametersToContext(context, owner);
}
inline OpenMM::ForceImpl* AGBNPForce::createImpl() const { return new AGBNPForceImpl(*this); }
inline void AGBNPForce::updateParametersInContext(OpenMM::Context& context) {
  dynamic_cast<AGBNPForceImpl&>(getImplInContext(context)).updateParametersInContext(getContextImpl(context));
}
}  // namespace AGBNPPlugin
