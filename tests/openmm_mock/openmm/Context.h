#pragma once
#include "openmm/Platform.h"
#include "openmm/System.h"
#include "openmm/internal/ContextImpl.h"
namespace OpenMM {
class Context {
 public:
  Context(const System& system, Platform& platform, void* platformData) : impl(new ContextImpl(*this, system, platform, platformData)) {
    impl->initialize();
  }
  ~Context() { delete impl; }
  const System& getSystem() const { return impl->getSystem(); }
  Platform& getPlatform() { return impl->getPlatform(); }
  ContextImpl& getImpl() { return *impl; }

 private:
  friend class Force;
  ContextImpl* impl;
};
inline ContextImpl& Force::getContextImpl(Context& context) { return context.getImpl(); }
inline ForceImpl& Force::getImplInContext(Context& context) {
  for (ForceImpl* f : context.getImpl().getForceImpls())
    if (&f->getOwner() == this) return *f;
  throw OpenMMException("getImplInContext: This Force is not present in the Context");
}
}  // namespace OpenMM
