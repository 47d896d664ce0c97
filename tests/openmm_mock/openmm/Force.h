#pragma once
namespace OpenMM {
class ForceImpl;
class Context;
class ContextImpl;
class Force {
 public:
  Force() : forceGroup(0) {}
  virtual ~Force() {}
  int getForceGroup() const { return forceGroup; }
  void setForceGroup(int group) { forceGroup = group; }
  virtual bool usesPeriodicBoundaryConditions() const { return false; }

 protected:
  friend class ContextImpl;
  virtual ForceImpl* createImpl() const = 0;
  ForceImpl& getImplInContext(Context& context);
  ContextImpl& getContextImpl(Context& context);

 private:
  int forceGroup;
};
}  // namespace OpenMM
