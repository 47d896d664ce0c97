#pragma once
namespace OpenMM {
class Vec3 {
 public:
  Vec3() : data{0, 0, 0} {}
  Vec3(double x, double y, double z) : data{x, y, z} {}
  double operator[](int i) const { return data[i]; }
  double& operator[](int i) { return data[i]; }

 private:
  double data[3];
};
}  // namespace OpenMM
