#pragma once
#include <hip/hip_runtime_api.h>
#include <vector>
#include "openmm/hip/HipArray.h"
namespace OpenMM {
// per-device compute context of OpenMM's HIP platform: the accessors a force kernel reads.  Atom ORDER: slot s of
// every per-atom array holds particle getAtomIndex()[s] of the System (OpenMM reorders atoms for locality); arrays are
// padded to getPaddedNumAtoms() (a multiple of 32).
class HipContext {
 public:
  static const int TileSize = 32;
  HipContext(int numAtoms, int deviceIndex, bool useDoublePrecision, bool useMixedPrecision)
      : numAtoms(numAtoms), paddedNumAtoms((numAtoms + TileSize - 1) / TileSize * TileSize), deviceIndex(deviceIndex),
        useDoublePrecision(useDoublePrecision), useMixedPrecision(useMixedPrecision), stream(nullptr) {
    if (hipSetDevice(deviceIndex) != hipSuccess) throw OpenMMException("Error setting the HIP device");
    if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) throw OpenMMException("Error creating the HIP stream");
    const int real = useDoublePrecision ? 8 : 4, mixed = (useDoublePrecision || useMixedPrecision) ? 8 : 4;
    posq.initialize(paddedNumAtoms, 4 * real, "posq");
    if (useMixedPrecision) posqCorrection.initialize(paddedNumAtoms, 16, "posqCorrection");
    force.initialize(3 * (size_t)paddedNumAtoms, 8, "force");
    energyBuffer.initialize(1024, mixed, "energyBuffer");
    atomIndexDevice.initialize(paddedNumAtoms, 4, "atomIndex");
    atomIndex.resize(paddedNumAtoms);
    for (int i = 0; i < paddedNumAtoms; i++) atomIndex[i] = i;
    atomIndexDevice.upload(atomIndex);
  }
  ~HipContext() {
    if (stream) (void)hipStreamDestroy(stream);
  }
  int getNumAtoms() const { return numAtoms; }
  int getPaddedNumAtoms() const { return paddedNumAtoms; }
  int getDeviceIndex() const { return deviceIndex; }
  bool getUseDoublePrecision() const { return useDoublePrecision; }
  bool getUseMixedPrecision() const { return useMixedPrecision; }
  hipStream_t getCurrentStream() { return stream; }
  HipArray& getPosq() { return posq; }
  HipArray& getPosqCorrection() { return posqCorrection; }
  HipArray& getLongForceBuffer() { return force; }
  HipArray& getEnergyBuffer() { return energyBuffer; }
  HipArray& getAtomIndexArray() { return atomIndexDevice; }
  const std::vector<int>& getAtomIndex() const { return atomIndex; }
  void setAtomIndex(const std::vector<int>& index) {  // (the real context does this in reorderAtoms())
    atomIndex = index;
    atomIndexDevice.upload(atomIndex);
  }

 private:
  int numAtoms, paddedNumAtoms, deviceIndex;
  bool useDoublePrecision, useMixedPrecision;
  hipStream_t stream;
  HipArray posq, posqCorrection, force, energyBuffer, atomIndexDevice;
  std::vector<int> atomIndex;
};
}  // namespace OpenMM
