#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <string>
#include <vector>
#include "openmm/OpenMMException.h"
namespace OpenMM {
// device array of OpenMM's HIP platform: the part a force kernel uses
class HipArray {
 public:
  HipArray() : pointer(nullptr), size(0), elementSize(0) {}
  ~HipArray() {
    if (pointer) (void)hipFree(pointer);
  }
  HipArray(const HipArray&) = delete;
  void initialize(size_t size, int elementSize, const std::string& name) {
    if (pointer) (void)hipFree(pointer);
    this->size = size;
    this->elementSize = elementSize;
    this->name = name;
    if (hipMalloc(&pointer, size * elementSize) != hipSuccess) throw OpenMMException("Error creating array " + name);
    (void)hipMemset(pointer, 0, size * elementSize);
  }
  bool isInitialized() const { return pointer != nullptr; }
  size_t getSize() const { return size; }
  int getElementSize() const { return elementSize; }
  const std::string& getName() const { return name; }
  void* getDevicePointer() { return pointer; }
  template <class T>
  void upload(const std::vector<T>& data) {
    if (sizeof(T) != (size_t)elementSize || data.size() != size) throw OpenMMException("Error uploading array " + name + ": wrong size");
    if (hipMemcpy(pointer, data.data(), size * elementSize, hipMemcpyHostToDevice) != hipSuccess) throw OpenMMException("Error uploading array " + name);
  }
  template <class T>
  void download(std::vector<T>& data) {
    if (sizeof(T) != (size_t)elementSize) throw OpenMMException("Error downloading array " + name + ": wrong element size");
    data.resize(size);
    if (hipMemcpy(data.data(), pointer, size * elementSize, hipMemcpyDeviceToHost) != hipSuccess) throw OpenMMException("Error downloading array " + name);
  }

 private:
  void* pointer;
  size_t size;
  int elementSize;
  std::string name;
};
}  // namespace OpenMM
