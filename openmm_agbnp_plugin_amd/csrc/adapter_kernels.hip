// Data-convention adapters between an OpenMM GPU ComputeContext and the engine (GPU-platform side of the boundary).
//
// What the reference's GPU platform works on (platforms/opencl/src/OpenCLAGBNPKernels.cpp:541-556 and
// platforms/opencl/src/kernels/GVolReduceTree.cl:92-121):
//   posq          real4 per atom {x, y, z, q} in the CONTEXT's atom order (OpenMM reorders atoms for locality;
//                 atomIndex[slot] = the particle's index in the System / Force), float or double; in mixed
//                 precision a second float4 array holds the low-order parts (posqCorrection)
//   force buffer  64-bit fixed point, value * 2^32, three planes [x: 0..P) [y: P..2P) [z: 2P..3P) over the
//                 padded atom count P, context order, accumulated with integer atomics (GVolReduceTree.cl:117-119)
//   energy buffer per-thread slots of the context's energy accumulator, float or double (GVolReduceTree.cl:112)
// The engine computes in FP64 on [3N] xyz arrays in the Force's particle order, so the glue is two small kernels:
//   k_adapt_positions   posq[slot] (+ correction) -> xyz[3 * atomIndex[slot]]
//   k_adapt_outputs     xyz forces of particle atomIndex[slot] -> fixed-point adds at slot; energy -> one slot;
//                       the staging buffers are handed back as zeros (the engine ADDS into them, and adds nothing
//                       when an evaluation overflowed: such an evaluation then adds nothing to the context either)
#include <hip/hip_runtime.h>

#include "adapter_kernels.h"

namespace agbnp {

template <class Real4>
__global__ __launch_bounds__(256) void k_adapt_positions(int n, const Real4* __restrict__ posq, const float4* __restrict__ correction,
                                                         const int* __restrict__ atom_index, double* __restrict__ xyz) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n) return;
  const Real4 p = posq[slot];
  double x = (double)p.x, y = (double)p.y, z = (double)p.z;
  if (correction) {  // mixed precision: position = posq + posqCorrection, both float
    const float4 c = correction[slot];
    x += (double)c.x;
    y += (double)c.y;
    z += (double)c.z;
  }
  const int i = atom_index ? atom_index[slot] : slot;
  xyz[3 * i] = x;
  xyz[3 * i + 1] = y;
  xyz[3 * i + 2] = z;
}

// round-to-nearest conversion to the context's 2^32 fixed point (OpenMM: (long long)(f * 0x100000000))
__device__ __forceinline__ unsigned long long to_fixed(double f) { return (unsigned long long)(long long)rint(f * 4294967296.0); }

__global__ __launch_bounds__(256) void k_adapt_outputs(int n, int padded, const int* __restrict__ atom_index, double* __restrict__ force_xyz,
                                                       double* __restrict__ energy, unsigned long long* __restrict__ force_fixed,
                                                       void* __restrict__ energy_buffer, int energy_is_double, int energy_slot) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot == 0 && energy_buffer) {
    const double e = energy[0];
    energy[0] = 0.0;
    if (energy_is_double)
      static_cast<double*>(energy_buffer)[energy_slot] += e;
    else
      static_cast<float*>(energy_buffer)[energy_slot] += (float)e;
  }
  if (slot >= n) return;
  const int i = atom_index ? atom_index[slot] : slot;
  const double fx = force_xyz[3 * i], fy = force_xyz[3 * i + 1], fz = force_xyz[3 * i + 2];
  force_xyz[3 * i] = 0.0;
  force_xyz[3 * i + 1] = 0.0;
  force_xyz[3 * i + 2] = 0.0;
  atomicAdd(&force_fixed[slot], to_fixed(fx));
  atomicAdd(&force_fixed[slot + padded], to_fixed(fy));
  atomicAdd(&force_fixed[slot + 2 * padded], to_fixed(fz));
}

hipError_t launch_adapt_positions(int n, const void* posq, int posq_is_double, const void* correction, const int* atom_index, double* xyz,
                                  hipStream_t st) {
  const dim3 grid((n + 255) / 256), block(256);
  if (posq_is_double)
    hipLaunchKernelGGL(k_adapt_positions<double4>, grid, block, 0, st, n, static_cast<const double4*>(posq),
                       static_cast<const float4*>(nullptr), atom_index, xyz);
  else
    hipLaunchKernelGGL(k_adapt_positions<float4>, grid, block, 0, st, n, static_cast<const float4*>(posq),
                       static_cast<const float4*>(correction), atom_index, xyz);
  return hipGetLastError();
}

hipError_t launch_adapt_outputs(int n, int padded, const int* atom_index, double* force_xyz, double* energy, unsigned long long* force_fixed,
                                void* energy_buffer, int energy_is_double, int energy_slot, hipStream_t st) {
  hipLaunchKernelGGL(k_adapt_outputs, dim3((n + 255) / 256), dim3(256), 0, st, n, padded, atom_index, force_xyz, energy, force_fixed,
                     energy_buffer, energy_is_double, energy_slot);
  return hipGetLastError();
}

}  // namespace agbnp
