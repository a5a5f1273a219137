// Shared device helpers.
#pragma once
#include <hip/hip_runtime.h>

namespace gvi {

// One-wave workgroups (blockDim == 64): LDS operations of a wave execute in order, so a wave-level
// "barrier" only has to drain the LDS queue and stop the compiler from moving memory operations
// across it.  Unlike __syncthreads() it does not wait for outstanding global loads / stores (vmcnt)
// -- which made every barrier of the first sequential chain kernels cost a memory round trip.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// LDS-only workgroup barrier: unlike __syncthreads() it does not drain outstanding global loads / stores
// (vmcnt); everything the phases exchange goes through LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace gvi
