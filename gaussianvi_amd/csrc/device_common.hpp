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

// Publish a scalar result to host-mapped (fine-grained, uncached) memory: {value, sequence} go out in ONE 16-byte store, so
// the host, which spins on the sequence word, can never see a new sequence with an old value -- without the system-scope
// fence a two-store protocol needs.  (__threadfence_system() writes back the whole L2: behind a kernel that has just
// written megabytes of results it cost ~50 us, measured on the fused epilogue tail.)
__device__ __forceinline__ void publish_to_host(double* host_out, double value, double seq) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 v;
  v.x = value;
  v.y = seq;
  __builtin_nontemporal_store(v, (d2*)host_out);
}

// Predicated launch (gvi_ngd_run): the kernels of a speculatively queued iteration read one device word and return at once
// unless it holds the expected sequence number -- the previous iteration's tail writes it only when its trial was accepted.
// pred == nullptr: unconditional.
__device__ __forceinline__ bool pred_skip(const double* pred, double val) { return pred != nullptr && *pred != val; }

}  // namespace gvi
