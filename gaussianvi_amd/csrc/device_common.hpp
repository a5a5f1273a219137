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

// Publish a scalar result to host-mapped (fine-grained, uncached) memory.
// FAST form (default): {value, sequence} go out in ONE 16-byte store, so the host, which spins on the sequence word, can never
// see a new sequence with an old value -- without the system-scope fence a two-store protocol needs.
// (__threadfence_system() writes back the whole L2: behind a kernel that has just written megabytes of results it cost
// ~50 us, measured on the fused epilogue tail.)  That a 16-byte store reaches host memory as one unit is how this part
// behaves, not something the HIP memory model promises.
// CHECKED form (option "safe_publish"; the pointer then carries bit 0 as a tag, see pub_slot() on the host): the slot is
// [sequence | value | value | sequence], every word a system-scope atomic store, the values before the sequence words with a
// system fence in between; the host takes the value only when both sequence words and both value copies agree, so a torn or
// reordered publish is seen as "not yet there" instead of as a wrong cost.  Four doubles per slot in both forms.
// stress test of the hand-over (gvi_debug_cost_log): every publish also leaves its value at gvi_dbg_log[(int)seq & gvi_dbg_mask]
__device__ double* gvi_dbg_log = nullptr;
__device__ int gvi_dbg_mask = 0;

__device__ __forceinline__ void publish_to_host(double* host_out_tagged, double value, double seq) {
  if (double* lg = gvi_dbg_log) lg[(int)seq & gvi_dbg_mask] = value;
  const unsigned long long bits = (unsigned long long)host_out_tagged;
  double* host_out = (double*)(bits & ~1ull);
  if (bits & 1ull) {
    __threadfence_system();
    __hip_atomic_store(host_out + 1, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host_out + 2, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __hip_atomic_store(host_out + 0, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host_out + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
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

// The same predicate, checked LATE: pred_issue requests the word at the kernel's first instruction, the kernel goes on with
// loads that have no side effects (its load phase), and pred_fail -- placed in front of the first store to memory -- waits for
// it.  A predicate load in front of everything else is a dependent ~1 us round trip at the start of every launch of a
// pipelined iteration (measured: three chain launches whose predicate fails take 10.2 us, 3.4 us each).
struct LazyPred { double v, want; bool has; };
__device__ __forceinline__ LazyPred pred_issue(const double* pred, double val) {
  LazyPred p;
  p.want = val; p.has = pred != nullptr; p.v = val;
  if (p.has) p.v = __hip_atomic_load(pred, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return p;
}
__device__ __forceinline__ bool pred_fail(const LazyPred& p) { return p.has && p.v != p.want; }

}  // namespace gvi
