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

// One scalar load from each of the first LINES 64-byte lines of the kernel's argument block, in ONE batch with one wait.
// A launch's arguments were written by the host a moment ago, so a line's first read on an XCD goes to memory (0.7 us,
// timing build of factor_fused_kernel; with the arguments in host memory, HIP_FORCE_DEV_KERNARG=0, a C3 iteration of four
// launches takes 11 us longer), and the compiler fetches a field where it is first needed, behind the branch before it:
// the prologue of the fused kernel was six such round trips one after the other.  After this pass they hit the scalar cache.
// (Spelled out as one asm block: left to the compiler the same loads are issued in batches of eight with a wait each, and
// loads split over several asm blocks would leave registers with a load in flight that the allocator believes to be free.)
// LINES = ceil(sizeof(arguments) / 64), a specialisation per kernel that uses it.
template <int LINES> __device__ __forceinline__ void kernarg_warm();
template <> __device__ __forceinline__ void kernarg_warm<7>() {
#ifndef GVI_NO_KERNARG_WARM      // (A/B build)
  unsigned w[7];
  asm volatile(
               "s_load_dword %0, %7, 0x0\n"
               "s_load_dword %1, %7, 0x40\n"
               "s_load_dword %2, %7, 0x80\n"
               "s_load_dword %3, %7, 0xc0\n"
               "s_load_dword %4, %7, 0x100\n"
               "s_load_dword %5, %7, 0x140\n"
               "s_load_dword %6, %7, 0x180\n"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(w[0]), "=&s"(w[1]), "=&s"(w[2]), "=&s"(w[3]), "=&s"(w[4]), "=&s"(w[5]), "=&s"(w[6])
               : "s"(__builtin_amdgcn_kernarg_segment_ptr()));
#endif
}
template <> __device__ __forceinline__ void kernarg_warm<13>() {
#ifndef GVI_NO_KERNARG_WARM      // (A/B build)
  unsigned w[13];
  asm volatile(
               "s_load_dword %0, %13, 0x0\n"
               "s_load_dword %1, %13, 0x40\n"
               "s_load_dword %2, %13, 0x80\n"
               "s_load_dword %3, %13, 0xc0\n"
               "s_load_dword %4, %13, 0x100\n"
               "s_load_dword %5, %13, 0x140\n"
               "s_load_dword %6, %13, 0x180\n"
               "s_load_dword %7, %13, 0x1c0\n"
               "s_load_dword %8, %13, 0x200\n"
               "s_load_dword %9, %13, 0x240\n"
               "s_load_dword %10, %13, 0x280\n"
               "s_load_dword %11, %13, 0x2c0\n"
               "s_load_dword %12, %13, 0x300\n"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(w[0]), "=&s"(w[1]), "=&s"(w[2]), "=&s"(w[3]), "=&s"(w[4]), "=&s"(w[5]), "=&s"(w[6]), "=&s"(w[7]), "=&s"(w[8]), "=&s"(w[9]), "=&s"(w[10]), "=&s"(w[11]), "=&s"(w[12])
               : "s"(__builtin_amdgcn_kernarg_segment_ptr()));
#endif
}
template <> __device__ __forceinline__ void kernarg_warm<14>() {
#ifndef GVI_NO_KERNARG_WARM      // (A/B build)
  unsigned w[14];
  asm volatile(
               "s_load_dword %0, %14, 0x0\n"
               "s_load_dword %1, %14, 0x40\n"
               "s_load_dword %2, %14, 0x80\n"
               "s_load_dword %3, %14, 0xc0\n"
               "s_load_dword %4, %14, 0x100\n"
               "s_load_dword %5, %14, 0x140\n"
               "s_load_dword %6, %14, 0x180\n"
               "s_load_dword %7, %14, 0x1c0\n"
               "s_load_dword %8, %14, 0x200\n"
               "s_load_dword %9, %14, 0x240\n"
               "s_load_dword %10, %14, 0x280\n"
               "s_load_dword %11, %14, 0x2c0\n"
               "s_load_dword %12, %14, 0x300\n"
               "s_load_dword %13, %14, 0x340\n"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(w[0]), "=&s"(w[1]), "=&s"(w[2]), "=&s"(w[3]), "=&s"(w[4]), "=&s"(w[5]), "=&s"(w[6]), "=&s"(w[7]), "=&s"(w[8]), "=&s"(w[9]), "=&s"(w[10]), "=&s"(w[11]), "=&s"(w[12]), "=&s"(w[13])
               : "s"(__builtin_amdgcn_kernarg_segment_ptr()));
#endif
}
template <> __device__ __forceinline__ void kernarg_warm<26>() {
#ifndef GVI_NO_KERNARG_WARM      // (A/B build)
  unsigned w[26];
  asm volatile(
               "s_load_dword %0, %26, 0x0\n"
               "s_load_dword %1, %26, 0x40\n"
               "s_load_dword %2, %26, 0x80\n"
               "s_load_dword %3, %26, 0xc0\n"
               "s_load_dword %4, %26, 0x100\n"
               "s_load_dword %5, %26, 0x140\n"
               "s_load_dword %6, %26, 0x180\n"
               "s_load_dword %7, %26, 0x1c0\n"
               "s_load_dword %8, %26, 0x200\n"
               "s_load_dword %9, %26, 0x240\n"
               "s_load_dword %10, %26, 0x280\n"
               "s_load_dword %11, %26, 0x2c0\n"
               "s_load_dword %12, %26, 0x300\n"
               "s_load_dword %13, %26, 0x340\n"
               "s_load_dword %14, %26, 0x380\n"
               "s_load_dword %15, %26, 0x3c0\n"
               "s_load_dword %16, %26, 0x400\n"
               "s_load_dword %17, %26, 0x440\n"
               "s_load_dword %18, %26, 0x480\n"
               "s_load_dword %19, %26, 0x4c0\n"
               "s_load_dword %20, %26, 0x500\n"
               "s_load_dword %21, %26, 0x540\n"
               "s_load_dword %22, %26, 0x580\n"
               "s_load_dword %23, %26, 0x5c0\n"
               "s_load_dword %24, %26, 0x600\n"
               "s_load_dword %25, %26, 0x640\n"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(w[0]), "=&s"(w[1]), "=&s"(w[2]), "=&s"(w[3]), "=&s"(w[4]), "=&s"(w[5]), "=&s"(w[6]), "=&s"(w[7]), "=&s"(w[8]), "=&s"(w[9]), "=&s"(w[10]), "=&s"(w[11]), "=&s"(w[12]), "=&s"(w[13]), "=&s"(w[14]), "=&s"(w[15]), "=&s"(w[16]), "=&s"(w[17]), "=&s"(w[18]), "=&s"(w[19]), "=&s"(w[20]), "=&s"(w[21]), "=&s"(w[22]), "=&s"(w[23]), "=&s"(w[24]), "=&s"(w[25])
               : "s"(__builtin_amdgcn_kernarg_segment_ptr()));
#endif
}

}  // namespace gvi
