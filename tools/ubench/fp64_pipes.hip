// Micro-benchmark: fp64 VALU FMA rate, fp64 MFMA (v_mfma_f64_16x16x4_f64) rate, and whether the two
// pipes overlap when one wave issues both.  Decides whether moving the z z^T accumulation of the
// moments kernel onto f64 MFMA (with psi staying on the VALU) can pay.
//   hipcc --offload-arch=gfx950 -O3 fp64_pipes.hip -o fp64_pipes && ./fp64_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: VALU only, 1: MFMA only, 2: both interleaved
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  double a = seed + threadIdx.x * 1e-9, b = 1.0 - 1e-9;
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = i;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int it = 0; it < iters; ++it) {
    if (MODE != 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], b, a);      // 64 independent-ish FMAs
    }
    if (MODE != 0) {                                                            // 4 MFMAs = 4*2048 flop
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  s += c0[0] + c1[1] + c2[2] + c3[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(int blocks, int iters, double* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 64, 0.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3;
}

int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * 4096);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {                 // waves per SIMD (256-thread block = 1 wave per SIMD)
    const int blocks = 256 * wps;
    const double tv = run<0>(blocks, iters, out), tm = run<1>(blocks, iters, out), tb = run<2>(blocks, iters, out);
    const double waves = blocks * 4.0;
    const double fv = waves * iters * 64.0 * 64 * 2, fm = waves * iters * 4.0 * 2048;
    printf("waves/SIMD %d: VALU %.3f ms = %.1f TF | MFMA %.3f ms = %.1f TF | both %.3f ms = %.1f TF (sum of parts %.3f ms, max %.3f ms)\n",
           wps, tv * 1e3, fv / tv / 1e12, tm * 1e3, fm / tm / 1e12, tb * 1e3, (fv + fm) / tb / 1e12, (tv + tm) * 1e3,
           (tv > tm ? tv : tm) * 1e3);
  }
  return 0;
}
