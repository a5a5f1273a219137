// Micro-benchmark: fp64 VALU FMA rate, fp64 MFMA (v_mfma_f64_16x16x4_f64) rate with 4 / 8 / 16 independent accumulator
// tiles per wave, and whether the two pipes overlap when one wave issues both.  Decides whether moving the z z^T
// accumulation of the moments kernel onto f64 MFMA (with psi staying on the VALU) can pay.
//   hipcc --offload-arch=gfx950 -O3 fp64_pipes.hip -o fp64_pipes && ./fp64_pipes [mode]
//   mode (for counter runs: ONE kernel variant per process, so rocprofv3 --pmc attributes cleanly):
//     all (default) | valu | mfma4 | mfma8 | mfma16 | both
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: VALU only, 1: MFMA only, 2: both interleaved; NACC independent MFMA accumulator tiles (dependency distance)
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  double a = seed + threadIdx.x * 1e-9, b = 1.0 - 1e-9;
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = i;
  d4 c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) c[i] = d4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if (MODE != 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], b, a);      // 64 FMAs, dependency distance 16
    }
    if (MODE != 0) {                                                            // 16 MFMAs = 16 * 2048 flop
#pragma unroll
      for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += c[i][i & 3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NACC>
double run(int blocks, int iters, double* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, out, 64, 0.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "all";
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * 4096);
  const int iters = 5000;
  const bool all = !strcmp(mode, "all");
  for (int wps = 1; wps <= 4; wps *= 2) {                 // waves per SIMD (256-thread block = 1 wave per SIMD)
    const int blocks = 256 * wps;
    const double waves = blocks * 4.0;
    const double fv = waves * iters * 64.0 * 64 * 2, fm = waves * iters * 16.0 * 2048;
    if (all || !strcmp(mode, "valu")) { const double t = run<0, 4>(blocks, iters, out); printf("waves/SIMD %d: VALU v_fma_f64            %.3f ms = %.1f TF\n", wps, t * 1e3, fv / t / 1e12); }
    if (all || !strcmp(mode, "mfma4")) { const double t = run<1, 4>(blocks, iters, out); printf("waves/SIMD %d: MFMA f64 16x16x4,  4 tiles %.3f ms = %.1f TF\n", wps, t * 1e3, fm / t / 1e12); }
    if (all || !strcmp(mode, "mfma8")) { const double t = run<1, 8>(blocks, iters, out); printf("waves/SIMD %d: MFMA f64 16x16x4,  8 tiles %.3f ms = %.1f TF\n", wps, t * 1e3, fm / t / 1e12); }
    if (all || !strcmp(mode, "mfma16")) { const double t = run<1, 16>(blocks, iters, out); printf("waves/SIMD %d: MFMA f64 16x16x4, 16 tiles %.3f ms = %.1f TF\n", wps, t * 1e3, fm / t / 1e12); }
    if (all || !strcmp(mode, "both")) { const double t = run<2, 8>(blocks, iters, out); printf("waves/SIMD %d: VALU + MFMA (8 tiles) interleaved %.3f ms = %.1f TF total\n", wps, t * 1e3, (fv + fm) / t / 1e12); }
  }
  return 0;
}
