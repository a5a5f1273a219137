// Micro-benchmark: throughput of LDS fp64 accumulation per CU, to decide how moments_orbit_kernel folds an orbit into
// the factor's accumulators.
//   atomic D C : ds_add_f64, lane -> entry (lane % D), C private copies of every entry selected by lane % C
//                (layout [entry][copy]); D * C distinct addresses, same-address multiplicity 64 / lcm-ish
//   rmw        : ds_read_b64 + v_add_f64 + ds_write_b64 on 64 distinct addresses (what a conflict-free design could use)
// One block of 256 threads per CU (4 waves share the CU's LDS), ITER instructions per wave; prints cycles per wave
// instruction per CU at the reported clock.
//   hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic && ./lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_atomic(double* out, int iters, int D, int C, int scramble) {
  __shared__ double acc[4][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int e = lane; e < 1024; e += 64) acc[wave][e] = 0.0;
  __syncthreads();
  int entry = lane % D;
  if (scramble) entry = (lane * 7 + (lane >> 3)) % D;
  const int copy = lane % C;
  double* p = &acc[wave][entry * C + copy];
  double v = 1.0 + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  double s = 0;
  for (int e = lane; e < 1024; e += 64) s += acc[wave][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_rmw(double* out, int iters) {
  __shared__ double acc[4][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int e = lane; e < 1024; e += 64) acc[wave][e] = 0.0;
  __syncthreads();
  volatile double* p = &acc[wave][lane];
  double v = 1.0 + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { p[u * 64] = p[u * 64] + v; }
  }
  __syncthreads();
  double s = 0;
  for (int e = lane; e < 1024; e += 64) s += acc[wave][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  double* out;
  hipMalloc(&out, (size_t)cus * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  auto report = [&](const char* name, int D, int C, int scr, float ms) {
    const double instr_per_cu = 4.0 * iters * 8;                 // wave instructions that went through one CU's LDS
    std::printf("%-8s D=%2d C=%2d scr=%d  %8.3f ms  %7.1f cycles per wave instruction per CU (%.2f GHz)\n", name, D, C, scr, ms,
                ms * 1e-3 * ghz * 1e9 / instr_per_cu, ghz);
  };
  for (int scr = 0; scr < 2; ++scr)
    for (int D : {1, 2, 4, 8, 12, 16, 32, 64})
      for (int C : {1, 4, 8, 16}) {
        if (D * C > 1024) continue;
        hipLaunchKernelGGL(k_atomic, dim3(cus), dim3(256), 0, 0, out, 10, D, C, scr);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_atomic, dim3(cus), dim3(256), 0, 0, out, iters, D, C, scr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        report("atomic", D, C, scr, ms);
      }
  hipLaunchKernelGGL(k_rmw, dim3(cus), dim3(256), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_rmw, dim3(cus), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  report("rmw", 64, 1, 0, ms);
  return 0;
}
