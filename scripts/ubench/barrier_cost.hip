// What does one workgroup-wide exchange through LDS cost on gfx950?  512 threads (8 wavefronts), one workgroup per CU:
//   variant 0: s_barrier only;  1: one lane per wavefront writes LDS, barrier, every lane reads it back (the shape of a pivot step's exchange);
//   2: as 1 with 32 dependent v_fma_f64 behind the read;  3: as 1 but only wavefront (i % 8) writes (owner pattern) and all read 8 doubles.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/barrier_cost.hip -o scripts/ubench/_build/barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V>
__global__ __launch_bounds__(512) void k(int iters, double* out, unsigned long long* cyc) {
  __shared__ double sh[2][1024];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double acc = tid * 1e-3;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (V == 1 || V == 2) { if (lane == 0) sh[i & 1][wave] = acc; }
    if (V == 3) { if (wave == (i & 7)) { for (int s = 0; s < 8; ++s) sh[i & 1][lane + 64 * s] = acc + s; } }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (V == 1 || V == 2) acc += sh[i & 1][(wave + 1) & 7];
    if (V == 2) { for (int c = 0; c < 32; ++c) acc = __builtin_fma(acc, 1.0000001, 1e-9); }
    if (V == 3) { double s8 = 0; for (int s = 0; s < 8; ++s) s8 += sh[i & 1][lane + 64 * s]; acc += s8; }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + tid] = acc;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 8 * 512 * 256); hipMalloc(&cyc, 8);
  const int iters = 4096;
  for (int v = 0; v < 4; ++v) {
    for (int nb : {1, 256}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (v == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(512), 0, 0, iters, out, cyc);
        if (v == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(512), 0, 0, iters, out, cyc);
        if (v == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(512), 0, 0, iters, out, cyc);
        if (v == 3) hipLaunchKernelGGL(k<3>, dim3(nb), dim3(512), 0, 0, iters, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("variant %d, %3d workgroups: %.3f us per iteration, %.0f shader cycles per iteration (thread 0)\n", v, nb, ms * 1e3 / iters, (double)c / iters);
    }
  }
  return 0;
}
