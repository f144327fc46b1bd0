// Micro-benchmark (gfx950): what v_permlane16_swap_b32 / v_permlane32_swap_b32 cost next to a dependent FP64 chain (one wavefront per SIMD) — can operands be
// handed from the rows 1..3 of a wavefront to row 0 in the shadow of the chain's latency?
//   hipcc -O3 --offload-arch=gfx950 -o scripts/ubench/_build/permlane_swap scripts/ubench/permlane_swap.hip
// Prints cycles per iteration of { 2 dependent v_fma_f64 + NS swaps (independent of the chain) } for NS = 0, 1, 2, 3, 4, 6, 8, and of NS swaps alone.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NS, int KIND, bool CHAIN>
__global__ __launch_bounds__(64) void k_mix(double* out, unsigned* out2, int iters, double a, double b) {
  double x = a + (double)threadIdx.x;
  unsigned u[8], w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { u[i] = threadIdx.x * (i + 3); w[i] = threadIdx.x + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (CHAIN) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        if (i == NS / 2 && CHAIN) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(b));
        if (KIND == 0) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[i]), "+v"(w[i]));
        else asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(w[i]));
      }
      if (NS == 0 && CHAIN) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(b));
    }
  }
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += u[i] ^ w[i];
  out[blockIdx.x * 64 + threadIdx.x] = x;
  out2[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NS, int KIND, bool CHAIN>
double run(int iters) {
  const int blocks = 1024;
  double* out; unsigned* out2;
  hipMalloc(&out, sizeof(double) * blocks * 64); hipMalloc(&out2, sizeof(unsigned) * blocks * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_mix<NS, KIND, CHAIN>), dim3(blocks), dim3(64), 0, 0, out, out2, 10, 1.0, 1.0000001);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_mix<NS, KIND, CHAIN>), dim3(blocks), dim3(64), 0, 0, out, out2, iters, 1.0, 1.0000001);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(out); hipFree(out2);
  return ms * 1e-3 * 2.4e9 / ((double)iters * 16);  // cycles per iteration at 2.4 GHz, one wavefront per SIMD
}

int main() {
#define ROW(KIND, name) printf("%s: 2 dependent v_fma_f64 + NS swaps, cycles per iteration: NS=0 %.1f  1 %.1f  2 %.1f  3 %.1f  4 %.1f  6 %.1f  8 %.1f | swaps alone: 4 %.1f  8 %.1f\n", name, \
    run<0, KIND, true>(20000), run<1, KIND, true>(20000), run<2, KIND, true>(20000), run<3, KIND, true>(20000), run<4, KIND, true>(20000), run<6, KIND, true>(20000), run<8, KIND, true>(20000), \
    run<4, KIND, false>(20000), run<8, KIND, false>(20000));
  ROW(0, "v_permlane16_swap_b32")
  ROW(1, "v_permlane32_swap_b32")
  return 0;
}
