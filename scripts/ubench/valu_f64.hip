// Micro-benchmark (gfx950): FP64 VALU issue rate vs dependent-chain latency vs EXEC mask, one or two wavefronts per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/ubench/_build/valu_f64 scripts/ubench/valu_f64.hip
// Prints cycles per wave-instruction for v_add_f64 / v_mul_f64 / v_fma_f64 chains of ILP 1, 2, 4, 8 with 64 / 32 / 16 active lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP, int OP>
__global__ __launch_bounds__(64) void k_chain(double* out, int iters, int active, double a, double b) {
  if ((int)threadIdx.x >= active) return;
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = a + (double)(threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        else if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        else asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ILP, int OP>
double run(int blocks, int active, int iters) {
  double* out;
  hipMalloc(&out, sizeof(double) * blocks * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_chain<ILP, OP>), dim3(blocks), dim3(64), 0, 0, out, 10, active, 1.0, 1.0000001);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_chain<ILP, OP>), dim3(blocks), dim3(64), 0, 0, out, iters, active, 1.0, 1.0000001);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  // cycles per wave-instruction per SIMD at 2.4 GHz: time * 2.4e9 / (instructions per wave * waves per SIMD)
  const double waves_per_simd = (double)blocks / 1024.0;
  return ms * 1e-3 * 2.4e9 / ((double)iters * 16 * ILP * (waves_per_simd < 1 ? 1 : waves_per_simd));
}

int main() {
  const char* opn[3] = {"v_add_f64", "v_mul_f64", "v_fma_f64"};
  for (int blocks : {1024, 2048, 4096}) {
    for (int active : {64, 32, 16}) {
      printf("blocks %d (%.0f waves/SIMD) active lanes %d\n", blocks, blocks / 1024.0, active);
#define ROW(OP) printf("  %-10s ILP1 %.2f  ILP2 %.2f  ILP4 %.2f  ILP8 %.2f cycles/inst\n", opn[OP], run<1, OP>(blocks, active, 20000), run<2, OP>(blocks, active, 20000), run<4, OP>(blocks, active, 20000), run<8, OP>(blocks, active, 20000));
      ROW(0) ROW(1) ROW(2)
    }
  }
  return 0;
}
