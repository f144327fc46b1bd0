// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access pattern of the lane-per-member kernels: 8 bytes per lane, batch-fastest
// (a wavefront touches one 512-byte segment per access, successive accesses of a lane are nb x 8 bytes apart).  MI355X_MICROARCH.md calibrates 16-byte-per-lane
// streaming reads only (FETCH_SIZE = 1/2 of the bytes) and calls other widths and WRITE_SIZE uncalibrated.  Every kernel moves a KNOWN number of bytes over
// buffers far larger than the 256 MB Infinity Cache; scripts/hbm_calib.sh runs one --pmc pass per counter and prints counter bytes / known bytes.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_calib hbm_calib.hip && ./hbm_calib [GiB per buffer]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// batch-fastest: element i of lane b at x[i * nb + b]; every lane walks rows i = 0 .. rows-1
__global__ void read8(const double* __restrict__ x, double* __restrict__ sink, long nb, int rows) {
  const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
  double acc = 0.0;
#pragma unroll 4
  for (int i = 0; i < rows; ++i) acc += x[(long)i * nb + b];
  if (acc == 12345.678) sink[b] = acc;  // never true: keeps the loads
}
__global__ void write8(double* __restrict__ y, long nb, int rows) {
  const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
#pragma unroll 4
  for (int i = 0; i < rows; ++i) y[(long)i * nb + b] = (double)i;
}
__global__ void copy8(const double* __restrict__ x, double* __restrict__ y, long nb, int rows) {
  const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
#pragma unroll 4
  for (int i = 0; i < rows; ++i) y[(long)i * nb + b] = x[(long)i * nb + b] * 1.5;
}
__global__ void rmw8(double* __restrict__ y, long nb, int rows) {  // read-modify-write of the same address (the difference-array update)
  const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
#pragma unroll 4
  for (int i = 0; i < rows; ++i) y[(long)i * nb + b] += 1.0;
}
__global__ void read16(const double2* __restrict__ x, double* __restrict__ sink, long nb, int rows) {  // the guide's calibrated pattern, for reference
  const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
  double acc = 0.0;
#pragma unroll 4
  for (int i = 0; i < rows; ++i) { double2 v = x[(long)i * nb + b]; acc += v.x + v.y; }
  if (acc == 12345.678) sink[b] = acc;
}
__global__ void write16(double2* __restrict__ y, long nb, int rows) {
  const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
#pragma unroll 4
  for (int i = 0; i < rows; ++i) y[(long)i * nb + b] = make_double2((double)i, 1.0);
}
// int32 pivots next to doubles: 4 bytes per lane (256-byte segments)
__global__ void write4(int* __restrict__ y, long nb, int rows) {
  const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
#pragma unroll 4
  for (int i = 0; i < rows; ++i) y[(long)i * nb + b] = i;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 4.0;
  const long nb = 262144;  // lanes (BASELINE config 4's ensemble)
  const int rows = (int)(gib * 1073741824.0 / (8.0 * nb));
  const size_t bytes = (size_t)rows * nb * 8;
  double *x, *y, *sink;
  CK(hipMalloc(&x, 2 * bytes)); CK(hipMalloc(&y, 2 * bytes)); CK(hipMalloc(&sink, nb * 8));
  CK(hipMemset(x, 0, 2 * bytes)); CK(hipMemset(y, 0, 2 * bytes));
  const dim3 g(nb / 256), blk(256);
  printf("nb=%ld rows=%d bytes_per_pass=%zu\n", nb, rows, bytes);
  for (int rep = 0; rep < 2; ++rep) {
    read8<<<g, blk>>>(x, sink, nb, rows);
    write8<<<g, blk>>>(y, nb, rows);
    copy8<<<g, blk>>>(x, y, nb, rows);
    rmw8<<<g, blk>>>(y, nb, rows);
    read16<<<g, blk>>>((const double2*)x, sink, nb, rows);
    write16<<<g, blk>>>((double2*)y, nb, rows);
    write4<<<g, blk>>>((int*)y, nb, rows);
    CK(hipDeviceSynchronize());
  }
  // timing (outside the profiler): GB/s of each pattern
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* name, double nbytes, auto&& f) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-8s %8.3f ms  %8.1f GB/s\n", name, ms, nbytes / ms / 1e6);
  };
  timeit("read8", bytes, [&] { read8<<<g, blk>>>(x, sink, nb, rows); });
  timeit("write8", bytes, [&] { write8<<<g, blk>>>(y, nb, rows); });
  timeit("copy8", 2.0 * bytes, [&] { copy8<<<g, blk>>>(x, y, nb, rows); });
  timeit("rmw8", 2.0 * bytes, [&] { rmw8<<<g, blk>>>(y, nb, rows); });
  timeit("read16", 2.0 * bytes, [&] { read16<<<g, blk>>>((const double2*)x, sink, nb, rows); });
  timeit("write16", 2.0 * bytes, [&] { write16<<<g, blk>>>((double2*)y, nb, rows); });
  timeit("write4", 0.5 * bytes, [&] { write4<<<g, blk>>>((int*)y, nb, rows); });
  return 0;
}
