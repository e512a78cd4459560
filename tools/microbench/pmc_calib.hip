// pmc_calib.hip — known-size memory streams for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md, "HBM": FETCH_SIZE reports half the bytes of a 16 B/lane streaming read; other widths and the
// write counter are uncalibrated). Four kernels over a 1 GiB buffer (larger than the 256 MiB Infinity Cache):
//   read8_kernel   every lane reads 8 B (one double)      -> the window solver's access width
//   read16_kernel  every lane reads 16 B (double2)        -> the documented case (reports 1/2)
//   write8_kernel  every lane writes 8 B
//   read1_kernel   every lane reads 1 B                   -> the front-end's pixel reads
// Run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE ; true bytes per launch = 2^30.
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr size_t kBytes = 1ull << 30;

__global__ void read8_kernel(const double *p, double *out, size_t n) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void read16_kernel(const double2 *p, double *out, size_t n) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double2 v = p[i];
    s += v.x + v.y;
  }
  if (s == 12345.678) out[0] = s;
}
__global__ void read1_kernel(const unsigned char *p, double *out, size_t n) {
  unsigned s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 0xfffffff1u) out[0] = s;
}
__global__ void write8_kernel(double *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}

int main() {
  double *buf, *out;
  if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  (void)hipMemset(buf, 0, kBytes);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(read8_kernel, dim3(4096), dim3(256), 0, 0, buf, out, kBytes / 8);
    hipLaunchKernelGGL(read16_kernel, dim3(4096), dim3(256), 0, 0, (const double2 *)buf, out, kBytes / 16);
    hipLaunchKernelGGL(read1_kernel, dim3(4096), dim3(256), 0, 0, (const unsigned char *)buf, out, kBytes);
    hipLaunchKernelGGL(write8_kernel, dim3(4096), dim3(256), 0, 0, buf, kBytes / 8);
  }
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  printf("pmc_calib: 3 x (read8, read16, read1, write8) over %zu bytes\n", kBytes);
  return 0;
}
