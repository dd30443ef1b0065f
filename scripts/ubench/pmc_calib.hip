// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access pattern of register spills (scratch_load/store_dword:
// one dword per lane, 256 B per wave instruction), as MI355X_MICROARCH.md asks before trusting an absolute byte count
// ("other access widths are uncalibrated").  Streams a 1 GiB buffer (>> L2 and Infinity Cache): read_dword reads
// 4 B/lane, write_dword writes 4 B/lane, read_x4 / write_x4 use 16 B/lane (the documented reference points).
//   hipcc --offload-arch=gfx950 -O3 pmc_calib.hip -o pmc_calib && rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./pmc_calib
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void read_dword(const float* __restrict__ in, float* out, size_t n) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void write_dword(float* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)i;
}
__global__ void read_x4(const float4* __restrict__ in, float* out, size_t n4) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = in[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void write_x4(float4* out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    out[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}

int main() {
  const size_t bytes = (size_t)1 << 30, n = bytes / 4;
  float *a, *b;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
  hipMemset(a, 0, bytes);
  hipMemset(b, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    read_dword<<<4096, 256>>>(a, b, n);
    write_dword<<<4096, 256>>>(b, n);
    read_x4<<<4096, 256>>>((const float4*)a, b, n / 4);
    write_x4<<<4096, 256>>>((float4*)b, n / 4);
  }
  hipDeviceSynchronize();
  printf("each kernel moves %zu bytes\n", bytes);
  return 0;
}
