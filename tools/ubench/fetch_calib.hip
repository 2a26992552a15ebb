// fetch_calib.hip — calibrates rocprofv3 FETCH_SIZE for the access pattern of sw_score_kernel's
// stage_load (four 1-byte global loads per lane, 16 lanes covering 64 contiguous bytes), on a buffer
// far larger than L2 + Infinity Cache, each byte read exactly once.  MI355X_MICROARCH.md §HBM:
// "Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void read_bytes(const uint8_t* __restrict__ p, size_t nseg, unsigned* out) {
  const int l16 = threadIdx.x & 15;
  size_t slot = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const size_t nslots = ((size_t)gridDim.x * 256) >> 4;
  unsigned acc = 0;
  for (size_t s = slot; s < nseg; s += nslots) {
    const uint8_t* q = p + s * 64 + 4 * l16;
    acc += (unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16) | ((unsigned)q[3] << 24);
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const size_t bytes = (size_t)4 << 30;
  uint8_t* d; unsigned* o;
  if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 1, bytes);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(read_bytes, dim3(256 * 8), dim3(256), 0, 0, d, bytes / 64, o);
  hipDeviceSynchronize();
  printf("read_bytes: %zu bytes requested once each\n", bytes);
  return 0;
}
