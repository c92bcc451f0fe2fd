// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the
// sampler's kernels (MI355X_MICROARCH.md: the counter reads HALF the bytes of a 16-B/lane stream,
// "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Every kernel reads a buffer of known size exactly once; run under
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib
// and compare FETCH_SIZE with the bytes printed here (tools/fetch_calib.sh does both).
//   u16_unit      2 B per lane, unit stride   (meta words, statistics / reset kernels)
//   u16_stride3   2 B per lane, every third   (meta words, colour phases)
//   f64_unit      8 B per lane, unit stride   (tri, jump plane 0 in the streaming kernels)
//   f64_stride3   8 B per lane, every third   (jump planes / tri in the colour phases)
//   f64_sparse16  8 B from one lane in 16     (a jump plane with 1/16 of the slots occupied)
//   x4_unit       16 B per lane, unit stride  (the guide's calibrated case: FETCH_SIZE = bytes / 2)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <class T, int STRIDE, int EVERY>
__global__ void read_kernel(const T *p, size_t n_elems, unsigned long long *sink) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
  unsigned long long acc = 0;
  for (; i * STRIDE < n_elems; i += (size_t)gridDim.x * blockDim.x) {
    if (EVERY > 1 && (threadIdx.x % EVERY) != 0) continue;
    const T v = p[i * STRIDE];
    const unsigned char *b = reinterpret_cast<const unsigned char *>(&v);
    acc += b[0];
  }
  if (acc == 0x7fffffffffffffffull) *sink = acc;   // never true: keeps the loads
}

struct alignas(16) X4 { unsigned int a, b, c, d; };

int main() {
  const size_t bytes = (size_t)1 << 30;     // 1 GiB: four times the Infinity Cache
  void *buf = nullptr;
  unsigned long long *sink = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  const int blocks = 256 * 32, threads = 256;
  // name, bytes the lanes ask for, bytes of the cache lines (128 B) they touch
  std::printf("kernel,bytes_requested,bytes_of_touched_128B_lines\n");
  hipLaunchKernelGGL((read_kernel<uint16_t, 1, 1>), dim3(blocks), dim3(threads), 0, 0, (const uint16_t *)buf, bytes / 2, sink);
  std::printf("u16_unit,%zu,%zu\n", bytes, bytes);
  hipLaunchKernelGGL((read_kernel<uint16_t, 3, 1>), dim3(blocks), dim3(threads), 0, 0, (const uint16_t *)buf, bytes / 2, sink);
  std::printf("u16_stride3,%zu,%zu\n", bytes / 3, bytes);
  hipLaunchKernelGGL((read_kernel<double, 1, 1>), dim3(blocks), dim3(threads), 0, 0, (const double *)buf, bytes / 8, sink);
  std::printf("f64_unit,%zu,%zu\n", bytes, bytes);
  hipLaunchKernelGGL((read_kernel<double, 3, 1>), dim3(blocks), dim3(threads), 0, 0, (const double *)buf, bytes / 8, sink);
  std::printf("f64_stride3,%zu,%zu\n", bytes / 3, bytes);
  hipLaunchKernelGGL((read_kernel<double, 1, 16>), dim3(blocks), dim3(threads), 0, 0, (const double *)buf, bytes / 8, sink);
  std::printf("f64_sparse16,%zu,%zu\n", bytes / 16, bytes);
  hipLaunchKernelGGL((read_kernel<X4, 1, 1>), dim3(blocks), dim3(threads), 0, 0, (const X4 *)buf, bytes / 16, sink);
  std::printf("x4_unit,%zu,%zu\n", bytes, bytes);
  if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 1; }
  hipFree(buf); hipFree(sink);
  return 0;
}
