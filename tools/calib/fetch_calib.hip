// FETCH_SIZE / WRITE_SIZE calibration: known byte counts, far past the 256 MiB Infinity Cache, in the access patterns of the
// split-precision conv (csrc/conv.hip) -- run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (tools/calib/run_calib.sh).
//   read_wide            16 B per lane, a wave instruction reads 1 KiB contiguous (the guide's reference pattern: reported as 1/2)
//   read_rows64_all      the 128 x 128-tile 3x3 kernel's staging: a "pixel" is a 512 B row (128 fp32 channels); one instruction reads 64 B (16 channels:
//                        4 lanes x 16 B) of 16 pixels; the 8 chunks of a row are read one after the other by the same thread (every byte read once)
//   read_rows64_first    only the first 64 B of every 512 B row (1/8 of the bytes): request granularity of a lone 64 B access
//   read_rows128_first   only the first 128 B of every row
//   read_rows256_all     16 lanes x 16 B = 256 B per pixel (the 1x1 / 64-channel stagings), both halves of a row
//   write_wide           16 B per lane, 1 KiB contiguous per wave instruction
//   write_rows256        the fp32 epilogue's stores: 16 lanes x 16 B = 256 B of a pixel row, 4 rows per instruction, second half of the row by another wave
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void read_wide(const uint4* __restrict__ b, long n16, unsigned* out) {
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) { const uint4 v = b[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int LANES_PER_ROW, int CHUNKS>      // LANES_PER_ROW x 16 B contiguous per row per instruction; CHUNKS consecutive such pieces of the row read one after the other
__global__ void read_rows(const uint4* __restrict__ b, long rows, unsigned* out) {
  unsigned acc = 0;
  const long nthr = (long)gridDim.x * blockDim.x;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < rows * LANES_PER_ROW; g += nthr) {
    const long row = g / LANES_PER_ROW; const int sub = (int)(g % LANES_PER_ROW);
#pragma unroll 1
    for (int c = 0; c < CHUNKS; ++c) {
      const uint4 v = b[row * 32 + c * LANES_PER_ROW + sub];      // 512 B rows = 32 x 16 B
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
      __builtin_amdgcn_s_sleep(8);                                // the chunks of a row are separate requests in time, as in the K loop
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void write_wide(uint4* __restrict__ b, long n16) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) b[i] = make_uint4((unsigned)i, 1, 2, 3);
}
__global__ void write_rows256(uint4* __restrict__ b, long rows) {      // thread -> (row, 16-lane group within half h); half 0 of all rows first, half 1 by other blocks
  const long nthr = (long)gridDim.x * blockDim.x;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < rows * 32; g += nthr) {
    const long half = g / (rows * 16), r = (g % (rows * 16)) / 16; const int sub = (int)(g % 16);
    b[r * 32 + half * 16 + sub] = make_uint4((unsigned)g, 1, 2, 3);
  }
}

int main() {
  const long bytes = 1L << 30;                 // 1 GiB: 4 x the Infinity Cache
  const long rows = bytes / 512, n16 = bytes / 16;
  uint4* buf; unsigned* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 1, bytes)); CK(hipMemset(out, 0, 64));
  const int G = 256 * 8, T = 256;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read_wide, dim3(G), dim3(T), 0, 0, buf, n16, out);
    hipLaunchKernelGGL((read_rows<4, 8>), dim3(G), dim3(T), 0, 0, buf, rows, out);
    hipLaunchKernelGGL((read_rows<4, 1>), dim3(G), dim3(T), 0, 0, buf, rows, out);
    hipLaunchKernelGGL((read_rows<8, 1>), dim3(G), dim3(T), 0, 0, buf, rows, out);
    hipLaunchKernelGGL((read_rows<16, 2>), dim3(G), dim3(T), 0, 0, buf, rows, out);
    hipLaunchKernelGGL(write_wide, dim3(G), dim3(T), 0, 0, buf, n16);
    hipLaunchKernelGGL(write_rows256, dim3(G), dim3(T), 0, 0, buf, rows);
    CK(hipDeviceSynchronize());
  }
  printf("calib done: buffer %ld bytes; expected bytes: read_wide %ld, read_rows<4,8> %ld, read_rows<4,1> %ld, read_rows<8,1> %ld, read_rows<16,2> %ld, write_wide %ld, write_rows256 %ld\n",
         bytes, bytes, bytes, bytes / 8, bytes / 4, bytes, bytes, bytes);
  return 0;
}
