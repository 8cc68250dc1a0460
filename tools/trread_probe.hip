// Probe: gfx950 `ds_read_b64_tr_b16`.  Expectation (cdna_hip_programming.md T10): within a 16-lane group, lane i supplies the
// address of 4 consecutive 16-bit elements M[i / 4][4 (i % 4) .. +3] of a 4 x 16 matrix and receives COLUMN i: M[0..3][i].
// For a pixel-major tile [pixel][channel] that is "4 consecutive pixels of one channel" = a K-contiguous MFMA operand piece
// for contractions over pixels (wgrad).   hipcc --offload-arch=gfx950 -O3 tools/trread_probe.hip -o variants/trread_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64 * 64];     // [pixel][channel], value = pixel * 64 + channel
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) tile[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x & 63, g = l >> 4, i = l & 15;
  const int p0 = 8, c0 = 16;
  // lane supplies &tile[p0 + 4 g + i / 4][c0 + 4 (i % 4)]
  const unsigned addr = (unsigned)(size_t)(&tile[(p0 + 4 * g + i / 4) * 64 + c0 + 4 * (i % 4)]);
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = (unsigned short)(v >> (16 * r));
}
int main() {
  unsigned short* d;
  (void)hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  std::vector<unsigned short> h(256);
  (void)hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int g = l >> 4, i = l & 15;
      const int want = (8 + 4 * g + r) * 64 + 16 + i;                      // pixel p0 + 4 g + r, channel c0 + i
      bad += h[l * 4 + r] != want;
    }
  printf("ds_read_b64_tr_b16: %s (%d mismatches); lane 0: %d %d %d %d, lane 17: %d %d %d %d\n", bad ? "UNEXPECTED" : "as expected", bad,
         h[0], h[1], h[2], h[3], h[68], h[69], h[70], h[71]);
  return bad != 0;
}
