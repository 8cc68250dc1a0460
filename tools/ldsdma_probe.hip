// Probe: does `buffer_load_dwordx4 ... lds` (gfx950) put lane L's 16 bytes at LDS[M0 base + 16 L]?   hipcc --offload-arch=gfx950 -O3 tools/ldsdma_probe.hip -o variants/ldsdma_probe (variants/ is git-ignored)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) char smem[];
__global__ void k(const char* src, char* dst, int n) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, n, 0x00020000);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // lane L fetches source vector L ^ 1 of the wave's KB; out-of-range lanes (wave 3, lanes >= 32) must read zeros
  const unsigned off = (wave == 3 && lane >= 32) ? 0xFFFFFFF0u : (unsigned)((lane ^ 1) * 16 + wave * 1024);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = smem[i];
}
int main() {
  const int n = 4096;
  std::vector<unsigned char> h(n), o(n);
  for (int i = 0; i < n; ++i) h[i] = (unsigned char)(i * 7 + (i >> 8));
  char *s, *d;
  hipMalloc(&s, n), hipMalloc(&d, n);
  hipMemcpy(s, h.data(), n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, s, d, n);
  hipMemcpy(o.data(), d, n, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const int v = i / 16, wave = v / 64, lane = v % 64;
    const unsigned char want = (wave == 3 && lane >= 32) ? 0 : h[(wave * 64 + (lane ^ 1)) * 16 + i % 16];
    bad += o[i] != want;
  }
  printf("lds dma dwordx4: %s (%d mismatching bytes)\n", bad ? "UNEXPECTED" : "as expected", bad);
  return bad != 0;
}
