// Effective shader clock under three loads on every CU: (1) MFMA only, (2) LDS-DMA fill only (L2-resident source), (3) both from
// different waves of one workgroup.  s_memtime counts shader cycles, wall_clock64() a constant 100 MHz: their ratio is the clock.
//   hipcc --offload-arch=gfx950 -O2 tools/clock_probe.cpp -o tools/_bin/clock_probe && tools/_bin/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f4;

__global__ void __launch_bounds__(512) k(const char* src, int mode, int iters, uint64_t* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool do_mfma = (mode & 1) && wave < 4 + 4 * ((mode >> 2) & 1);     // mode bit 2: all 8 waves multiply
  const bool do_fill = (mode & 2) && (wave >= 4 || ((mode >> 3) & 1));      // mode bit 3: all 8 waves fetch
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  const uint32_t voff = (uint32_t)(lane * 16 + wave * 1024 + (blockIdx.x % 64) * 1048576);
  __syncthreads();
  const uint64_t w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  int done = 0;
  for (int it = 0; wall_clock64() - w0 < (uint64_t)iters; ++it, ++done) {      // `iters` = duration in 10 ns ticks: every wave runs for the same time
    if (do_mfma) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    if (do_fill) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 4096 + u * 1024), 16, voff, (it & 31) * 32768 + u * 8192, 0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint64_t c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (s == 123.456f) sink[0] = s + smem[lane];
  if (lane == 0) {
    if (do_mfma) atomicAdd((unsigned long long*)&out[blockIdx.x * 4 + 2], (unsigned long long)done);
    if (do_fill) atomicAdd((unsigned long long*)&out[blockIdx.x * 4 + 3], (unsigned long long)done);
  }
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = c1 - c0; out[blockIdx.x * 4 + 1] = w1 - w0; }
}

int main() {
  char* src; hipMalloc(&src, 96 << 20); hipMemset(src, 1, 96 << 20);
  uint64_t* out; hipMalloc(&out, 256 * 32);
  float* sink; hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 300000;      // 3 ms
  const char* names[] = {"", "MFMA only (4 waves)", "fill only (4 waves)", "MFMA (waves 0-3) + fill (waves 4-7)", "", "MFMA only (8 waves)", "", "MFMA (8 waves) + fill (waves 4-7)",
                         "", "", "fill only (8 waves)", "MFMA (waves 0-3) + fill (8 waves)", "", "", "", "MFMA (8 waves) + fill (8 waves)"};
  for (int mode : {1, 2, 3, 5, 10, 7, 15}) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, src, mode, iters, out, sink);
    hipDeviceSynchronize();
    hipMemset(out, 0, 256 * 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, src, mode, iters, out, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(1024); hipMemcpy(h.data(), out, 256 * 32, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0, im = 0, ifl = 0; for (int i = 0; i < 256; ++i) { cyc += h[4 * i]; wall += h[4 * i + 1]; im += h[4 * i + 2]; ifl += h[4 * i + 3]; }
    cyc /= 256; wall /= 256;
    const double flops = im * 32.0 * 16 * 16 * 32 * 2, bytes = ifl * 4.0 * 1024, sec = wall * 1e-8;
    printf("%-40s %8.2f ms  clock %.2f GHz  %6.0f TF  %6.2f TB/s fill (%.1f B/clk/CU)\n", names[mode], ms, cyc / (wall * 10.0), flops / sec * 1e-12, bytes / sec * 1e-12,
           bytes / 256 / cyc);
  }
  return 0;
}
