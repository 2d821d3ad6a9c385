// Round 6 probe: where does the dispatcher put the workgroups of a two-per-CU launch?  A kernel with the layer GEMM's launch shape (512
// threads, 80 KB of dynamic LDS, so two workgroups share a CU) records per workgroup its XCC, SE / SH / CU ids and start time and then
// spins long enough that the whole first round is resident together.  Output: for every (XCC, SE, SH, CU) the block indices it held, in
// start order - the pairing rule `cu_pair_local` (common.h) assumes local indices j and j + 32 of an XCD share a CU.
//   hipcc --offload-arch=gfx950 -O2 tools/placement_probe.cpp -o tools/_bin/placement_probe && tools/_bin/placement_probe [grid]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <map>
#include <vector>
#include <algorithm>
#include <tuple>

__global__ void __launch_bounds__(512) probe(uint32_t* out, int spin) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint64_t t = __builtin_readcyclecounter();
    out[blockIdx.x * 4 + 0] = hw; out[blockIdx.x * 4 + 1] = xcc; out[blockIdx.x * 4 + 2] = (uint32_t)t; out[blockIdx.x * 4 + 3] = (uint32_t)(t >> 32);
    smem[0] = 1;
  }
  const uint64_t t0 = __builtin_readcyclecounter();
  while ((int64_t)(__builtin_readcyclecounter() - t0) < spin) { __builtin_amdgcn_s_sleep(8); }
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 512;
  uint32_t* d; hipMalloc(&d, grid * 16); hipMemset(d, 0, grid * 16);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 81920, 0, d, 2000000); hipDeviceSynchronize(); }
  std::vector<uint32_t> h(grid * 4); hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
  std::map<std::tuple<int, int, int, int>, std::vector<std::pair<uint64_t, int>>> cus;
  for (int b = 0; b < grid; ++b) {
    const uint32_t hw = h[b * 4], xcc = h[b * 4 + 1] & 15;
    const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cus[{(int)xcc, se, sh, cu}].push_back({((uint64_t)h[b * 4 + 3] << 32) | h[b * 4 + 2], b});
  }
  printf("grid %d: %zu distinct (xcc, se, sh, cu)\n", grid, cus.size());
  int same_xcc_as_mod8 = 0;
  for (int b = 0; b < grid; ++b) same_xcc_as_mod8 += ((h[b * 4 + 1] & 15) == (uint32_t)(b & 7));
  printf("blocks whose XCC id equals blockIdx %% 8: %d of %d\n", same_xcc_as_mod8, grid);
  int shown = 0, pair32 = 0, pairs = 0;
  for (auto& kv : cus) {
    auto v = kv.second; std::sort(v.begin(), v.end());
    if (v.size() >= 2) { ++pairs; const int j0 = v[0].second >> 3, j1 = v[1].second >> 3; if (j1 - j0 == 32 || j0 - j1 == 32) ++pair32; }
    if (shown < 40) {
      printf("xcc %d se %d sh %d cu %2d:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first));
      for (auto& p : v) printf("  b %4d (local %3d)", p.second, p.second >> 3);
      printf("\n"); ++shown;
    }
  }
  printf("CUs holding >= 2 workgroups: %d, of which the first two are local j and j + 32: %d\n", pairs, pair32);
  return 0;
}
