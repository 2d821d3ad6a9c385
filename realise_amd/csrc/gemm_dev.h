// Device-side pieces shared by the GEMM translation units (gemm.hip, gemm_nt8.hip): tile geometry, the direct-to-LDS load,
// and the fused epilogues of the NT kernels.  Every translation unit gets its own copy of the zero page (-fno-gpu-rdc).
#pragma once
#include "gemm.h"

namespace rl {

template <typename T> struct Geo;
template <> struct Geo<bf16_t> { static constexpr int BK = 64, VEC = 8, KSTEPS = 2; };
template <> struct Geo<float> { static constexpr int BK = 32, VEC = 4, KSTEPS = 8; };

static __device__ uint4 g_zero16[4];

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T>
__device__ __forceinline__ void epilogue4(const EpiParams<T>& ep, int M, int N, int row, int col, floatx4 v) {
  if (row >= M || col >= N) return;
  if (ep.alpha != 1.0f) v *= ep.alpha;
  if (ep.bias != nullptr) v += *(const floatx4*)(ep.bias + col);
  switch (ep.mode) {
    case EPI_STORE: {
      T* o = ep.out + (int64_t)row * ep.ldo + col;
      if (ep.accumulate) v += load4<T>(o);
      store4<T>(o, v);
    } break;
    case EPI_GELU: {
      if (ep.out2 != nullptr) store4<T>(ep.out2 + (int64_t)row * ep.ldo + col, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = gelu_fwd<T>(v[j]);
      store4<T>(ep.out + (int64_t)row * ep.ldo + col, v);
    } break;
    case EPI_DROP_RESID: {
      const uint32_t idx = (uint32_t)row * (uint32_t)N + (uint32_t)col;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= drop_mult(ep.drop_seed, ep.drop_thresh, ep.drop_scale, idx + j);
      v += load4<T>(ep.aux + (int64_t)row * ep.ldaux + col);
      store4<T>(ep.out + (int64_t)row * ep.ldo + col, v);
    } break;
    case EPI_GELU_BWD: {
      const floatx4 x = load4<T>(ep.aux + (int64_t)row * ep.ldaux + col);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= gelu_bwd<T>(x[j]);
      T* o = ep.out + (int64_t)row * ep.ldo + col;
      if (ep.accumulate) v += load4<T>(o);
      store4<T>(o, v);
    } break;
    default: break;
  }
}

// Same epilogues over 8 consecutive columns of one row (col % 8 == 0, N % 8 == 0): every global access is one
// dwordx4 per lane (bf16) and a wave-instruction covers whole 128-byte row segments.
template <typename T>
__device__ __forceinline__ void epilogue8(const EpiParams<T>& ep, int M, int N, int row, int col, floatx4 a, floatx4 b) {
  if (row >= M || col >= N) return;
  if (ep.alpha != 1.0f) { a *= ep.alpha; b *= ep.alpha; }
  if (ep.bias != nullptr) { a += *(const floatx4*)(ep.bias + col); b += *(const floatx4*)(ep.bias + col + 4); }
  T* o = ep.out + (int64_t)row * ep.ldo + col;
  switch (ep.mode) {
    case EPI_STORE: {
      if (ep.accumulate) { floatx4 pa, pb; load8<T>(o, pa, pb); a += pa; b += pb; }
      store8<T>(o, a, b);
    } break;
    case EPI_GELU: {
      if (ep.out2 != nullptr) store8<T>(ep.out2 + (int64_t)row * ep.ldo + col, a, b);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = gelu_fwd<T>(a[j]); b[j] = gelu_fwd<T>(b[j]); }
      store8<T>(o, a, b);
    } break;
    case EPI_DROP_RESID: {
      const uint32_t idx = (uint32_t)row * (uint32_t)N + (uint32_t)col;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] *= drop_mult(ep.drop_seed, ep.drop_thresh, ep.drop_scale, idx + j);
        b[j] *= drop_mult(ep.drop_seed, ep.drop_thresh, ep.drop_scale, idx + 4 + j);
      }
      floatx4 ra, rb;
      load8<T>(ep.aux + (int64_t)row * ep.ldaux + col, ra, rb);
      store8<T>(o, a + ra, b + rb);
    } break;
    case EPI_GELU_BWD: {
      floatx4 xa, xb;
      load8<T>(ep.aux + (int64_t)row * ep.ldaux + col, xa, xb);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] *= gelu_bwd<T>(xa[j]); b[j] *= gelu_bwd<T>(xb[j]); }
      if (ep.accumulate) { floatx4 pa, pb; load8<T>(o, pa, pb); a += pa; b += pb; }
      store8<T>(o, a, b);
    } break;
    default: break;
  }
}


}  // namespace rl
