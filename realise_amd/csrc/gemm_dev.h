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
__device__ __forceinline__ int epi_row(const EpiParams<T>& ep, int row) {
  if (ep.rm_hw_shift < 0) return row;
  const int hs = ep.rm_hw_shift - 2, ws = ep.rm_w_shift - 1;
  const int n = row >> hs, rem = row & ((1 << hs) - 1);
  const int y = ((rem >> ws) << 1) + (ep.rm_par >> 1), x = ((rem & ((1 << ws) - 1)) << 1) + (ep.rm_par & 1);
  return (n << ep.rm_hw_shift) + (y << ep.rm_w_shift) + x;
}

template <typename T>
__device__ __forceinline__ void epilogue4(const EpiParams<T>& ep, int M, int N, int row, int col, floatx4 v) {
  if (row >= M || col >= N) return;
  row = epi_row<T>(ep, row);
  if (ep.alpha != 1.0f) v *= ep.alpha;
  if (ep.col_scale != nullptr) v *= *(const floatx4*)(ep.col_scale + col);
  if (ep.bias != nullptr) v += *(const floatx4*)(ep.bias + col);
  switch (ep.mode) {
    case EPI_STORE: {
      T* o = ep.out + (int64_t)row * ep.ldo + col;
      if (ep.accumulate) v += load4<T>(o);
      store4<T>(o, v);
    } break;
    case EPI_AFFINE: {
      if (ep.aux != nullptr) v += load4<T>(ep.aux + (int64_t)row * ep.ldaux + col);
      if (ep.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      store4<T>(ep.out + (int64_t)row * ep.ldo + col, v);
    } break;
    case EPI_GELU: {
      if (ep.out2 != nullptr) store4<T>(ep.out2 + (int64_t)row * ep.ldo + col, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = gelu_fwd<T>(v[j]);
      store4<T>(ep.out + (int64_t)row * ep.ldo + col, v);
    } break;
    case EPI_DROP_RESID: {
      const uint32_t idx = (uint32_t)row * (uint32_t)N + (uint32_t)col;
      if ((N & 3) == 0) v *= drop_mult4(ep.drop_seed, ep.drop_thresh, ep.drop_scale, idx);          // col % 4 == 0: one hash for the quad
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= drop_mult(ep.drop_seed, ep.drop_thresh, ep.drop_scale, idx + j);
      }
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
  row = epi_row<T>(ep, row);
  if (ep.alpha != 1.0f) { a *= ep.alpha; b *= ep.alpha; }
  if (ep.col_scale != nullptr) { a *= *(const floatx4*)(ep.col_scale + col); b *= *(const floatx4*)(ep.col_scale + col + 4); }
  if (ep.bias != nullptr) { a += *(const floatx4*)(ep.bias + col); b += *(const floatx4*)(ep.bias + col + 4); }
  T* o = ep.out + (int64_t)row * ep.ldo + col;
  switch (ep.mode) {
    case EPI_STORE: {
      if (ep.accumulate) { floatx4 pa, pb; load8<T>(o, pa, pb); a += pa; b += pb; }
      store8<T>(o, a, b);
    } break;
    case EPI_AFFINE: {
      if (ep.aux != nullptr) { floatx4 ra, rb; load8<T>(ep.aux + (int64_t)row * ep.ldaux + col, ra, rb); a += ra; b += rb; }
      if (ep.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = fmaxf(a[j], 0.f); b[j] = fmaxf(b[j], 0.f); }
      }
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
      a *= drop_mult4(ep.drop_seed, ep.drop_thresh, ep.drop_scale, idx);          // N % 8 == 0, col % 8 == 0
      b *= drop_mult4(ep.drop_seed, ep.drop_thresh, ep.drop_scale, idx + 4);
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

// epilogue8 with the row's aux / old-output octet already in registers (`pre`: the residual of EPI_DROP_RESID, the pre-activation of
// EPI_GELU_BWD, or - `pre_is_old` - the output an accumulating EPI_STORE adds to).  The 8-wave kernels request these octets for ALL
// of a wave's items before the first store (gemm_nt8.hip): loaded item by item they sat behind their predecessors' stores, and hipcc
// waits with vmcnt(0) whenever loads and stores are pending together.  Same arithmetic, same order: same bits as epilogue8.
__device__ __forceinline__ void epilogue8_pre(const EpiParams<bf16_t>& ep, int M, int N, int row, int col, floatx4 a, floatx4 b, uint4 pre) {
  typedef bf16_t T;
  if (row >= M || col >= N) return;
  if (ep.alpha != 1.0f) { a *= ep.alpha; b *= ep.alpha; }
  if (ep.bias != nullptr) { a += *(const floatx4*)(ep.bias + col); b += *(const floatx4*)(ep.bias + col + 4); }
  T* o = ep.out + (int64_t)row * ep.ldo + col;
  floatx4 pa, pb;
  unpack8(pre, pa, pb);
  switch (ep.mode) {
    case EPI_STORE: {
      if (ep.accumulate) { a += pa; b += pb; }
      store8<T>(o, a, b);
    } break;
    case EPI_DROP_RESID: {
      const uint32_t idx = (uint32_t)row * (uint32_t)N + (uint32_t)col;
      a *= drop_mult4(ep.drop_seed, ep.drop_thresh, ep.drop_scale, idx);          // N % 8 == 0, col % 8 == 0
      b *= drop_mult4(ep.drop_seed, ep.drop_thresh, ep.drop_scale, idx + 4);
      store8<T>(o, a + pa, b + pb);
    } break;
    case EPI_GELU_BWD: {
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] *= gelu_bwd<T>(pa[j]); b[j] *= gelu_bwd<T>(pb[j]); }
      store8<T>(o, a, b);
    } break;
    default: break;
  }
}

// ---- TN (weight-gradient) device helpers shared by gemm.hip and gemm_tn8.hip ---------------------------------------------
template <typename T> struct TnGeo;
template <> struct TnGeo<bf16_t> { static constexpr int BP = 64, KSTEPS = 2, VEC = 8; };
template <> struct TnGeo<float> { static constexpr int BP = 32, KSTEPS = 8, VEC = 4; };

// Byte XOR applied to the column offset of reduction row p (keeps 16-byte chunks intact).  bf16: one
// ds_read_b64_tr_b16 half-wave touches rows {p0..p0+3} and {p0+8..p0+11} at the same 32-byte column block;
// the XOR spreads those 8 rows over 8 distinct 32-byte slots of the 256-byte bank row (4 slots when the
// tile row is only 128 bytes).  fp32: lanes 0-31 read rows p, p+1 -> two 64-byte halves.
template <typename T, int RP> __device__ __forceinline__ int tn_swz(int p) {
  if constexpr (sizeof(T) == 2) return (((p & 3) | (((p >> 3) & 1) << 2)) << 5) & (RP - 1);
  else return ((p & 1) << 6) & (RP - 1);
}

template <bool TR, int RP>
__device__ __forceinline__ bf16x8_t tn_frag_bf16(const char* tile, int ks, int col0, int l15, int g) {
  if constexpr (TR) {
    // 16-lane group g reads the [4 p][16 col] blocks at rows 8g+4h .. +3; lane q of the group points at
    // row (q >> 2), columns 4*(q & 3) .. +3 and receives column q, 4 consecutive p.
    typedef short4_t __attribute__((address_space(3))) * lds_s4;
    short4_t h[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int p = ks * 32 + 8 * g + 4 * hh + (l15 >> 2);
      const int colb = ((col0 + 4 * (l15 & 3)) * 2) ^ tn_swz<bf16_t, RP>(p);
      h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(tile + p * RP + colb));
    }
    typedef __attribute__((ext_vector_type(8))) short short8_t;
    short8_t r = {h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
    return __builtin_bit_cast(bf16x8_t, r);
  } else {
    typedef __attribute__((ext_vector_type(8))) unsigned short ushort8_t;
    ushort8_t r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = ks * 32 + 8 * g + e;
      r[e] = *(const unsigned short*)(tile + p * RP + (((col0 + l15) * 2) ^ tn_swz<bf16_t, RP>(p)));
    }
    return __builtin_bit_cast(bf16x8_t, r);
  }
}

enum TnOut { TN_OUT_DIRECT = 0, TN_OUT_SLAB = 1, TN_OUT_ATOMIC = 2 };

__device__ __forceinline__ int64_t tn_out_index(const TnEpi& ep, int i, int j) {   // -1: padding column
  if (ep.mode == TN_PLAIN) return (int64_t)i * ep.ldo + j;
  const int tap = j / ep.Cpad, ci = j - tap * ep.Cpad;
  return ci < ep.Cin ? ((int64_t)i * ep.Cin + ci) * ep.KHW + tap + ep.tap0 : -1;
}

__device__ __forceinline__ float tn_alpha(const TnEpi& ep) { return ep.alpha_dev != nullptr ? ep.alpha * *ep.alpha_dev : ep.alpha; }

__device__ __forceinline__ void tn_epilogue4(const TnEpi& ep, int how, int split, int I, int J, int i, int j, floatx4 v) {
  if (i >= I || j >= J) return;
  if (how == TN_OUT_SLAB) {                       // dense [split][I][J]; J % 4 == 0
    *(floatx4*)(ep.slab + ((int64_t)split * I + i) * J + j) = v;
    return;
  }
  v *= tn_alpha(ep);
  if (how == TN_OUT_DIRECT && ep.mode == TN_PLAIN && (ep.ldo & 3) == 0) {
    floatx4* o = (floatx4*)(ep.out + (int64_t)i * ep.ldo + j);
    if (ep.overwrite) *o = v; else *o = *o + v;
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t idx = tn_out_index(ep, i, j + r);
    if (idx < 0) continue;
    if (how == TN_OUT_DIRECT) { if (ep.overwrite) ep.out[idx] = v[r]; else ep.out[idx] += v[r]; }
    else atomicAdd(ep.out + idx, v[r]);
  }
}


}  // namespace rl
