// Shared device helpers for the ReaLiSe MI355X (gfx950 / CDNA4) kernels.
// Wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rl {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // one MFMA 16x16x32 operand
typedef __attribute__((ext_vector_type(4))) float floatx4;      // one MFMA 16x16 accumulator
typedef __attribute__((ext_vector_type(4))) short short4_t;

struct bf16_t { uint16_t v; };                                    // storage-only bf16

__device__ __forceinline__ float bf2f(uint16_t x) { return __uint_as_float(((uint32_t)x) << 16); }
// fp32 -> bf16, round-to-nearest-even, on the hardware converter (gfx950: v_cvt_pk_bf16_f32, two values per instruction)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw_t;
typedef __attribute__((ext_vector_type(2))) float floatx2_hw_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const floatx2_hw_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw_t));
}
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)(pack2bf(f, 0.0f) & 0xffffu); }
// exp for the softmax / cross-entropy kernels: full-precision expf in parity mode (T = float), the 2-instruction
// v_exp_f32 form in speed mode (relative error ~1e-6, far inside bf16 resolution)
template <typename T> __device__ __forceinline__ float exp_t(float x) {
  if constexpr (sizeof(T) == 4) return expf(x); else return __expf(x);
}

template <typename T> __device__ __forceinline__ float to_f(T x);
template <> __device__ __forceinline__ float to_f<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t x) { return bf2f(x.v); }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { bf16_t r; r.v = f2bf(x); return r; }

// ---- 4-wide load / store of T (8 B for bf16, 16 B for f32) ---------------------------------
template <typename T> __device__ __forceinline__ floatx4 load4(const T* p);
template <> __device__ __forceinline__ floatx4 load4<float>(const float* p) { return *(const floatx4*)p; }
template <> __device__ __forceinline__ floatx4 load4<bf16_t>(const bf16_t* p) {
  uint2 u = *(const uint2*)p;
  floatx4 r;
  r[0] = __uint_as_float(u.x << 16); r[1] = __uint_as_float(u.x & 0xffff0000u);
  r[2] = __uint_as_float(u.y << 16); r[3] = __uint_as_float(u.y & 0xffff0000u);
  return r;
}
template <typename T> __device__ __forceinline__ void store4(T* p, floatx4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, floatx4 v) { *(floatx4*)p = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, floatx4 v) {
  uint2 u;
  u.x = pack2bf(v[0], v[1]);
  u.y = pack2bf(v[2], v[3]);
  *(uint2*)p = u;
}

// ---- 8-wide load / store (16 B for bf16: one dwordx4 per lane; 32 B for f32) ---------------
template <typename T> __device__ __forceinline__ void load8(const T* p, floatx4& a, floatx4& b);
template <> __device__ __forceinline__ void load8<float>(const float* p, floatx4& a, floatx4& b) { a = *(const floatx4*)p; b = *(const floatx4*)(p + 4); }
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, floatx4& a, floatx4& b) {
  const uint4 u = *(const uint4*)p;
  a[0] = __uint_as_float(u.x << 16); a[1] = __uint_as_float(u.x & 0xffff0000u);
  a[2] = __uint_as_float(u.y << 16); a[3] = __uint_as_float(u.y & 0xffff0000u);
  b[0] = __uint_as_float(u.z << 16); b[1] = __uint_as_float(u.z & 0xffff0000u);
  b[2] = __uint_as_float(u.w << 16); b[3] = __uint_as_float(u.w & 0xffff0000u);
}
// eight bf16 already in registers (one dwordx4) -> two floatx4
__device__ __forceinline__ void unpack8(const uint4 u, floatx4& a, floatx4& b) {
  a[0] = __uint_as_float(u.x << 16); a[1] = __uint_as_float(u.x & 0xffff0000u);
  a[2] = __uint_as_float(u.y << 16); a[3] = __uint_as_float(u.y & 0xffff0000u);
  b[0] = __uint_as_float(u.z << 16); b[1] = __uint_as_float(u.z & 0xffff0000u);
  b[2] = __uint_as_float(u.w << 16); b[3] = __uint_as_float(u.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store8(T* p, floatx4 a, floatx4 b);
template <> __device__ __forceinline__ void store8<float>(float* p, floatx4 a, floatx4 b) { *(floatx4*)p = a; *(floatx4*)(p + 4) = b; }
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, floatx4 a, floatx4 b) {
  uint4 u;
  u.x = pack2bf(a[0], a[1]);
  u.y = pack2bf(a[2], a[3]);
  u.z = pack2bf(b[0], b[1]);
  u.w = pack2bf(b[2], b[3]);
  *(uint4*)p = u;
}

// ---- MMA atoms -----------------------------------------------------------------------------
// Both atoms compute a 16x16 fp32 tile.  Operand register layout is symmetric for the
// two inputs: lane l supplies, for row/col index (l & 15), the K-slots owned by group
// g = l >> 4.  Result layout (dtype independent on gfx950):
//   mma(x, y, c): lane l holds c[r] = D[i = 4*g + r][j = l & 15], i indexes x rows, j indexes y.
struct MmaBF16 {
  typedef bf16_t T;
  static constexpr int K = 32;                 // K per instruction
  typedef bf16x8_t Frag;
  static __device__ __forceinline__ floatx4 mma(Frag x, Frag y, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
  }
};
struct MmaF32 {                                // exact fp32: bitwise a k-ordered fmaf chain
  typedef float T;
  static constexpr int K = 4;
  typedef float Frag;
  static __device__ __forceinline__ floatx4 mma(Frag x, Frag y, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c, 0, 0, 0);
  }
};
template <typename T> struct MmaOf;
template <> struct MmaOf<bf16_t> { typedef MmaBF16 type; };
template <> struct MmaOf<float> { typedef MmaF32 type; };

// ---- LDS tile with K contiguous rows of KELEMS elements, 16-byte chunks XOR-swizzled by row --
// chunk c of row r lives at chunk (c ^ (r & 7)): conflict-free ds_read_b128 for MFMA operand
// reads (rows vary across lanes, same logical chunk) and for the 8-lanes-per-row staging writes.
template <typename T, int KELEMS> struct KTile {
  static constexpr int ROWB = KELEMS * (int)sizeof(T);
  static constexpr int CH = ROWB / 16;
  static_assert(CH >= 8 && (CH & (CH - 1)) == 0, "row must hold a power-of-two >= 8 chunks");
  static __device__ __forceinline__ int off(int row, int chunk) { return row * ROWB + ((chunk ^ (row & 7)) << 4); }
  static constexpr int bytes(int rows) { return rows * ROWB; }
};

// operand fragment for MMA k-step `ks` (ks counts Mma::K-sized steps inside the tile row)
template <typename T, int KELEMS>
__device__ __forceinline__ typename MmaOf<T>::type::Frag ktile_frag(const char* tile, int row, int ks, int g);
template <> __device__ __forceinline__ bf16x8_t ktile_frag<bf16_t, 64>(const char* tile, int row, int ks, int g) {
  return *(const bf16x8_t*)(tile + KTile<bf16_t, 64>::off(row, ks * 4 + g));
}
template <> __device__ __forceinline__ float ktile_frag<float, 32>(const char* tile, int row, int ks, int g) {
  return *(const float*)(tile + KTile<float, 32>::off(row, ks) + 4 * g);
}
template <> __device__ __forceinline__ float ktile_frag<float, 64>(const char* tile, int row, int ks, int g) {
  return *(const float*)(tile + KTile<float, 64>::off(row, ks) + 4 * g);
}

// ---- counter-based dropout RNG (K17): mask is recomputed in backward, never stored ----------
// One hash serves FOUR neighbouring elements (16 bits each).  Integer multiplies are quarter-rate on CDNA, so the cost that matters
// in the epilogues / softmax that call this per value is multiplies per element: three 32 x 32 -> 64-bit products per quad (0.75 per
// element; the two-round 32-bit mixer this replaces spent 1.5), both halves of every product used.  Chi-square of the 16-bit
// lanes, lag / seed-to-seed correlations of the keep mask checked on 4M counters (tests/test_round3_cpu.py restates the mixer).
__device__ __forceinline__ uint2 rng_hash4(uint32_t seed, uint32_t quad) {
  const uint64_t m1 = (uint64_t)(quad ^ seed) * 0x9E3779B1u;
  const uint32_t x = (uint32_t)m1 ^ (uint32_t)(m1 >> 32) ^ (seed * 0x632BE5ABu);
  const uint64_t m2 = (uint64_t)x * 0x85EBCA77u, m3 = (uint64_t)(x ^ 0x27D4EB2Fu) * 0xC2B2AE3Du;
  return uint2{(uint32_t)(m2 >> 32) ^ (uint32_t)m3, (uint32_t)(m3 >> 32) ^ (uint32_t)m2};
}
// the 16 random bits of element `sub` (0..3) of a quad
__device__ __forceinline__ uint32_t rng_lane16(uint2 h, uint32_t sub) { return (((sub & 2u) ? h.y : h.x) >> ((sub & 1u) << 4)) & 0xffffu; }
// keep element idx?  thresh = p * 2^32 (0 disables dropout); the comparison uses the top 16 bits of thresh, i.e. the drop
// probability is p rounded down to a multiple of 2^-16.
__device__ __forceinline__ float drop_mult(uint32_t seed, uint32_t thresh, float scale, uint32_t idx) {
  if (thresh == 0u) return 1.0f;
  return rng_lane16(rng_hash4(seed, idx >> 2), idx & 3u) >= (thresh >> 16) ? scale : 0.0f;
}
// the four multipliers of elements idx .. idx + 3, idx % 4 == 0
__device__ __forceinline__ floatx4 drop_mult4(uint32_t seed, uint32_t thresh, float scale, uint32_t idx) {
  if (thresh == 0u) return floatx4{1.0f, 1.0f, 1.0f, 1.0f};
  const uint2 h = rng_hash4(seed, idx >> 2);
  const uint32_t t = thresh >> 16;
  return floatx4{(h.x & 0xffffu) >= t ? scale : 0.0f, (h.x >> 16) >= t ? scale : 0.0f, (h.y & 0xffffu) >= t ? scale : 0.0f, (h.y >> 16) >= t ? scale : 0.0f};
}

// the same without the thresh == 0 test (callers that are only instantiated for a live dropout site: the test is a uniform BRANCH
// per call that splits a straight-line row loop into blocks the scheduler cannot move loads across)
__device__ __forceinline__ floatx4 drop_mult4_nz(uint32_t seed, uint32_t thresh, float scale, uint32_t idx) {
  const uint2 h = rng_hash4(seed, idx >> 2);
  const uint32_t t = thresh >> 16;
  return floatx4{(h.x & 0xffffu) >= t ? scale : 0.0f, (h.x >> 16) >= t ? scale : 0.0f, (h.y & 0xffffu) >= t ? scale : 0.0f, (h.y >> 16) >= t ? scale : 0.0f};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// The same sum through the DPP cross-lane paths of the VALU instead of six ds_bpermute round trips through the LDS crossbar (a
// dependent chain of ~6 x 60-100 clk per reduction: the LayerNorm kernels, one or two reductions per row, spent a third of a row's
// time in it).  quad_perm -> row_half_mirror -> row_mirror leave every lane with its 16-lane row's sum; row_bcast:15 / :31 carry the
// row sums into rows 1, 3 and then 2, 3, so lane 63 (and every lane of row 3) holds the total.  Another summation ORDER than
// wave_sum: results agree to fp32 rounding, not bit for bit.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float row16_sum_dpp(float v) {      // every lane: the sum over its aligned 16-lane row
  v = dpp_add<0xB1, 0xF>(v);       // quad_perm:[1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);       // quad_perm:[2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);      // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);      // row_mirror
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {        // wave-uniform result
  v = row16_sum_dpp(v);
  v = dpp_add<0x142, 0xA>(v);      // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);      // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float half_sum_dpp(float v, int half) {      // sum over the lane's 32-lane half (half = lane >> 5)
  v = row16_sum_dpp(v);
  v = dpp_add<0x142, 0xA>(v);      // lanes 16-31 / 48-63 now hold the sums of half 0 / half 1
  const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
  const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  return half ? s1 : s0;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}
// Speed-mode GELU: erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below bf16 resolution) - one v_exp_f32, one
// v_rcp_f32 and 6 FMAs instead of the ~45-instruction erff; the same exponential serves the density term of the gradient.
// T = float (parity mode) keeps erff / expf.
// (contraction is pinned off inside: every kernel that inlines these - LDS-staged or register epilogue, forward or backward - must
// round identically, whatever the surrounding code invites the compiler to fuse)
__device__ __forceinline__ void gelu_parts_fast(float x, float& two_phi, float& e) {
#pragma clang fp contract(off)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  e = __expf(-z * z);                                        // exp(-x^2 / 2)
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
  const float q = p * t * e;                                 // erfc(|x| / sqrt 2)
  two_phi = x < 0.f ? q : 2.0f - q;                          // 2 * Phi(x), no cancellation in the negative tail
}
template <typename T> __device__ __forceinline__ float gelu_fwd(float x) {
#pragma clang fp contract(off)
  if constexpr (sizeof(T) == 4) return gelu_erf(x);
  else { float tp, e; gelu_parts_fast(x, tp, e); return 0.5f * x * tp; }
}
template <typename T> __device__ __forceinline__ float gelu_bwd(float x) {
#pragma clang fp contract(off)
  if constexpr (sizeof(T) == 4) return gelu_erf_grad(x);
  else { float tp, e; gelu_parts_fast(x, tp, e); return fmaf(x * 0.39894228040143267794f, e, 0.5f * tp); }
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// GRU gate nonlinearities: libm forms in parity mode (T = float); in speed mode the v_exp_f32 forms (relative error ~1e-6, far inside
// bf16 resolution; tanh(x) = 1 - 2 / (exp(2x) + 1) saturates cleanly at +-1 for large |x|).  Shared by gru_step_fwd and the fused
// GRU epilogue of the 8-wave GEMM, which must give the same bits.
template <typename T> __device__ __forceinline__ float gru_sigmoid(float x) {
  if constexpr (sizeof(T) == 4) return sigmoidf_(x); else return __frcp_rn(1.0f + __expf(-x));
}
template <typename T> __device__ __forceinline__ float gru_tanh(float x) {
  if constexpr (sizeof(T) == 4) return tanhf(x); else return 1.0f - 2.0f * __frcp_rn(__expf(2.0f * x) + 1.0f);
}
// one hidden unit of a GRU step (models.py:818-826, nn.GRU): gi_* = W_ih x + b_ih, gh_* = W_hh h + b_hh.  Contraction is pinned off and
// the one multiply-add that matters is spelled out, so the gate kernel and the fused GEMM epilogue - two translation units - round alike.
template <typename T>
__device__ __forceinline__ void gru_unit(float gi_r, float gi_z, float gi_n, float gh_r, float gh_z, float gh_n, float h_prev, float& r, float& z,
                                         float& n, float& h) {
#pragma clang fp contract(off)
  r = gru_sigmoid<T>(gi_r + gh_r);
  z = gru_sigmoid<T>(gi_z + gh_z);
  n = gru_tanh<T>(gi_n + r * gh_n);
  h = (1.0f - z) * n + z * h_prev;
}

// XCD-aware bijective block remap (8 XCDs; block b is observed on XCD b % 8): each XCD gets a
// contiguous run of logical tiles so neighbouring tiles share operand panels in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}
// CU pairing (round 6 probe).  With two workgroups per CU the dispatcher of an idle XCD places local workgroups j and j + 32 on the same
// CU (32 CUs, round-robin, observed); this permutation of the local index makes those two the logical neighbours 2j and 2j + 1 - two
// column tiles of one tile row, which fetch the SAME A K-tiles at about the same time: the second fetch can hit the CU's vector L1
// instead of occupying a miss slot.  Identity on an incomplete last block of 64.  Changes placement only, never results.
__device__ __forceinline__ int cu_pair_local(int j, int run) {
  const int blk = j & ~63, r = j & 63;
  if (blk + 64 > run) return j;
  return blk + (r < 32 ? 2 * r : 2 * (r - 32) + 1);
}
__device__ __forceinline__ int xcd_remap_paired(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + cu_pair_local(bid >> 3, x < r ? q + 1 : q);
}

}  // namespace rl

// Probe build (python -m realise_amd.build --probes -> librealise_hip_probes.so): the measured-and-rejected kernel variants
// (realise_set_nt_variant 1..8 / 11..44, realise_set_tn_variant 8) and the "no fetch / no MFMA" probe modes of the production kernels.
// The production library compiles them out: its kernels carry no probe branches.
#ifndef RL_PROBES
#define RL_PROBES 0
#endif

#define RL_OK 0
#define RL_ERR_ARG 1
#define RL_ERR_LAUNCH 2
