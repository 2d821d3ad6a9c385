// HBM-bound kernels, part 1: masks, embeddings + LayerNorm (K1/K4/K10), bias gradients, dropout,
// masked cross-entropy (K13), gate fusion (K11).  All row-wise kernels use one 64-lane wave per
// row with 8-16 byte per-lane accesses (a 768-wide row = 3 coalesced 4-element vectors per lane);
// statistics are always fp32.
#include "ops.h"

namespace rl {
__global__ void col_fold_kernel(float* __restrict__ slots, int slot_stride, int nrec, int n, float* out0, float* out1, int C, int overwrite, int clean, float alpha);

#define RL_LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH)
static constexpr int LN_MAXV = 4;   // up to 4 x (64 lanes x 4 elems) = 1024 columns per row

__global__ void mask_to_additive_kernel(const int64_t* __restrict__ m, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (1.0f - (float)m[i]) * -10000.0f;       // modeling_bert.py:696-697
}
int mask_to_additive(hipStream_t st, const int64_t* masks, float* out, int n) {
  hipLaunchKernelGGL(mask_to_additive_kernel, dim3((n + 255) / 256), dim3(256), 0, st, masks, out, n);
  return RL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// LayerNorm forward (BertEmbeddings modeling_bert.py:183-193, BertSelfOutput/BertOutput :273-277,
// :339-343, resnet_layernorm models.py:838)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) ln_fwd_kernel(LnFwdArgs<T> a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.rows) return;
  const int H = a.H;
  floatx4 v[LN_MAXV];
  float sum = 0.f;
  const int64_t tok = (a.in_mode == 1) ? a.ids[row] : 0;
  const int prow = a.pos_zero ? 0 : (row % a.S);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (c < H) {
      floatx4 x;
      if (a.in_mode == 1) x = *(const floatx4*)(a.word + tok * H + c);
      else x = load4<T>(a.x + (int64_t)(a.row_index ? a.row_index[row] : row) * H + c);
      if (a.in_mode != 0) x += *(const floatx4*)(a.pos + (int64_t)prow * H + c) + *(const floatx4*)(a.type0 + c);
      v[i] = x;
      sum += x[0] + x[1] + x[2] + x[3];
    }
  }
  const float mean = wave_sum(sum) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; sq += d * d; }
    }
  }
  const float var = wave_sum(sq) / (float)H;
  const float rstd = 1.0f / sqrtf(var + a.eps);
  if (lane == 0 && a.rstd != nullptr) a.rstd[row] = rstd;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      const floatx4 gm = *(const floatx4*)(a.gamma + c), bt = *(const floatx4*)(a.beta + c);
      floatx4 xh, y;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[j] = (v[i][j] - mean) * rstd;
        y[j] = xh[j] * gm[j] + bt[j];
      }
      y *= drop_mult4(a.drop.seed, a.drop.thresh, a.drop.scale, (uint32_t)row * (uint32_t)H + (uint32_t)c);       // H % 4 == 0
      if (a.xhat != nullptr) store4<T>(a.xhat + (int64_t)row * H + c, xh);
      store4<T>(a.y + (int64_t)row * H + c, y);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 fast path of the two kernels below (H % 256 == 0, H <= 1024, plain input): HALF a wave owns a row - 32 lanes x NV8 chunks
// of 8 consecutive columns - so every global access is 16 bytes per lane (the 8-byte accesses of the one-wave-per-row form run at
// 0.55-0.7 of that rate, MI355X_MICROARCH.md) and a wave-instruction covers two whole 512-byte row segments; a wave keeps RP
// row pairs in flight (all loads issued before the first reduction).  Reductions run inside the 32-lane half.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_sum(float v) {        // sum over the 32 lanes of this half-wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void unpack8(const uint4 u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint4 u;
  u.x = pack2bf(f[0], f[1]); u.y = pack2bf(f[2], f[3]); u.z = pack2bf(f[4], f[5]); u.w = pack2bf(f[6], f[7]);
  return u;
}

template <int NV8, int RP, bool DPP>
__global__ void __launch_bounds__(256) ln_fwd16_kernel(LnFwdArgs<bf16_t> a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
  const int H = a.H;
  const int row0 = (blockIdx.x * 4 + wave) * (2 * RP) + half;
  if (a.row_live != nullptr) {           // (the wave's 2 RP rows lie in one 16-row block: wave-uniform)
    static_assert(16 % (2 * RP) == 0, "a wave's rows must not straddle 16-row blocks");
    const uint4 lv = *(const uint4*)(a.row_live + (((blockIdx.x * 4 + wave) * (2 * RP)) & ~15));
    if ((lv.x | lv.y | lv.z | lv.w) == 0u) return;
  }
  uint4 raw[RP][NV8];
#pragma unroll
  for (int r = 0; r < RP; ++r) {
    const int row = row0 + 2 * r;
#pragma unroll
    for (int i = 0; i < NV8; ++i)
      raw[r][i] = row < a.rows ? *(const uint4*)((const char*)a.x + ((int64_t)row * H + (i * 32 + hl) * 8) * 2) : uint4{0u, 0u, 0u, 0u};
  }
  float gm[NV8][8], bt[NV8][8];
#pragma unroll
  for (int i = 0; i < NV8; ++i) {
    const int c = (i * 32 + hl) * 8;
    *(floatx4*)&gm[i][0] = *(const floatx4*)(a.gamma + c); *(floatx4*)&gm[i][4] = *(const floatx4*)(a.gamma + c + 4);
    *(floatx4*)&bt[i][0] = *(const floatx4*)(a.beta + c); *(floatx4*)&bt[i][4] = *(const floatx4*)(a.beta + c + 4);
  }
#pragma unroll
  for (int r = 0; r < RP; ++r) {
    const int row = row0 + 2 * r;
    float v[NV8][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
      unpack8(raw[r][i], v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    }
    const float mean = (DPP ? half_sum_dpp(sum, half) : half_sum(sum)) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq += d * d; }
    const float rstd = 1.0f / sqrtf((DPP ? half_sum_dpp(sq, half) : half_sum(sq)) / (float)H + a.eps);
    if (row >= a.rows) continue;
    if (hl == 0 && a.rstd != nullptr) a.rstd[row] = rstd;
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
      const int c = (i * 32 + hl) * 8;
      float xh[8], y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xh[j] = (v[i][j] - mean) * rstd;
        y[j] = xh[j] * gm[i][j] + bt[i][j];
      }
      if (a.drop.thresh != 0u) {
        const floatx4 d0 = drop_mult4(a.drop.seed, a.drop.thresh, a.drop.scale, (uint32_t)row * (uint32_t)H + (uint32_t)c);
        const floatx4 d1 = drop_mult4(a.drop.seed, a.drop.thresh, a.drop.scale, (uint32_t)row * (uint32_t)H + (uint32_t)c + 4u);
#pragma unroll
        for (int j = 0; j < 4; ++j) { y[j] *= d0[j]; y[4 + j] *= d1[j]; }
      }
      if (a.xhat != nullptr) *(uint4*)((char*)a.xhat + ((int64_t)row * H + c) * 2) = pack8(xh);
      *(uint4*)((char*)a.y + ((int64_t)row * H + c) * 2) = pack8(y);
    }
  }
}

static int g_ce_fast = 1;                            // bf16 cross-entropy with the row in registers (realise_set_ln key 4)
void set_ce_fast(int on) { g_ce_fast = on; }
static int g_ln_fast = 1, g_ln_bwd_blocks = 512;
// round 5 (realise_set_ln key 5): 1 = the row reductions of the bf16 LayerNorm kernels through DPP (wave_sum_dpp) and the backward on
// ln_bwd16v2_kernel (no barrier before the first row, four rows of a wave in flight, one-barrier epilogue); 0 = the round-4 kernels
static int g_ln_v2 = 1, g_ln_bwd_blocks_v2 = 256;
void set_ln_v2(int on) { g_ln_v2 = on; }
static int g_bn_fast = 1, g_bn_chunks = 1024;       // bf16 16-byte BatchNorm / column-reduction kernels; row chunks (= workgroups) of the reductions
static int g_bn_onepass = 1;                        // bf16 training statistics in one pass about the running mean (bn_stats_train16)
void set_bn_fast(int on) { g_bn_fast = on & 1; g_bn_onepass = (on & 1) && !(on & 2); }
void set_bn_chunks(int n) { if (n >= 1 && n <= 4096) g_bn_chunks = n; }
int bn_fast() { return g_bn_fast; }
void set_ln_fast(int on) { g_ln_fast = on; }
void set_ln_bwd_blocks(int n) { if (g_ln_v2) g_ln_bwd_blocks_v2 = n > 0 ? n : 256; else g_ln_bwd_blocks = n > 0 ? n : 512; }      // (of the active variant)

template <typename T> int ln_fwd_fast(hipStream_t st, const LnFwdArgs<T>& a) { return -1; }
template <> int ln_fwd_fast<bf16_t>(hipStream_t st, const LnFwdArgs<bf16_t>& a) {
  if (!g_ln_fast || a.in_mode != 0 || a.row_index != nullptr || (a.H % 256) != 0 || a.H > 1024) return -1;
  constexpr int RP = 2;
  const int blocks = (a.rows + 4 * 2 * RP - 1) / (4 * 2 * RP);
#define RL_LNF(NV) do { if (g_ln_v2) hipLaunchKernelGGL((ln_fwd16_kernel<NV, RP, true>), dim3(blocks), dim3(256), 0, st, a); \
                       else hipLaunchKernelGGL((ln_fwd16_kernel<NV, RP, false>), dim3(blocks), dim3(256), 0, st, a); } while (0)
  switch (a.H / 256) {
    case 1: RL_LNF(1); break;
    case 2: RL_LNF(2); break;
    case 3: RL_LNF(3); break;
    default: RL_LNF(4); break;
  }
#undef RL_LNF
  return RL_LAUNCH_CHECK();
}

template <typename T> int ln_fwd(hipStream_t st, const LnFwdArgs<T>& a) {
  if (a.rows <= 0) return RL_OK;
  if ((a.H & 3) || a.H > LN_MAXV * 256) return RL_ERR_ARG;
  { const int rc = ln_fwd_fast<T>(st, a); if (rc >= 0) return rc; }
  hipLaunchKernelGGL((ln_fwd_kernel<T>), dim3((a.rows + 3) / 4), dim3(256), 0, st, a);
  return RL_LAUNCH_CHECK();
}
template int ln_fwd<bf16_t>(hipStream_t, const LnFwdArgs<bf16_t>&);
template int ln_fwd<float>(hipStream_t, const LnFwdArgs<float>&);

// ---------------------------------------------------------------------------------------------
// LayerNorm backward: dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)); dgamma/dbeta
// partials are kept in registers across a grid-stride loop over rows, reduced through LDS, then
// one atomicAdd per column per workgroup.
// ---------------------------------------------------------------------------------------------
template <typename T, int NV>
__global__ void __launch_bounds__(256) ln_bwd_kernel(LnBwdArgs<T> a) {
  __shared__ float red[2][4][NV * 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int H = a.H;
  floatx4 dg[NV], db[NV], gm[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    dg[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    db[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int c = (i * 64 + lane) * 4;
    gm[i] = c < H ? *(const floatx4*)(a.gamma + c) : floatx4{0.f, 0.f, 0.f, 0.f};
  }
  // padding rows (row_live[row] == 0, a wave owns a row: wave-uniform): dy is an exact zero, so dx is - written without reading the row
  auto dead = [&](int row) -> bool { return a.row_live != nullptr && row < a.rows && a.row_live[row] == 0; };
  auto load_row = [&](int row, floatx4 (&dy)[NV], floatx4 (&xh)[NV]) {
    const bool skip = dead(row);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      dy[i] = floatx4{0.f, 0.f, 0.f, 0.f};
      xh[i] = floatx4{0.f, 0.f, 0.f, 0.f};
      if (c < H && row < a.rows && !skip) {
        dy[i] = load4<T>(a.dy + (int64_t)row * H + c);
        xh[i] = load4<T>(a.xhat + (int64_t)row * H + c);
      }
    }
  };
  auto do_row = [&](int row, floatx4 (&dy)[NV], floatx4 (&xh)[NV]) {
    if (dead(row)) {
      const floatx4 z4 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < H) { store4<T>(a.dx + (int64_t)row * H + c, z4); if (a.dx_drop != nullptr) store4<T>(a.dx_drop + (int64_t)row * H + c, z4); }
      }
      return;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        const floatx4 dm_in = drop_mult4(a.in_drop.seed, a.in_drop.thresh, a.in_drop.scale, (uint32_t)row * (uint32_t)H + (uint32_t)c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dy[i][j] *= dm_in[j];
          const float t = dy[i][j] * gm[i][j];
          s1 += t;
          s2 += t * xh[i][j];
          dg[i][j] += dy[i][j] * xh[i][j];
          db[i][j] += dy[i][j];
        }
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
    const float rstd = a.rstd[row];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        floatx4 dx, dxd;
#pragma unroll
        for (int j = 0; j < 4; ++j) dx[j] = rstd * (dy[i][j] * gm[i][j] - s1 - xh[i][j] * s2);
        dxd = dx * drop_mult4(a.out_drop.seed, a.out_drop.thresh, a.out_drop.scale, (uint32_t)row * (uint32_t)H + (uint32_t)c);
        store4<T>(a.dx + (int64_t)row * H + c, dx);
        if (a.dx_drop != nullptr) store4<T>(a.dx_drop + (int64_t)row * H + c, dxd);
      }
    }
  };
  // two rows of a wave in flight: both rows' loads are issued before either row's reductions (one row at a time left the kernel at
  // 1.9 TB/s: load, two wave reductions, store, and only then the next row's loads)
  const int stride = gridDim.x * 4;
  for (int row = blockIdx.x * 4 + wave; row < a.rows; row += 2 * stride) {
    floatx4 dy0[NV], xh0[NV], dy1[NV], xh1[NV];
    load_row(row, dy0, xh0);
    load_row(row + stride, dy1, xh1);
    do_row(row, dy0, xh0);
    if (row + stride < a.rows) do_row(row + stride, dy1, xh1);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][wave][c + j] = dg[i][j]; red[1][wave][c + j] = db[i][j]; }
  }
  __syncthreads();
  // With scratch: this workgroup's [dgamma | dbeta] partial goes to its own record as plain stores and ln_fold_kernel adds the
  // records in a fixed order - no atomics, no zero-fill, bitwise reproducible.  Without scratch (few rows, C-ABI callers):
  // atomics straight into the gradients.
  float* rec = a.slots != nullptr ? a.slots + (int64_t)blockIdx.x * 2 * H : nullptr;
  for (int c = threadIdx.x; c < H; c += 256) {
    const float g = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    const float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    if (rec != nullptr) { rec[c] = g; rec[H + c] = b; }
    else {
      if (a.dgamma != nullptr) atomicAdd(a.dgamma + c, g);
      if (a.dbeta != nullptr) atomicAdd(a.dbeta + c, b);
    }
  }
}

// bf16 fast path of LayerNorm backward (H % 16 == 0, H <= 1024).  The fused [dgamma | dbeta] accumulators are what costs registers
// here, and registers are what decides how many rows a CU keeps in flight (the 8-byte-per-lane form: 146 VGPRs, 3 waves per SIMD,
// 1.9 TB/s; a half-wave-per-row form with 24 columns per lane: 186 VGPRs, 2.2 TB/s).  This form gives a WAVE one row and a lane two
// 8-column chunks (columns 8l.. and H/2 + 8l..: H / 16 active lanes - 48 of 64 at H = 768; the idle quarter costs nothing on a
// memory-bound kernel): 16-byte accesses, 32 accumulator registers, and the next row's loads are issued before the current row is
// processed (two register sets, statically named), so every wave always has a row in flight.
struct LnRow16 { uint4 dy[2], xh[2]; float rstd; };      // (rstd rides with the row's data loads: loaded after the reductions it was an exposed latency per row)
template <bool DROP>
__global__ void __launch_bounds__(256, 3) ln_bwd16_kernel(LnBwdArgs<bf16_t> a) {
  // LDS: gamma [H] + one [4 waves][H] reduction buffer used twice (dgamma, then dbeta): 15 KB at H = 768.  The footprint matters
  // beyond occupancy: in a training step this kernel runs NEXT TO the grouped weight-gradient GEMM of the side stream, whose two
  // 64 KB workgroups leave 32 KB of a CU's LDS - with a 36 KB footprint these workgroups queued behind it (58 us per call under
  // overlap against 33 alone).
  extern __shared__ float ln_lds[];
  float* gsm = ln_lds;                 // [H]
  float* red = ln_lds + a.H;           // [4][H]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int H = a.H, HH = H >> 1;
  const bool active = lane * 16 < H;
  const int c0 = lane * 8, c1 = HH + lane * 8;
  // Row liveness of every row this wave will visit, fetched ONCE up front (lane k holds the flag of the wave's k-th row): read inside
  // the row loop, the flag byte was a dependent global load in front of every row's data loads - a memory latency per row on a
  // wave that only visits four.  It is also the FIRST load of the kernel, ahead of gamma: the first rows' data loads wait on it, and
  // they are issued before gamma goes to the LDS (gamma -> barrier -> flags -> rows was three memory latencies in a row on a
  // workgroup that lives for about ten).
  const int stride = gridDim.x * 4;
  const int row_first = blockIdx.x * 4 + wave;
  int live_reg = 1;
  if (a.row_live != nullptr) {
    const int64_t rk = (int64_t)row_first + (int64_t)lane * stride;
    live_reg = rk < a.rows ? (int)a.row_live[rk] : 0;
  }
  float gpre[4];                       // H <= 1024: at most four gamma elements per thread
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int c = threadIdx.x + 256 * i; gpre[i] = c < H ? a.gamma[c] : 0.f; }
  float dg[2][8], db[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; }
  auto is_live = [&](int row) -> bool {          // wave-uniform
    if (a.row_live == nullptr) return true;
    const int k = (row - row_first) / stride;
    if (k < 64) return __builtin_amdgcn_readlane(live_reg, k) != 0;
    return row < a.rows && a.row_live[row] != 0;
  };
  auto load = [&](int row, LnRow16& r) {
    const bool ok = active && row < a.rows && is_live(row);
    const uint32_t o0 = ((uint32_t)row * (uint32_t)H + (uint32_t)c0) * 2u, o1 = o0 + (uint32_t)H;      // rows * H * 2 < 4 GiB (launcher)
    const uint4 z = uint4{0u, 0u, 0u, 0u};
    r.dy[0] = ok ? *(const uint4*)((const char*)a.dy + o0) : z; r.dy[1] = ok ? *(const uint4*)((const char*)a.dy + o1) : z;
    r.xh[0] = ok ? *(const uint4*)((const char*)a.xhat + o0) : z; r.xh[1] = ok ? *(const uint4*)((const char*)a.xhat + o1) : z;
    r.rstd = ok ? a.rstd[row] : 0.f;
  };
  auto process = [&](int row, LnRow16& r) {
    if (!is_live(row)) {      // a padding row (wave-uniform): dy = 0 -> dx = 0, no dgamma / dbeta term
      if (active) {
        const uint4 z = uint4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint32_t off = ((uint32_t)row * (uint32_t)H + (uint32_t)(i ? c1 : c0)) * 2u;
          *(uint4*)((char*)a.dx + off) = z;
          if (a.dx_drop != nullptr) *(uint4*)((char*)a.dx_drop + off) = z;
        }
      }
      return;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = i ? c1 : c0;
      float dy[8], xh[8], gm[8];
      unpack8(r.dy[i], dy);
      unpack8(r.xh[i], xh);
      *(floatx4*)&gm[0] = *(const floatx4*)&gsm[active ? c : 0]; *(floatx4*)&gm[4] = *(const floatx4*)&gsm[active ? c + 4 : 4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = dy[j] * gm[j];
        s1 += t;
        s2 += t * xh[j];
        dg[i][j] += dy[j] * xh[j];
        db[i][j] += dy[j];
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
    if (!active || row >= a.rows) return;
    const float rstd = r.rstd;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = i ? c1 : c0;
      float dy[8], xh[8], gm[8], dx[8];
      unpack8(r.dy[i], dy);
      unpack8(r.xh[i], xh);
      *(floatx4*)&gm[0] = *(const floatx4*)&gsm[c]; *(floatx4*)&gm[4] = *(const floatx4*)&gsm[c + 4];
#pragma unroll
      for (int j = 0; j < 8; ++j) dx[j] = rstd * (dy[j] * gm[j] - s1 - xh[j] * s2);
      const uint32_t off = ((uint32_t)row * (uint32_t)H + (uint32_t)c) * 2u;
      *(uint4*)((char*)a.dx + off) = pack8(dx);
      if (a.dx_drop != nullptr) {
        if constexpr (DROP) {
#pragma unroll
          for (int hq = 0; hq < 2; ++hq) {
            const floatx4 dm = drop_mult4(a.out_drop.seed, a.out_drop.thresh, a.out_drop.scale, (uint32_t)row * (uint32_t)H + (uint32_t)(c + 4 * hq));
#pragma unroll
            for (int j = 0; j < 4; ++j) dx[4 * hq + j] *= dm[j];
          }
        }
        *(uint4*)((char*)a.dx_drop + off) = pack8(dx);
      }
    }
  };
  int row = row_first;
  LnRow16 ra, rb;
  load(row, ra);
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int c = threadIdx.x + 256 * i; if (c < H) gsm[c] = gpre[i]; }
  __syncthreads();
  while (row < a.rows) {
    load(row + stride, rb);
    process(row, ra);
    row += stride;
    if (row >= a.rows) break;
    load(row + stride, ra);
    process(row, rb);
    row += stride;
  }
  float* rec = a.slots != nullptr ? a.slots + (int64_t)blockIdx.x * 2 * H : nullptr;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();                   // (pass 0: the gamma reads of the row loop are done; pass 1: pass 0's sums are read)
    if (active) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = i ? c1 : c0;
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave * H + c + j] = pass ? db[i][j] : dg[i][j];
      }
    }
    __syncthreads();
    float* out = pass ? a.dbeta : a.dgamma;
    for (int c = threadIdx.x; c < H; c += 256) {
      const float v = red[c] + red[H + c] + red[2 * H + c] + red[3 * H + c];
      if (rec != nullptr) rec[pass * H + c] = v;
      else if (out != nullptr) atomicAdd(out + c, v);
    }
  }
}
// Round 5: the same row arithmetic on a leaner skeleton.  tools/ln_probe.py's block sweep put the FIXED cost of a workgroup of the
// kernel above at ~8 us against ~1.5 us per row: a prologue of three dependent memory latencies (liveness flags -> first rows -> gamma
// through the LDS and a barrier), an epilogue of two LDS passes with four barriers, and per row two wave reductions of six
// ds_bpermute round trips each.  Here
//   * nothing waits before the first row: the flags, gamma (straight into 16 registers per lane - no LDS, no barrier) and the first
//     row's data are requested together (that row is fetched whether or not it is live; the later ones only when live);
//   * a wave keeps FOUR rows in flight (statically named register sets) and walks twice as many rows, so half as many workgroups
//     pay the fixed cost and write records;
//   * the row sums go through DPP (wave_sum_dpp);
//   * the epilogue is ONE barrier: every wave stores its 2 x 16 partial columns as 16-byte LDS writes, 256 threads add the four
//     waves in order and write the record as 16-byte stores.  LDS 32 H bytes (24 KB at H = 768), touched by the epilogue only.
// Same record layout ([dgamma | dbeta] per workgroup), same fold kernels; a wave's rows are summed in row order, the four waves in
// wave order: deterministic, but another order than the kernel above.
typedef __attribute__((ext_vector_type(4))) uint32_t ln_u32x4;
// STORES THE COMPILER DOES NOT TRACK.  hipcc's waitcnt insertion treats vmcnt as out of order as soon as loads and stores are pending
// together (one counter for both on gfx9; a younger store may indeed be acknowledged before an older load) and then waits with
// vmcnt(0): in a row loop that stores row k while rows k+1.. are in flight, EVERY use of a loaded row drained the whole queue - the
// prefetched rows included - which is why the round-3/4 kernel gained nothing from more rows in flight.  With the stores as inline asm
// the compiler sees only loads pending, which complete in issue order: it waits with exact counts (vmcnt(12) before row k = "the
// three younger rows may still be in flight"); the untracked stores can only make such a wait longer, never too short.  (The first
// form of this kernel did it the other way round - asm loads, counted waits by hand - and read garbage: the compiler is free to
// copy or spill a value it believes was defined by the asm statement while the load is still in flight.)  s_nop: the store-data
// hazard of a > 8-byte VMEM store followed by a write of its data registers, which the compiler cannot see through the asm.
__device__ __forceinline__ void ln_store16(ln_u32x4 v, uint32_t voff, uint32_t soff, __amdgpu_buffer_rsrc_t rsrc) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#endif
}
struct LnRowA { ln_u32x4 dy[2], xh[2]; };

template <bool DROP>
__global__ void __launch_bounds__(256, 1) ln_bwd16v2_kernel(LnBwdArgs<bf16_t> a) {
  // Branch-free row path: every global access of the row loop is a raw buffer access - the row's byte offset in the SCALAR offset
  // operand, the lane's column offset in the vector operand, which is parked beyond the buffer for whatever must not happen (a lane
  // beyond H / 16, a row beyond this wave's last, the loads of a padding row): out-of-range loads return zeros and move no bytes,
  // out-of-range stores are dropped (the range check looks at the vector offset only).  A padding row runs the same arithmetic on
  // zeros (dx = 0 * (0 - 0 - 0) = 0, nothing added to dgamma / dbeta) and stores its zero row: the row loop is straight-line code.
  extern __shared__ float ln_lds[];          // [4 waves][2 H]: epilogue only
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int H = a.H, HH = H >> 1;
  const bool active = lane * 16 < H;
  constexpr uint32_t OOB = 0xFFFFFF00u;
  const uint32_t nbytes = (uint32_t)a.rows * (uint32_t)H * 2u;              // < 4 GiB - 256 (launcher)
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_xh = __builtin_amdgcn_make_buffer_rsrc((void*)a.xhat, 0, (int)nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dx = __builtin_amdgcn_make_buffer_rsrc((void*)a.dx, 0, (int)nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dx_drop != nullptr ? a.dx_drop : a.dx), 0, a.dx_drop != nullptr ? (int)nbytes : 0, 0x00020000);
  const int c0 = active ? lane * 8 : 0, c1 = active ? HH + lane * 8 : 0;
  const uint32_t lo0 = active ? (uint32_t)c0 * 2u : OOB, lo1 = active ? (uint32_t)c1 * 2u : OOB;      // byte offsets inside a row; idle lanes parked
  const int stride = gridDim.x * 4;
  const int row_first = blockIdx.x * 4 + wave;
  const int nrow = row_first < a.rows ? (a.rows - row_first + stride - 1) / stride : 0;      // this wave's rows: row_first + k * stride, k < nrow <= 64
  const uint32_t row_bytes = (uint32_t)H * 2u;
  // (bit-ands, not &&: no short-circuit branches around the loads)
  auto load = [&](int k, LnRowA& r, int fetch) {
    const uint32_t row = (uint32_t)(row_first + k * stride);
    const uint32_t sbase = fetch ? row * row_bytes : 0u;
    const uint32_t v0 = fetch ? lo0 : OOB, v1 = fetch ? lo1 : OOB;
    r.dy[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, v0, sbase, 0);
    r.dy[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, v1, sbase, 0);
    r.xh[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_xh, v0, sbase, 0);
    r.xh[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_xh, v1, sbase, 0);
  };
  LnRowA r0, r1, r2, r3;
  // ---- prologue: the liveness flags (lane k: is the wave's k-th row live; no table: every row), gamma and the FIRST row's data are
  // requested together - one memory latency, not three in a row; that row is fetched whether or not it is live
  int live_reg = 1;
  if (a.row_live != nullptr) live_reg = lane < nrow ? (int)a.row_live[(int64_t)row_first + (int64_t)lane * stride] : 0;
  const float rstd_reg = lane < nrow ? a.rstd[(int64_t)row_first + (int64_t)lane * stride] : 0.f;      // lane k: rstd of the wave's k-th row
  float gm[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i ? c1 : c0;
    *(floatx4*)&gm[i][0] = *(const floatx4*)(a.gamma + c); *(floatx4*)&gm[i][4] = *(const floatx4*)(a.gamma + c + 4);
  }
  load(0, r0, (int)(nrow > 0));
  const float rstd_keep = rstd_reg;
  auto live_k = [&](int k) -> int { return (int)(k < nrow) & (int)(__builtin_amdgcn_readlane(live_reg, k & 63) != 0); };      // k wave-uniform
  load(1, r1, live_k(1)); load(2, r2, live_k(2)); load(3, r3, live_k(3));
  float dg[2][8], db[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; }
  // keep = 0: a padding row (its loads were parked and returned zeros - except the wave's first row, fetched before the flags were known) or a row
  // beyond the wave's last: everything read for it counts as zeros
  auto process = [&](int k, const LnRowA& r, int keep) {
    const int store = (int)(k < nrow);
    const uint32_t row = (uint32_t)(row_first + k * stride);
    const uint32_t msk = keep ? 0xFFFFFFFFu : 0u;
    // (opaque to the loop optimiser: as row * H + c it became eight per-lane induction variables carried - and spilled - across the loop)
    const uint32_t row_h = (uint32_t)__builtin_amdgcn_readfirstlane((int)(row * (uint32_t)H));
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float dy[8], xh[8];
      unpack8(uint4{r.dy[i][0] & msk, r.dy[i][1] & msk, r.dy[i][2] & msk, r.dy[i][3] & msk}, dy);
      unpack8(uint4{r.xh[i][0] & msk, r.xh[i][1] & msk, r.xh[i][2] & msk, r.xh[i][3] & msk}, xh);      // (a stale xhat row may hold anything: 0 * NaN)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = dy[j] * gm[i][j];
        s1 += t;
        s2 += t * xh[j];
        dg[i][j] += dy[j] * xh[j];
        db[i][j] += dy[j];
      }
    }
    s1 = wave_sum_dpp(s1) / (float)H;
    s2 = wave_sum_dpp(s2) / (float)H;
    const float rstd = keep ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rstd_keep), k & 63)) : 0.f;      // (a padding row's saved rstd is as stale as its xhat)
    const uint32_t sbase = store ? row * row_bytes : 0u;
    // (fences: the dropout hashes and the second unpack are not to be hoisted above the reductions - they would all be live across them)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i) __builtin_amdgcn_sched_barrier(0);
      const int c = i ? c1 : c0;
      float dy[8], xh[8], dx[8];
      unpack8(uint4{r.dy[i][0] & msk, r.dy[i][1] & msk, r.dy[i][2] & msk, r.dy[i][3] & msk}, dy);
      unpack8(uint4{r.xh[i][0] & msk, r.xh[i][1] & msk, r.xh[i][2] & msk, r.xh[i][3] & msk}, xh);
#pragma unroll
      for (int j = 0; j < 8; ++j) dx[j] = rstd * (dy[j] * gm[i][j] - s1 - xh[j] * s2);
      const uint32_t voff = store ? (i ? lo1 : lo0) : OOB;
      ln_store16(__builtin_bit_cast(ln_u32x4, pack8(dx)), voff, sbase, rs_dx);
      if constexpr (DROP) {
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
          const floatx4 dm = drop_mult4_nz(a.out_drop.seed, a.out_drop.thresh, a.out_drop.scale, row_h + (uint32_t)(c + 4 * hq));
#pragma unroll
          for (int j = 0; j < 4; ++j) dx[4 * hq + j] *= dm[j];
        }
      }
      ln_store16(__builtin_bit_cast(ln_u32x4, pack8(dx)), voff, sbase, rs_dd);      // (no dx_drop: a zero-size buffer drops it)
    }
  };
  // Row k is processed, then row k + 4 requested into its registers: four rows of a wave in flight.  The waits are the compiler's (exact
  // counts: only loads are pending as far as it knows).  sched_barrier: it must not interleave four rows' arithmetic (> 256 registers).
#define RL_LN_STEP(K, R, KEEP) do { __builtin_amdgcn_sched_barrier(0); process(K, R, KEEP); __builtin_amdgcn_sched_barrier(0); \
    load((K) + 4, R, live_k((K) + 4)); __builtin_amdgcn_sched_barrier(0); } while (0)
  for (int k = 0; k < nrow; k += 4) {
    RL_LN_STEP(k, r0, live_k(k));
    RL_LN_STEP(k + 1, r1, live_k(k + 1));
    RL_LN_STEP(k + 2, r2, live_k(k + 2));
    RL_LN_STEP(k + 3, r3, live_k(k + 3));
  }
#undef RL_LN_STEP
  // ---- epilogue: [dgamma | dbeta] of the workgroup = the four waves' partials added in wave order
  float* mine = ln_lds + (size_t)wave * 2 * H;
  if (active) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = i ? c1 : c0;
      *(floatx4*)(mine + c) = *(const floatx4*)&dg[i][0]; *(floatx4*)(mine + c + 4) = *(const floatx4*)&dg[i][4];
      *(floatx4*)(mine + H + c) = *(const floatx4*)&db[i][0]; *(floatx4*)(mine + H + c + 4) = *(const floatx4*)&db[i][4];
    }
  }
  __syncthreads();
  float* rec = a.slots != nullptr ? a.slots + (int64_t)blockIdx.x * 2 * H : nullptr;
  for (int q = threadIdx.x; q < 2 * H / 4; q += 256) {
    const floatx4 v = ((*(const floatx4*)(ln_lds + 4 * q) + *(const floatx4*)(ln_lds + 2 * H + 4 * q)) + *(const floatx4*)(ln_lds + 4 * H + 4 * q)) +
                      *(const floatx4*)(ln_lds + 6 * H + 4 * q);
    if (rec != nullptr) *(floatx4*)(rec + 4 * q) = v;
    else {
      float* out = 4 * q < H ? a.dgamma : a.dbeta;
      const int c = 4 * q < H ? 4 * q : 4 * q - H;
      if (out != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(out + c + j, v[j]);
      }
    }
  }
}
// dgamma / dbeta += sum of the per-workgroup records: one float4 column group per 64 threads (record lanes k, k + 64, ...),
// four groups per block, fixed-order LDS tree.
__global__ void __launch_bounds__(256) ln_fold_kernel(const float* __restrict__ recs, int nrec, int H, float* dgamma, float* dbeta) {
  __shared__ floatx4 red[4][64];
  const int q = threadIdx.x >> 6, kl = threadIdx.x & 63;
  const int i = (blockIdx.x * 4 + q) * 4;                  // column in [0, 2H)
  floatx4 s = floatx4{0.f, 0.f, 0.f, 0.f};
  if (i < 2 * H)
    for (int k = kl; k < nrec; k += 64) s += *(const floatx4*)(recs + (int64_t)k * 2 * H + i);
  red[q][kl] = s;
  __syncthreads();
#pragma unroll
  for (int w = 32; w > 0; w >>= 1) {
    if (kl < w) red[q][kl] += red[q][kl + w];
    __syncthreads();
  }
  if (kl == 0 && i < 2 * H) {
    float* o = i < H ? dgamma + i : dbeta + (i - H);
    *(floatx4*)o += red[q][0];
  }
}
__global__ void __launch_bounds__(256) ln_fold_multi_kernel(LnFoldSites f) {
  __shared__ floatx4 red[4][64];
  const LnFoldSite site = f.s[blockIdx.y];
  const int H = f.H;
  const int q = threadIdx.x >> 6, kl = threadIdx.x & 63;
  const int i = (blockIdx.x * 4 + q) * 4;                  // column in [0, 2H)
  floatx4 s = floatx4{0.f, 0.f, 0.f, 0.f};
  if (i < 2 * H)
    for (int k = kl; k < site.nrec; k += 64) s += *(const floatx4*)(site.recs + (int64_t)k * 2 * H + i);
  red[q][kl] = s;
  __syncthreads();
#pragma unroll
  for (int w = 32; w > 0; w >>= 1) {
    if (kl < w) red[q][kl] += red[q][kl + w];
    __syncthreads();
  }
  if (kl == 0 && i < 2 * H) {
    float* o = i < H ? site.dgamma + i : site.dbeta + (i - H);
    *(floatx4*)o += red[q][0];
  }
}
int ln_fold_multi(hipStream_t st, const LnFoldSites& sites) {
  if (sites.n <= 0) return RL_OK;
  if (sites.n > LN_FOLD_MAX || (sites.H & 3)) return RL_ERR_ARG;
  hipLaunchKernelGGL(ln_fold_multi_kernel, dim3((2 * sites.H / 4 + 3) / 4, sites.n), dim3(256), 0, st, sites);
  return RL_LAUNCH_CHECK();
}

template <typename T> int ln_bwd(hipStream_t st, const LnBwdArgs<T>& a) {
  if (a.rows <= 0) return RL_OK;
  if ((a.H & 3) || a.H > LN_MAXV * 256) return RL_ERR_ARG;
  int blocks = (a.rows + 3) / 4;
  const int cap = a.H <= 512 ? 1024 : 768;   // resident workgroups: 4 (<= 110 VGPRs) or 3 (146 VGPRs with two rows in flight) waves per SIMD x 256 CUs
  if (blocks > cap) blocks = cap;
  LnBwdArgs<T> b = a;
  if (blocks <= 96 || a.dgamma == nullptr || a.dbeta == nullptr) b.slots = nullptr;
  const int nv = (a.H + 255) / 256;
  if constexpr (sizeof(T) == 2) {
    if (g_ln_fast && a.in_drop.thresh == 0u && (a.H % 16) == 0 && a.H <= 1024 && ((int64_t)a.rows + 8192) * a.H * 2 < (1ll << 32) - 4096) {     // (the embedding LayerNorm, whose input gradient is dropout-masked first, keeps the general kernel)
      const int groups = (a.rows + 3) / 4;
      if (g_ln_v2) {
        blocks = groups < g_ln_bwd_blocks_v2 ? groups : g_ln_bwd_blocks_v2;
        if (blocks * 256 < a.rows) blocks = (a.rows + 255) / 256;       // a wave's rows must fit its 64 liveness lanes
        if (blocks <= 96 || a.dgamma == nullptr || a.dbeta == nullptr) b.slots = nullptr; else b.slots = a.slots;
        if (b.slots != nullptr && blocks > 1024) return RL_ERR_ARG;     // (LN_SLOT_FLOATS holds 1024 records)
        const size_t lds2 = (size_t)8 * a.H * sizeof(float);
        if (a.out_drop.thresh != 0u) hipLaunchKernelGGL((ln_bwd16v2_kernel<true>), dim3(blocks), dim3(256), lds2, st, b);
        else hipLaunchKernelGGL((ln_bwd16v2_kernel<false>), dim3(blocks), dim3(256), lds2, st, b);
        if (a.deferred_records != nullptr) *a.deferred_records = b.slots != nullptr ? blocks : 0;
        else if (b.slots != nullptr) hipLaunchKernelGGL(ln_fold_kernel, dim3((2 * a.H / 4 + 3) / 4), dim3(256), 0, st, b.slots, blocks, a.H, a.dgamma, a.dbeta);
        return RL_LAUNCH_CHECK();
      }
      blocks = groups < g_ln_bwd_blocks ? groups : g_ln_bwd_blocks;
      if (blocks <= 96 || a.dgamma == nullptr || a.dbeta == nullptr) b.slots = nullptr; else b.slots = a.slots;
      const size_t lds = (size_t)5 * a.H * sizeof(float);
      if (a.out_drop.thresh != 0u) hipLaunchKernelGGL((ln_bwd16_kernel<true>), dim3(blocks), dim3(256), lds, st, b);
      else hipLaunchKernelGGL((ln_bwd16_kernel<false>), dim3(blocks), dim3(256), lds, st, b);
      if (a.deferred_records != nullptr) *a.deferred_records = b.slots != nullptr ? blocks : 0;
      else if (b.slots != nullptr) hipLaunchKernelGGL(ln_fold_kernel, dim3((2 * a.H / 4 + 3) / 4), dim3(256), 0, st, b.slots, blocks, a.H, a.dgamma, a.dbeta);
      return RL_LAUNCH_CHECK();
    }
  }
  if (nv == 1) hipLaunchKernelGGL((ln_bwd_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, b);
  else if (nv == 2) hipLaunchKernelGGL((ln_bwd_kernel<T, 2>), dim3(blocks), dim3(256), 0, st, b);
  else if (nv == 3) hipLaunchKernelGGL((ln_bwd_kernel<T, 3>), dim3(blocks), dim3(256), 0, st, b);
  else hipLaunchKernelGGL((ln_bwd_kernel<T, 4>), dim3(blocks), dim3(256), 0, st, b);
  if (a.deferred_records != nullptr) *a.deferred_records = b.slots != nullptr ? blocks : 0;
  else if (b.slots != nullptr) hipLaunchKernelGGL(ln_fold_kernel, dim3((2 * a.H / 4 + 3) / 4), dim3(256), 0, st, b.slots, blocks, a.H, a.dgamma, a.dbeta);
  return RL_LAUNCH_CHECK();
}
template int ln_bwd<bf16_t>(hipStream_t, const LnBwdArgs<bf16_t>&);
template int ln_bwd<float>(hipStream_t, const LnBwdArgs<float>&);

// ---------------------------------------------------------------------------------------------
// Embedding backward: thread owns 4 columns of one sequence position s and walks the batch.
// ---------------------------------------------------------------------------------------------
// Padded positions receive an exactly-zero gradient (masked as attention keys, excluded from the loss and from the
// masked mean), and they all carry token id 0: skipping all-zero quads is exact and removes the thousands-way
// same-address atomic pile-up on that one embedding row.
__device__ __forceinline__ bool quad_is_zero(floatx4 d) { return d[0] == 0.f && d[1] == 0.f && d[2] == 0.f && d[3] == 0.f; }
template <typename T>
__global__ void embed_bwd_kernel(const T* __restrict__ de, const int64_t* __restrict__ ids, int B, int S, int H,
                                 float* word_grad, float* pos_grad, int pos_zero, float* type_grad) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int s = blockIdx.y;
  if (c >= H) return;
  const int bchunk = (B + gridDim.z - 1) / gridDim.z;
  const int b0 = blockIdx.z * bchunk, b1 = min(B, b0 + bchunk);
  floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int b = b0; b < b1; ++b) {
    const int64_t row = (int64_t)b * S + s;
    const floatx4 d = load4<T>(de + row * H + c);
    if (quad_is_zero(d)) continue;
    acc += d;
    if (word_grad != nullptr) {
      float* w = word_grad + ids[row] * H + c;
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(w + j, d[j]);
    }
  }
  if (quad_is_zero(acc)) return;
  // The token-type row (every token adds to it) and, with position_ids == 0, the single position row are plain column sums of de:
  // they come from a column reduction (embed_bwd below) - as per-block atomics they were a 1024-way pile-up on 768 addresses
  // (~80 of this kernel's 117 us).
  if (pos_zero) return;
  float* p = pos_grad + (int64_t)s * H + c;
#pragma unroll
  for (int j = 0; j < 4; ++j) atomicAdd(p + j, acc[j]);
}
template <typename T>
int embed_bwd(hipStream_t st, const T* de, const int64_t* ids, int B, int S, int H, float* word_grad, float* pos_grad,
              int pos_zero, float* type_grad) {
  if (H & 3) return RL_ERR_ARG;
  const int tx = 64;
  const int bz = B >= 32 ? 8 : (B >= 8 ? 4 : 1);
  if (word_grad != nullptr || !pos_zero)
    hipLaunchKernelGGL((embed_bwd_kernel<T>), dim3((H / 4 + tx - 1) / tx, S, bz), dim3(tx), 0, st, de, ids, B, S, H, word_grad,
                       pos_grad, pos_zero, type_grad);
  int rc = bias_grad<T>(st, de, H, B * S, H, type_grad, nullptr);
  if (rc == RL_OK && pos_zero) rc = bias_grad<T>(st, de, H, B * S, H, pos_grad, nullptr);
  return rc != RL_OK ? rc : RL_LAUNCH_CHECK();
}
template int embed_bwd<bf16_t>(hipStream_t, const bf16_t*, const int64_t*, int, int, int, float*, float*, int, float*);
template int embed_bwd<float>(hipStream_t, const float*, const int64_t*, int, int, int, float*, float*, int, float*);

// ---------------------------------------------------------------------------------------------
// Column reductions over a row-major [rows][C] matrix: 32 threads x 4 columns wide, 8 row lanes.
// ---------------------------------------------------------------------------------------------
template <typename F>
__global__ void __launch_bounds__(256) col_reduce_kernel(F f, int rows_max, int C, int tpr_shift, float* out0, float* out1, RowBound rb,
                                                          float* slots, int slot_stride) {
  // tpr = threads per row (each owns 4 consecutive columns); 256 / tpr rows are read per pass, so a wave always touches
  // whole contiguous rows (C = 64: 16 threads x 4 cols = one 128-byte bf16 row, 4 rows per wave-instruction).
  __shared__ floatx4 red[2][256];
  const int tpr = 1 << tpr_shift, rpp = 256 >> tpr_shift;
  const int cx = threadIdx.x & (tpr - 1), ry = threadIdx.x >> tpr_shift;
  const int col = (blockIdx.x * tpr + cx) * 4;
  const int rows = rb_rows(rb, rows_max);
  const int chunk = (rows + gridDim.y - 1) / gridDim.y;          // live rows are re-split evenly over the grid
  const int r0 = blockIdx.y * chunk;
  const int r1 = min(rows, r0 + chunk);
  floatx4 a0 = floatx4{0.f, 0.f, 0.f, 0.f}, a1 = floatx4{0.f, 0.f, 0.f, 0.f};
  if (col < C) {
    // four independent accumulator pairs: the loads of four row passes are in flight together (one pass at a time left a
    // 64-column reduction over 957k rows at 1.2 TB/s: a thread's 8-byte load, then its dependent add, then the next load)
    floatx4 b0 = a0, b1 = a0, c0 = a0, c1 = a0, d0 = a0, d1 = a0;
    int r = r0 + ry;
    for (; r + 3 * rpp < r1; r += 4 * rpp) {
      f(r, col, a0, a1);
      f(r + rpp, col, b0, b1);
      f(r + 2 * rpp, col, c0, c1);
      f(r + 3 * rpp, col, d0, d1);
    }
    for (; r < r1; r += rpp) f(r, col, a0, a1);
    a0 = (a0 + b0) + (c0 + d0);
    a1 = (a1 + b1) + (c1 + d1);
  }
  red[0][threadIdx.x] = a0;
  red[1][threadIdx.x] = a1;
  __syncthreads();
  if (ry == 0 && col < C) {
    for (int k = 1; k < rpp; ++k) { a0 += red[0][k * tpr + cx]; a1 += red[1][k * tpr + cx]; }
    if (slot_stride) {     // deterministic path: this row chunk's partial sums go to their own [C | C] record
      float* rec = slots + (int64_t)blockIdx.y * slot_stride;      // col_fold_kernel adds the records up in a fixed order
      *(floatx4*)(rec + col) = a0;
      if (out1 != nullptr) *(floatx4*)(rec + C + col) = a1;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        atomicAdd(out0 + col + j, a0[j]);
        if (out1 != nullptr) atomicAdd(out1 + col + j, a1[j]);
      }
    }
  }
}
// out[i] (+)= sum over the nrec partial records, in a fixed order: 32 record lanes per column quad (each adds records
// k, k + 32, ... in order), then a fixed pairwise tree over the 32 lane sums.  One workgroup per 32 columns.
// overwrite: out = sum (no zero-fill needed before the reduction); clean: the records are left zeroed (accumulator records that the
// next launch adds into with atomics need no memset).
__global__ void __launch_bounds__(256) col_fold_kernel(float* __restrict__ slots, int slot_stride, int nrec, int n, float* out0,
                                                        float* out1, int C, int overwrite, int clean, float alpha) {
  __shared__ floatx4 red[32][8];
  const int cq = threadIdx.x & 7, kl = threadIdx.x >> 3;
  const int i = (blockIdx.x * 8 + cq) * 4;                 // i in [0, n): n = C or 2C laid out [C | C], C % 4 == 0
  floatx4 s = floatx4{0.f, 0.f, 0.f, 0.f};
  if (i < n)
    for (int k = kl; k < nrec; k += 32) {
      floatx4* p = (floatx4*)(slots + (int64_t)k * slot_stride + i);
      s += *p;
      if (clean) *p = floatx4{0.f, 0.f, 0.f, 0.f};
    }
  red[kl][cq] = s;
  __syncthreads();
#pragma unroll
  for (int w = 16; w > 0; w >>= 1) {
    if (kl < w) red[kl][cq] += red[kl + w][cq];
    __syncthreads();
  }
  if (kl == 0 && i < n) {
    float* o = i < C ? out0 + i : out1 + (i - C);
    if (overwrite) *(floatx4*)o = red[0][cq] * alpha;
    else *(floatx4*)o += red[0][cq] * alpha;
  }
}
template <typename F>
static int launch_col_reduce(hipStream_t st, const F& f, int rows, int C, float* out0, float* out1, RowBound rb = RowBound(),
                             float* slots = nullptr, float alpha = 1.0f) {
  if (rows <= 0 || C <= 0) return RL_OK;
  if (C & 3) return RL_ERR_ARG;
  int tpr_shift = 5;
  while (tpr_shift > 2 && (1 << tpr_shift) * 4 > C) --tpr_shift;        // 32 threads x 4 columns unless the row is narrower
  const int tpr = 1 << tpr_shift;
  const int gx = (C + tpr * 4 - 1) / (tpr * 4);
  int gy = 1024 / gx;
  if (gy < 1) gy = 1;
  const int rpp = 256 / tpr;
  const int max_gy = (rows + 4 * rpp - 1) / (4 * rpp);                  // at least 4 passes per workgroup
  if (gy > max_gy) gy = max_gy;
  if (gy < 1) gy = 1;
  if (slots != nullptr) {      // per-chunk partial records + ordered fold: bitwise reproducible (gy * 2C <= COL_SLOT_FLOATS)
    const int stride = 2 * C;
    if ((int64_t)gy * stride > COL_SLOT_FLOATS) gy = COL_SLOT_FLOATS / stride;
    if (gy < 1) return RL_ERR_ARG;
    hipLaunchKernelGGL((col_reduce_kernel<F>), dim3(gx, gy), dim3(256), 0, st, f, rows, C, tpr_shift, out0, out1, rb, slots, stride);
    const int n = out1 ? 2 * C : C;
    hipLaunchKernelGGL(col_fold_kernel, dim3((n + 31) / 32), dim3(256), 0, st, slots, stride, gy, n, out0, out1, C, 1, 0, alpha);     // OVERWRITES out0 / out1
  } else {
    if (alpha != 1.0f) return RL_ERR_ARG;        // the scaled form exists on the record + fold path only
    hipLaunchKernelGGL((col_reduce_kernel<F>), dim3(gx, gy), dim3(256), 0, st, f, rows, C, tpr_shift, out0, out1, rb, (float*)nullptr, 0);
  }
  return RL_LAUNCH_CHECK();
}

// ---- bf16 fast path of the column reductions: 16-byte loads, C a power of two in [8, 2048] ----------------------------------
// One workgroup spans all C columns (C / 8 threads per row, 2048 / C rows per pass) over a contiguous chunk of rows.  The four row
// passes of an iteration are LOADED before the first is added (4 x 16 B per input tensor in flight per thread, against 4 x 8 B and
// a 64-bit row * ld multiply per load in the generic kernel), row weights come from a shift, and the per-column constants (mean,
// rstd) sit in registers for the whole loop.  Partial sums go to the chunk's record and the same ordered fold as above: bitwise
// reproducible.  The 64-channel reductions over the 957k rows of block 1 went from 1.1-1.6 TB/s to the figures in
// profiles/round3_bn_probe.log.
struct RowW16 {                       // multiplicity of a row's glyph (dedup), hw a power of two
  const float* counts; int hw_shift;
  __device__ __forceinline__ float operator()(int r) const { return counts ? counts[r >> hw_shift] : 1.0f; }
};
template <int N> struct Acc16 { floatx4 lo[N], hi[N]; };

struct WSum16F {                      // sum_r w_r x[r][c]
  static constexpr int NACC = 1;
  const bf16_t* x; RowW16 w;
  struct Item { uint4 x; };
  struct Ctx {};
  __device__ __forceinline__ Ctx prep(int) const { return Ctx{}; }
  __device__ __forceinline__ Item load(int64_t e) const { return Item{*(const uint4*)(x + e)}; }
  __device__ __forceinline__ void add(Acc16<1>& a, const Item& it, const Ctx&, int r) const {
    floatx4 lo, hi; unpack8(it.x, lo, hi);
    const float ww = w(r);
    a.lo[0] += lo * ww; a.hi[0] += hi * ww;
  }
};
struct SumSqC16F {                    // sum_r w_r (x[r][c] - mean[c])^2
  static constexpr int NACC = 1;
  const bf16_t* x; const float* mean; RowW16 w;
  struct Item { uint4 x; };
  struct Ctx { floatx4 mlo, mhi; };
  __device__ __forceinline__ Ctx prep(int col) const { return Ctx{*(const floatx4*)(mean + col), *(const floatx4*)(mean + col + 4)}; }
  __device__ __forceinline__ Item load(int64_t e) const { return Item{*(const uint4*)(x + e)}; }
  __device__ __forceinline__ void add(Acc16<1>& a, const Item& it, const Ctx& c, int r) const {
    floatx4 lo, hi; unpack8(it.x, lo, hi);
    lo -= c.mlo; hi -= c.mhi;
    const float ww = w(r);
    a.lo[0] += lo * lo * ww; a.hi[0] += hi * hi * ww;
  }
};
// BatchNorm backward sums of NB normalisations that share the incoming gradient and ReLU mask (bn2 and the shortcut's BN both feed
// out = relu(bn2(c2) + bns(cs)), char_cnn.py:30-32): acc 0 = sum g, acc 1 + b = sum g * xhat_b, with dy and the mask read ONCE.
template <int NB> struct BnBwd16F {
  static constexpr int NACC = 1 + NB;
  const bf16_t* dy; const bf16_t* relu_src; const bf16_t* x[NB]; const float* mean[NB]; const float* rstd[NB];
  struct Item { uint4 g, o, x[NB]; };
  struct Ctx { floatx4 mlo[NB], mhi[NB], rlo[NB], rhi[NB]; };
  __device__ __forceinline__ Ctx prep(int col) const {
    Ctx c;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      c.mlo[b] = *(const floatx4*)(mean[b] + col); c.mhi[b] = *(const floatx4*)(mean[b] + col + 4);
      c.rlo[b] = *(const floatx4*)(rstd[b] + col); c.rhi[b] = *(const floatx4*)(rstd[b] + col + 4);
    }
    return c;
  }
  __device__ __forceinline__ Item load(int64_t e) const {
    Item it;
    it.g = *(const uint4*)(dy + e);
    it.o = relu_src ? *(const uint4*)(relu_src + e) : uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
    for (int b = 0; b < NB; ++b) it.x[b] = *(const uint4*)(x[b] + e);
    return it;
  }
  __device__ __forceinline__ void add(Acc16<NACC>& a, const Item& it, const Ctx& c, int) const {
    floatx4 glo, ghi, olo, ohi;
    unpack8(it.g, glo, ghi); unpack8(it.o, olo, ohi);
#pragma unroll
    for (int j = 0; j < 4; ++j) { glo[j] = olo[j] > 0.f ? glo[j] : 0.f; ghi[j] = ohi[j] > 0.f ? ghi[j] : 0.f; }
    a.lo[0] += glo; a.hi[0] += ghi;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      floatx4 xlo, xhi; unpack8(it.x[b], xlo, xhi);
      a.lo[1 + b] += glo * ((xlo - c.mlo[b]) * c.rlo[b]);
      a.hi[1 + b] += ghi * ((xhi - c.mhi[b]) * c.rhi[b]);
    }
  }
};

// One-pass BatchNorm training statistics: S1 = sum w (x - K), S2 = sum w (x - K)^2 about the per-channel pivot K = the first row of
// the batch itself (round 3 pivoted on the running mean: fresh or foreign running statistics far from the batch mean made the
// subtraction below cancel - ADVICE round 3).  mean = K + S1 / n, var = S2 / n - (S1 / n)^2: with fp32 sums the cancellation error
// is ~1e-7 (1 + (mean - K)^2 / var) relative, far inside bf16 activations' own rounding; the fp32 parity mode keeps the two-pass form.
struct Moments16F {
  static constexpr int NACC = 2;
  const bf16_t* x; RowW16 w;
  struct Item { uint4 x; };
  struct Ctx { floatx4 klo, khi; };
  // pivot K = the batch's own first row (a sample of the distribution being measured: |mean - K| is of the order of the standard
  // deviation whatever the running statistics hold - fresh buffers, buffers loaded from another domain)
  __device__ __forceinline__ Ctx prep(int col) const { Ctx c; unpack8(*(const uint4*)(x + col), c.klo, c.khi); return c; }
  __device__ __forceinline__ Item load(int64_t e) const { return Item{*(const uint4*)(x + e)}; }
  __device__ __forceinline__ void add(Acc16<2>& a, const Item& it, const Ctx& c, int r) const {
    floatx4 lo, hi; unpack8(it.x, lo, hi);
    lo -= c.klo; hi -= c.khi;
    const float ww = w(r);
    const floatx4 wlo = lo * ww, whi = hi * ww;
    a.lo[0] += wlo; a.hi[0] += whi;
    a.lo[1] += wlo * lo; a.hi[1] += whi * hi;
  }
};
// Ordered fold of the [S1 | S2] records (same lane / tree order as col_fold_kernel) + everything bn_train_kernel does, one workgroup
// per 32 channels: mean, rstd, scale, shift, running statistics (unbiased variance, momentum), num_batches_tracked.
__global__ void __launch_bounds__(256) bn_fold_train_kernel(const float* __restrict__ slots, int slot_stride, int nrec, int C, float inv_n, float unbias,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                                             float* rmean, float* rvar, float* mean, float* rstd, float* scale, float* shift, int64_t* nbt,
                                                             const bf16_t* __restrict__ x0) {
  __shared__ floatx4 red[2][32][8];
  const int cq = threadIdx.x & 7, kl = threadIdx.x >> 3;
  const int c = (blockIdx.x * 8 + cq) * 4;
  floatx4 s1 = floatx4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
  if (c < C)
    for (int k = kl; k < nrec; k += 32) {
      const float* rec = slots + (int64_t)k * slot_stride;
      s1 += *(const floatx4*)(rec + c);
      s2 += *(const floatx4*)(rec + C + c);
    }
  red[0][kl][cq] = s1; red[1][kl][cq] = s2;
  __syncthreads();
#pragma unroll
  for (int w = 16; w > 0; w >>= 1) {
    if (kl < w) { red[0][kl][cq] += red[0][kl + w][cq]; red[1][kl][cq] += red[1][kl + w][cq]; }
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) *nbt += 1;
  if (kl == 0 && c < C) {
    s1 = red[0][0][cq]; s2 = red[1][0][cq];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = s1[j] * inv_n, m = to_f<bf16_t>(x0[c + j]) + d;   // the pivot was the batch's first row (Moments16F::prep)
      const float var = fmaxf(s2[j] * inv_n - d * d, 0.f);
      const float rs = 1.0f / sqrtf(var + eps), g = gamma[c + j];
      mean[c + j] = m; rstd[c + j] = rs; scale[c + j] = g * rs; shift[c + j] = beta[c + j] - m * g * rs;
      rmean[c + j] = (1.0f - momentum) * rmean[c + j] + momentum * m;
      rvar[c + j] = (1.0f - momentum) * rvar[c + j] + momentum * (var * unbias);
    }
  }
}
template <typename F>
__global__ void __launch_bounds__(256) col_reduce16_kernel(F f, int rows_max, int c_shift, RowBound rb, float* __restrict__ slots, int slot_stride,
                                                            int dup0_at) {
  __shared__ floatx4 red[2][256];
  const int tshift = c_shift - 3, tpr = 1 << tshift, rpp = 256 >> tshift;
  const int cx = threadIdx.x & (tpr - 1), ry = threadIdx.x >> tshift, col = cx * 8;
  const int rows = rb_rows(rb, rows_max);
  const int chunk = (rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * chunk, r1 = min(rows, r0 + chunk);
  Acc16<F::NACC> acc;
#pragma unroll
  for (int k = 0; k < F::NACC; ++k) acc.lo[k] = acc.hi[k] = floatx4{0.f, 0.f, 0.f, 0.f};
  const typename F::Ctx ctx = f.prep(col);
  int r = r0 + ry;
  const int64_t step = (int64_t)rpp << c_shift;
  int64_t e = ((int64_t)r << c_shift) + col;
  for (; r + 3 * rpp < r1; r += 4 * rpp, e += 4 * step) {
    const typename F::Item i0 = f.load(e), i1 = f.load(e + step), i2 = f.load(e + 2 * step), i3 = f.load(e + 3 * step);
    f.add(acc, i0, ctx, r); f.add(acc, i1, ctx, r + rpp); f.add(acc, i2, ctx, r + 2 * rpp); f.add(acc, i3, ctx, r + 3 * rpp);
  }
  for (; r < r1; r += rpp, e += step) { const typename F::Item i0 = f.load(e); f.add(acc, i0, ctx, r); }
  float* rec = slots + (int64_t)blockIdx.x * slot_stride;
  const int C = 1 << c_shift;
#pragma unroll
  for (int k = 0; k < F::NACC; ++k) {
    red[0][threadIdx.x] = acc.lo[k];
    red[1][threadIdx.x] = acc.hi[k];
    __syncthreads();
    if (ry == 0) {
      floatx4 lo = acc.lo[k], hi = acc.hi[k];
      for (int j = 1; j < rpp; ++j) { lo += red[0][j * tpr + cx]; hi += red[1][j * tpr + cx]; }
      // record layout: accumulator k at [k * C, (k + 1) * C); with dup0_at > 0 (two normalisations) accumulator 0 is stored again at
      // dup0_at * C and accumulators >= dup0_at move up one slot, so that the folded output reads [sum g | sum g xhat_0 | sum g | sum g xhat_1]
      const int slot = (dup0_at > 0 && k >= dup0_at) ? k + 1 : k;
      *(floatx4*)(rec + slot * C + col) = lo; *(floatx4*)(rec + slot * C + col + 4) = hi;
      if (k == 0 && dup0_at > 0) { *(floatx4*)(rec + dup0_at * C + col) = lo; *(floatx4*)(rec + dup0_at * C + col + 4) = hi; }
    }
    __syncthreads();
  }
}
static inline int pow2_shift(int C) { int s = 0; while ((1 << s) < C) ++s; return (1 << s) == C ? s : -1; }
// fast path applies: bf16, C a power of two in [8, 2048], hw a power of two, ordered-fold scratch present
static inline bool col16_ok(int C, const RowBound& rb) {
  const int cs = pow2_shift(C);
  return g_bn_fast && cs >= 3 && cs <= 11 && pow2_shift(rb.hw) >= 0 && rb.slots != nullptr;
}
// nout = folded outputs of C floats each, written contiguously to `out` (overwritten); alpha scales them
template <typename F>
static int launch_col_reduce16(hipStream_t st, const F& f, int rows, int C, float* out, int nout, int dup0_at, const RowBound& rb, float alpha) {
  if (rows <= 0) return RL_OK;
  const int cs = pow2_shift(C), rpp = 2048 >> cs;
  const int stride = nout * C;
  int g = g_bn_chunks;
  const int max_g = (rows + 4 * rpp - 1) / (4 * rpp);
  if (g > max_g) g = max_g;
  if ((int64_t)g * stride > COL_SLOT_FLOATS) g = COL_SLOT_FLOATS / stride;
  if (g < 1) g = 1;
  hipLaunchKernelGGL((col_reduce16_kernel<F>), dim3(g), dim3(256), 0, st, f, rows, cs, rb, rb.slots, stride, dup0_at);
  hipLaunchKernelGGL(col_fold_kernel, dim3((stride + 31) / 32), dim3(256), 0, st, rb.slots, stride, g, stride, out, out + C, C, 1, 0, alpha);
  return RL_LAUNCH_CHECK();
}

template <typename T> struct SumF {
  const T* x; int64_t ld;
  __device__ __forceinline__ void operator()(int r, int c, floatx4& a0, floatx4&) const { a0 += load4<T>(x + (int64_t)r * ld + c); }
};
template <typename T> int bias_grad(hipStream_t st, const T* dy, int64_t ld, int rows, int N, float* out, const int* rows_dev) {
  if (ld & 3) return RL_ERR_ARG;
  SumF<T> f{dy, ld};
  RowBound rb;
  rb.rows_dev = rows_dev;
  return launch_col_reduce(st, f, rows, N, out, nullptr, rb);
}
template int bias_grad<bf16_t>(hipStream_t, const bf16_t*, int64_t, int, int, float*, const int*);
template int bias_grad<float>(hipStream_t, const float*, int64_t, int, int, float*, const int*);

template <typename T> struct WSumF {
  const T* x; int64_t ld; RowBound rb;
  __device__ __forceinline__ void operator()(int r, int c, floatx4& a0, floatx4&) const {
    a0 += load4<T>(x + (int64_t)r * ld + c) * rb_weight(rb, r);
  }
};
template <typename T> int col_sum(hipStream_t st, const T* x, int P, int C, float* out, RowBound rb, float scale) {
  if constexpr (sizeof(T) == 2) {
    if (col16_ok(C, rb)) { WSum16F f{x, RowW16{rb.counts, pow2_shift(rb.hw)}}; return launch_col_reduce16(st, f, P, C, out, 1, 0, rb, scale); }
  }
  WSumF<T> f{x, (int64_t)C, rb};
  return launch_col_reduce(st, f, P, C, out, nullptr, rb, rb.slots, scale);
}
template int col_sum<bf16_t>(hipStream_t, const bf16_t*, int, int, float*, RowBound, float);
template int col_sum<float>(hipStream_t, const float*, int, int, float*, RowBound, float);

template <typename T> struct SumSqCF {
  const T* x; int64_t ld; const float* mean; RowBound rb;
  __device__ __forceinline__ void operator()(int r, int c, floatx4& a0, floatx4&) const {
    const floatx4 d = load4<T>(x + (int64_t)r * ld + c) - *(const floatx4*)(mean + c);
    a0 += d * d * rb_weight(rb, r);
  }
};
template <typename T> int col_sumsq_centered(hipStream_t st, const T* x, int P, int C, const float* mean, float* out, RowBound rb) {
  if constexpr (sizeof(T) == 2) {
    if (col16_ok(C, rb)) { SumSqC16F f{x, mean, RowW16{rb.counts, pow2_shift(rb.hw)}}; return launch_col_reduce16(st, f, P, C, out, 1, 0, rb, 1.0f); }
  }
  SumSqCF<T> f{x, (int64_t)C, mean, rb};
  return launch_col_reduce(st, f, P, C, out, nullptr, rb, rb.slots);
}
template int col_sumsq_centered<bf16_t>(hipStream_t, const bf16_t*, int, int, const float*, float*, RowBound);
template int col_sumsq_centered<float>(hipStream_t, const float*, int, int, const float*, float*, RowBound);

template <typename T> struct BnBwdF {
  const T* dy; const T* relu_src; const T* x; const float* mean; const float* rstd; int64_t ld;
  __device__ __forceinline__ void operator()(int r, int c, floatx4& a0, floatx4& a1) const {
    floatx4 g = load4<T>(dy + (int64_t)r * ld + c);
    if (relu_src != nullptr) {
      const floatx4 o = load4<T>(relu_src + (int64_t)r * ld + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
    }
    const floatx4 xh = (load4<T>(x + (int64_t)r * ld + c) - *(const floatx4*)(mean + c)) * *(const floatx4*)(rstd + c);
    a0 += g;
    a1 += g * xh;
  }
};
template <typename T>
int bn_bwd_reduce(hipStream_t st, const T* dy, const T* relu_src, const T* x, const float* mean, const float* rstd, int P, int C,
                  float* sums, RowBound rb) {
  if constexpr (sizeof(T) == 2) {
    if (col16_ok(C, rb)) { BnBwd16F<1> f{dy, relu_src, {x}, {mean}, {rstd}}; return launch_col_reduce16(st, f, P, C, sums, 2, 0, rb, 1.0f); }
  }
  BnBwdF<T> f{dy, relu_src, x, mean, rstd, (int64_t)C};
  return launch_col_reduce(st, f, P, C, sums, sums + C, rb, rb.slots);
}
int bn_stats_train16(hipStream_t st, const bf16_t* x, int P, int C, int n_stat, const float* gamma, const float* beta, float eps, float momentum,
                     float* rmean, float* rvar, float* mean, float* rstd, float* scale, float* shift, int64_t* nbt, RowBound rb) {
  if (!col16_ok(C, rb) || !g_bn_onepass || P <= 0) return RL_ERR_ARG;
  const int cs = pow2_shift(C), rpp = 2048 >> cs, stride = 2 * C;
  int g = g_bn_chunks;
  const int max_g = (P + 4 * rpp - 1) / (4 * rpp);
  if (g > max_g) g = max_g;
  if ((int64_t)g * stride > COL_SLOT_FLOATS) g = COL_SLOT_FLOATS / stride;
  if (g < 1) g = 1;
  Moments16F f{x, RowW16{rb.counts, pow2_shift(rb.hw)}};
  hipLaunchKernelGGL((col_reduce16_kernel<Moments16F>), dim3(g), dim3(256), 0, st, f, P, cs, rb, rb.slots, stride, 0);
  const int n = n_stat > 0 ? n_stat : P;
  hipLaunchKernelGGL(bn_fold_train_kernel, dim3((C + 31) / 32), dim3(256), 0, st, rb.slots, stride, g, C, 1.0f / (float)n,
                     (float)n / (float)(n > 1 ? n - 1 : 1), gamma, beta, eps, momentum, rmean, rvar, mean, rstd, scale, shift, nbt, x);
  return RL_LAUNCH_CHECK();
}
// Two normalisations sharing dy and the ReLU mask: sums = [sum g | sum g xhat_a | sum g | sum g xhat_b] (4C floats).  Returns
// RL_ERR_ARG when the fast path does not apply (the caller then runs the two single reductions).
int bn_bwd_reduce2(hipStream_t st, const bf16_t* dy, const bf16_t* relu_src, const bf16_t* xa, const float* mean_a, const float* rstd_a,
                   const bf16_t* xb, const float* mean_b, const float* rstd_b, int P, int C, float* sums, RowBound rb) {
  if (!col16_ok(C, rb)) return RL_ERR_ARG;
  BnBwd16F<2> f{dy, relu_src, {xa, xb}, {mean_a, mean_b}, {rstd_a, rstd_b}};
  return launch_col_reduce16(st, f, P, C, sums, 4, 2, rb, 1.0f);
}
template int bn_bwd_reduce<bf16_t>(hipStream_t, const bf16_t*, const bf16_t*, const bf16_t*, const float*, const float*, int, int, float*, RowBound);
template int bn_bwd_reduce<float>(hipStream_t, const float*, const float*, const float*, const float*, const float*, int, int, float*, RowBound);

// ---------------------------------------------------------------------------------------------
// Dropout as a stand-alone map (the one before the classifier, models.py:858)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, DropParams d) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  floatx4 v = load4<T>(x + i);
  v *= drop_mult4(d.seed, d.thresh, d.scale, (uint32_t)i);
  store4<T>(y + i, v);
}
template <typename T> int dropout_apply(hipStream_t st, const T* x, T* y, int rows, int H, DropParams d) {
  const int64_t n = (int64_t)rows * H;
  if (n & 3) return RL_ERR_ARG;
  hipLaunchKernelGGL((dropout_kernel<T>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, x, y, n, d);
  return RL_LAUNCH_CHECK();
}
template int dropout_apply<bf16_t>(hipStream_t, const bf16_t*, bf16_t*, int, int, DropParams);
template int dropout_apply<float>(hipStream_t, const float*, float*, int, int, DropParams);

// ---------------------------------------------------------------------------------------------
// Masked cross-entropy (models.py:862-869): one 256-thread workgroup per token row.
// ---------------------------------------------------------------------------------------------
// rows that enter the loss: loss_mask == 1 (models.py:864) and label != -100 (CrossEntropyLoss's default ignore_index)
#define RL_CE_IGNORE_INDEX (-100)
__global__ void count_active_kernel(const int64_t* __restrict__ m, const int64_t* __restrict__ labels, int n, float* out) {
  float c = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    c += (m[i] == 1 && labels[i] != RL_CE_IGNORE_INDEX) ? 1.f : 0.f;
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0 && c != 0.f) atomicAdd(out, c);
}

__device__ __forceinline__ float block_reduce(float v, float* sm, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[wave] = v;
  __syncthreads();
  float r = sm[0];
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, sm[w]) : r + sm[w];
  return r;
}

// row_liveness (ops.h): one workgroup.  Phase 1: last flagged position of every sentence; phase 2: row bytes; phase 3: the 64-,
// 32- and 16-row block lists by ballot + prefix scan (ascending, deterministic).
__global__ void __launch_bounds__(1024) row_liveness_kernel(const int64_t* __restrict__ masks, const int64_t* __restrict__ loss_masks, int B, int S,
                                                             uint8_t* __restrict__ row_live, int* __restrict__ tiles64, int* __restrict__ tiles32,
                                                             int* __restrict__ tiles16, int* __restrict__ n_tiles, int* __restrict__ rlen, int* __restrict__ rows) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int T = B * S, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // phases 1 + 2: a wave per sentence finds 1 + the last flagged position, then writes the sentence's row bytes
  for (int b = wave; b < B; b += 16) {
    int last = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      const bool f = s < S && (masks[(int64_t)b * S + s] == 1 || (loss_masks != nullptr && loss_masks[(int64_t)b * S + s] == 1));
      const unsigned long long bal = __ballot(f);
      if (bal != 0ull) last = s0 + 64 - __clzll(bal);
    }
    for (int s = lane; s < S; s += 64) row_live[(int64_t)b * S + s] = s < last ? 1 : 0;
    if (lane == 0 && rlen != nullptr) rlen[b] = last;
  }
  __syncthreads();
  // phase 3: block lists
  for (int pass = 0; pass < 4; ++pass) {          // (pass 3: the live ROWS themselves, n_tiles[3] of them - the row-granular GEMM list)
    const int bp = pass == 3 ? 1 : 64 >> pass;
    int* out = pass == 0 ? tiles64 : (pass == 1 ? tiles32 : (pass == 2 ? tiles16 : rows));
    if (out == nullptr) continue;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    const int nblk = T / bp;
    for (int k0 = 0; k0 < nblk; k0 += 1024) {
      const int k = k0 + threadIdx.x;
      bool live = false;
      if (k < nblk && bp == 1) live = row_live[k] != 0;
      else if (k < nblk)
        for (int r = 0; r < bp; r += 16) {                       // 16 row bytes at a time
          const uint4 v = *(const uint4*)(row_live + (int64_t)k * bp + r);
          live = live || (v.x | v.y | v.z | v.w) != 0u;
        }
      const unsigned long long bal = __ballot(live);
      const int before = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wsum[wave] = __popcll(bal);
      __syncthreads();
      int off = base_s;
      for (int w = 0; w < wave; ++w) off += wsum[w];
      if (live) out[off + before] = k;
      __syncthreads();
      if (threadIdx.x == 0) { int t = base_s; for (int w = 0; w < 16; ++w) t += wsum[w]; base_s = t; }
      __syncthreads();
    }
    if (threadIdx.x == 0) n_tiles[pass] = base_s;
    __syncthreads();
  }
}
int row_liveness(hipStream_t st, const int64_t* masks, const int64_t* loss_masks, int B, int S, uint8_t* row_live, int* tiles64, int* tiles32,
                 int* tiles16, int* n_tiles, int* rlen, int* rows) {
  if (B < 1 || S < 1 || ((int64_t)B * S) % 64 != 0 || masks == nullptr) return RL_ERR_ARG;
  hipLaunchKernelGGL(row_liveness_kernel, dim3(1), dim3(1024), 0, st, masks, loss_masks, B, S, row_live, tiles64, tiles32, tiles16, n_tiles, rlen, rows);
  return RL_LAUNCH_CHECK();
}

// Rows that enter the loss, in order: act_idx[j] = token row of the j-th active row, inv[row] = j (or -1), *n_act and *count = how
// many, row_loss[row] = 0 for the others.  One workgroup of 1024 threads (rows <= 65536), ballot + wave-prefix scan: deterministic.
__global__ void __launch_bounds__(1024) active_rows_kernel(const int64_t* __restrict__ m, const int64_t* __restrict__ labels, int n,
                                                            int* __restrict__ act_idx, int* __restrict__ inv, int* __restrict__ n_act,
                                                            float* __restrict__ count, float* __restrict__ row_loss) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (int r0 = 0; r0 < n; r0 += 1024) {
    const int r = r0 + threadIdx.x;
    const bool a = r < n && m[r] == 1 && labels[r] != RL_CE_IGNORE_INDEX;
    const unsigned long long b = __ballot(a);
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(b);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (r < n) {
      if (a) { act_idx[off + before] = r; inv[r] = off + before; }
      else { inv[r] = -1; if (row_loss != nullptr) row_loss[r] = 0.f; }
    }
    __syncthreads();
    if (threadIdx.x == 0) { int t = base_s; for (int w = 0; w < 16; ++w) t += wsum[w]; base_s = t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { *n_act = base_s; *count = (float)base_s; }
}

template <typename T>
__global__ void __launch_bounds__(256)
ce_kernel(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels, const int64_t* __restrict__ loss_mask,
          int V, float* loss_out, const float* __restrict__ count, T* __restrict__ dlogits, float* __restrict__ row_loss, int64_t ld_dl,
          const int* __restrict__ act_idx, const int* __restrict__ n_act, int logits_compact) {
  __shared__ float sm[4];
  // compacted form (act_idx != nullptr): workgroup j works on the j-th ACTIVE row and writes gradient row j; the rows outside the loss
  // are not visited at all (their loss terms were zeroed by active_rows_kernel, their gradient rows do not exist)
  if (act_idx != nullptr && (int)blockIdx.x >= *n_act) return;
  const int row = act_idx != nullptr ? act_idx[blockIdx.x] : (int)blockIdx.x;
  const T* x = logits + (int64_t)(logits_compact ? (int)blockIdx.x : row) * ld;
  T* dx = dlogits ? dlogits + (int64_t)(act_idx != nullptr ? (int)blockIdx.x : row) * ld_dl : nullptr;
  // a padded gradient row (ld_dl > V: 128-byte aligned rows for the GEMMs that consume it) keeps exact zeros in its tail
  if (dx != nullptr)
    for (int c = V + threadIdx.x * 4; c < ld_dl; c += 1024) store4<T>(dx + c, floatx4{0.f, 0.f, 0.f, 0.f});
  const int64_t lab64 = labels[row];
  const bool active = loss_mask[row] == 1 && lab64 != RL_CE_IGNORE_INDEX;
  if (!active) {
    if (row_loss != nullptr && threadIdx.x == 0) row_loss[row] = 0.f;
    if (dx != nullptr)
      for (int c = threadIdx.x * 4; c < V; c += 1024) store4<T>(dx + c, floatx4{0.f, 0.f, 0.f, 0.f});
    return;
  }
  float mx = -3.0e38f;
  for (int c = threadIdx.x * 4; c < V; c += 1024) {
    const floatx4 v = load4<T>(x + c);
    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  mx = block_reduce(mx, sm, true);
  float s = 0.f;
  for (int c = threadIdx.x * 4; c < V; c += 1024) {
    const floatx4 v = load4<T>(x + c);
    s += exp_t<T>(v[0] - mx) + exp_t<T>(v[1] - mx) + exp_t<T>(v[2] - mx) + exp_t<T>(v[3] - mx);
  }
  s = block_reduce(s, sm, false);
  const float inv_n = 1.0f / count[0];
  // a label outside [0, V) is an error PyTorch raises on; here it poisons the loss (NaN) instead of reading out of bounds
  const bool lab_ok = lab64 >= 0 && lab64 < V;
  const int lab = lab_ok ? (int)lab64 : 0;
  const float lse = mx + logf(s);
  if (threadIdx.x == 0) {
    const float l = lab_ok ? (lse - to_f<T>(x[lab])) * inv_n : __builtin_nanf("");
    if (row_loss != nullptr) row_loss[row] = l;          // deterministic path: per-row terms, folded in a fixed order below
    else atomicAdd(loss_out, l);
  }
  if (dx != nullptr) {
    const float inv_s = 1.0f / s;
    for (int c = threadIdx.x * 4; c < V; c += 1024) {
      const floatx4 v = load4<T>(x + c);
      floatx4 d;
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = (exp_t<T>(v[j] - mx) * inv_s - ((c + j) == lab ? 1.0f : 0.0f)) * inv_n;
      store4<T>(dx + c, d);
    }
  }
}
// bf16 fast path of the masked cross-entropy (V % 8 == 0, V <= 256 * 8 * NCH, 16-byte aligned rows): the row - 21128 logits = 42 KB -
// is read ONCE into registers (NCH 16-byte chunks per thread, all loads in flight together) and the maximum, the sum of exponentials
// and the gradient row come from the registers; the generic kernel walks the row three times with 8-byte loads and, with thousands of
// rows in flight, every walk comes from HBM (PMC round 3: 659 MB read per launch for 207 MB of active rows).
template <int NCH>
__global__ void __launch_bounds__(256)
ce_row16_kernel(const bf16_t* logits, int64_t ld, const int64_t* __restrict__ labels, const int64_t* __restrict__ loss_mask,
                int V, float* loss_out, const float* __restrict__ count, bf16_t* dlogits, float* __restrict__ row_loss, int64_t ld_dl,
                const int* __restrict__ act_idx, const int* __restrict__ n_act, int logits_compact) {
  __shared__ float sm[4];
  if (act_idx != nullptr && (int)blockIdx.x >= *n_act) return;
  const int row = act_idx != nullptr ? act_idx[blockIdx.x] : (int)blockIdx.x;
  // logits_compact (round 6, K13): the logits exist for the loss rows only, row j of `logits` = the j-th active row - possibly the SAME
  // buffer as dlogits: the row is read whole into registers before its gradient row is stored over it
  const bf16_t* x = logits + (int64_t)(logits_compact ? (int)blockIdx.x : row) * ld;
  bf16_t* dx = dlogits ? dlogits + (int64_t)(act_idx != nullptr ? (int)blockIdx.x : row) * ld_dl : nullptr;
  if (dx != nullptr)
    for (int c = V + threadIdx.x * 4; c < ld_dl; c += 1024) store4<bf16_t>(dx + c, floatx4{0.f, 0.f, 0.f, 0.f});
  const int64_t lab64 = labels[row];
  const bool active = loss_mask[row] == 1 && lab64 != RL_CE_IGNORE_INDEX;
  const int nch = V >> 3;
  if (!active) {
    if (row_loss != nullptr && threadIdx.x == 0) row_loss[row] = 0.f;
    if (dx != nullptr)
      for (int k = threadIdx.x; k < nch; k += 256) *(uint4*)(dx + 8 * k) = uint4{0u, 0u, 0u, 0u};
    return;
  }
  uint4 r[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int k = threadIdx.x + 256 * i;
    r[i] = k < nch ? *(const uint4*)(x + 8 * k) : uint4{0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u};      // -inf pairs: exp() = 0
  }
  // the label's logit is read here, with the row, and pinned behind its wait: in the in-place form (logits == dlogits) other waves
  // store gradient chunks over the row as soon as they are past the two reductions below
  const bool lab_ok0 = lab64 >= 0 && lab64 < V;
  float x_lab = to_f<bf16_t>(x[lab_ok0 ? (int)lab64 : 0]);
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x_lab));
#endif
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    float v[8];
    unpack8(r[i], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, v[j]);
  }
  mx = block_reduce(mx, sm, true);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    float v[8];
    unpack8(r[i], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += exp_t<bf16_t>(v[j] - mx);
  }
  s = block_reduce(s, sm, false);
  const float inv_n = 1.0f / count[0];
  const bool lab_ok = lab64 >= 0 && lab64 < V;
  const int lab = lab_ok ? (int)lab64 : 0;
  const float lse = mx + logf(s);
  if (threadIdx.x == 0) {
    const float l = lab_ok ? (lse - x_lab) * inv_n : __builtin_nanf("");
    if (row_loss != nullptr) row_loss[row] = l;
    else atomicAdd(loss_out, l);
  }
  if (dx != nullptr) {
    const float inv_s = 1.0f / s;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int k = threadIdx.x + 256 * i;
      if (k < nch) {
        float v[8], d[8];
        unpack8(r[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = (exp_t<bf16_t>(v[j] - mx) * inv_s - ((8 * k + j) == lab ? 1.0f : 0.0f)) * inv_n;
        *(uint4*)(dx + 8 * k) = pack8(d);
      }
    }
  }
}
template <typename T>
static bool ce_row16_launch(hipStream_t, const T*, int64_t, const int64_t*, const int64_t*, int, int, float*, float*, T*, float*, int64_t, const int*, const int*, int) { return false; }
template <>
bool ce_row16_launch<bf16_t>(hipStream_t st, const bf16_t* logits, int64_t ld, const int64_t* labels, const int64_t* loss_mask, int rows, int V,
                             float* loss_out, float* count_buf, bf16_t* dlogits, float* row_loss, int64_t ld_dl, const int* act_idx, const int* n_act, int compact) {
  constexpr int NCH = 11;          // 256 threads x 11 chunks x 8 = 22528 >= 21128
  if (!g_ce_fast || (V & 7) || (ld & 7) || (ld_dl & 7) || V > 256 * 8 * NCH || ((uintptr_t)logits & 15) || ((uintptr_t)dlogits & 15)) return false;
  hipLaunchKernelGGL((ce_row16_kernel<NCH>), dim3(rows), dim3(256), 0, st, logits, ld, labels, loss_mask, V, loss_out, count_buf, dlogits, row_loss, ld_dl,
                     act_idx, n_act, compact);
  return true;
}

// loss = sum of the per-row terms in a fixed order (256 strided partial sums, then a fixed tree): bitwise reproducible
__global__ void __launch_bounds__(256) ce_fold_kernel(const float* __restrict__ row_loss, int rows, float* loss_out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < rows; i += 256) s += row_loss[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss_out = red[0];
}
template <typename T>
int ce_loss(hipStream_t st, const T* logits, int64_t ld, const int64_t* labels, const int64_t* loss_mask, int rows, int V,
            float* loss_out, float* count_buf, T* dlogits, float* row_loss, int64_t ld_dl, const CeCompact& cc) {
  if (ld_dl <= 0) ld_dl = ld;
  if ((V & 3) || (ld & 3) || (ld_dl & 3) || ld_dl < V) return RL_ERR_ARG;
  if (cc.act_idx != nullptr) {       // compacted gradient rows (needs the per-row loss terms: the ordered fold is the only sum)
    if (row_loss == nullptr || cc.inv == nullptr || cc.n_act == nullptr || rows > 65536) return RL_ERR_ARG;
    // phases: 1 = list the active rows, 2 = loss + gradient rows, 0 = both (cc.logits_compact needs the list before the logits exist:
    // the caller runs phase 1, the classifier over the listed rows, then phase 2)
    if (cc.phase != 2)
      hipLaunchKernelGGL(active_rows_kernel, dim3(1), dim3(1024), 0, st, loss_mask, labels, rows, cc.act_idx, cc.inv, cc.n_act, count_buf, row_loss);
    if (cc.phase == 1) return RL_LAUNCH_CHECK();
    const bool fast = ce_row16_launch<T>(st, logits, ld, labels, loss_mask, rows, V, loss_out, count_buf, dlogits, row_loss, ld_dl, (const int*)cc.act_idx,
                                         (const int*)cc.n_act, cc.logits_compact);
    if (!fast) {
      if (cc.logits_compact && (const void*)logits == (const void*)dlogits) return RL_ERR_ARG;      // (the generic kernel walks a row three times: not in place)
      hipLaunchKernelGGL((ce_kernel<T>), dim3(rows), dim3(256), 0, st, logits, ld, labels, loss_mask, V, loss_out, count_buf, dlogits, row_loss, ld_dl,
                         (const int*)cc.act_idx, (const int*)cc.n_act, cc.logits_compact);
    }
    hipLaunchKernelGGL(ce_fold_kernel, dim3(1), dim3(256), 0, st, row_loss, rows, loss_out);
    return RL_LAUNCH_CHECK();
  }
  (void)hipMemsetAsync(loss_out, 0, sizeof(float), st);
  (void)hipMemsetAsync(count_buf, 0, sizeof(float), st);
  hipLaunchKernelGGL(count_active_kernel, dim3(64), dim3(256), 0, st, loss_mask, labels, rows, count_buf);
  if (!ce_row16_launch<T>(st, logits, ld, labels, loss_mask, rows, V, loss_out, count_buf, dlogits, row_loss, ld_dl, nullptr, nullptr, 0))
    hipLaunchKernelGGL((ce_kernel<T>), dim3(rows), dim3(256), 0, st, logits, ld, labels, loss_mask, V, loss_out, count_buf, dlogits, row_loss, ld_dl,
                       (const int*)nullptr, (const int*)nullptr, 0);
  if (row_loss != nullptr) hipLaunchKernelGGL(ce_fold_kernel, dim3(1), dim3(256), 0, st, row_loss, rows, loss_out);
  return RL_LAUNCH_CHECK();
}
template int ce_loss<bf16_t>(hipStream_t, const bf16_t*, int64_t, const int64_t*, const int64_t*, int, int, float*, float*, bf16_t*, float*, int64_t, const CeCompact&);
template int ce_loss<float>(hipStream_t, const float*, int64_t, const int64_t*, const int64_t*, int, int, float*, float*, float*, float*, int64_t, const CeCompact&);

// out[j] = in[idx[j]] for j < *n (rows of H elements, 16 bytes per lane)
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ in, const int* __restrict__ idx, const int* __restrict__ n, int H, T* __restrict__ out) {
  const int j = blockIdx.x;
  if (j >= *n) return;
  const uint4* src = (const uint4*)(in + (int64_t)idx[j] * H);
  uint4* dst = (uint4*)(out + (int64_t)j * H);
  for (int c = threadIdx.x; c < H * (int)sizeof(T) / 16; c += blockDim.x) dst[c] = src[c];
}
template <typename T> int gather_rows(hipStream_t st, const T* in, const int* idx, const int* n_dev, int max_rows, int H, T* out) {
  if ((H * (int)sizeof(T)) & 15) return RL_ERR_ARG;
  hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(max_rows), dim3(96), 0, st, in, idx, n_dev, H, out);
  return RL_LAUNCH_CHECK();
}
template int gather_rows<bf16_t>(hipStream_t, const bf16_t*, const int*, const int*, int, int, bf16_t*);
template int gather_rows<float>(hipStream_t, const float*, const int*, const int*, int, int, float*);
// out[row] = inv[row] >= 0 ? in[inv[row]] * dropout(row, c) : 0   (the inverse of the gather, fused with the dropout map of the site)
template <typename T>
__global__ void scatter_rows_drop_kernel(const T* __restrict__ in, const int* __restrict__ inv, int H, T* __restrict__ out, DropParams d,
                                         const float* __restrict__ scale_dev) {
  const int row = blockIdx.x, j = inv[row];
  const float sc = scale_dev != nullptr ? *scale_dev : 1.0f;
  for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) {
    floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
    if (j >= 0) v = load4<T>(in + (int64_t)j * H + c) * drop_mult4(d.seed, d.thresh, d.scale, (uint32_t)row * (uint32_t)H + (uint32_t)c) * sc;
    store4<T>(out + (int64_t)row * H + c, v);
  }
}
template <typename T> int scatter_rows_drop(hipStream_t st, const T* in, const int* inv, int rows, int H, T* out, DropParams d, const float* scale_dev) {
  if (H & 3) return RL_ERR_ARG;
  hipLaunchKernelGGL((scatter_rows_drop_kernel<T>), dim3(rows), dim3(192), 0, st, in, inv, H, out, d, scale_dev);
  return RL_LAUNCH_CHECK();
}
// the same scatter fed by the fp32 partial planes of a split-K GEMM: out[row] = inv[row] >= 0 ? (sum_s slab[s][inv[row]]) * dropout : 0,
// planes added in index order (bit-reproducible)
template <typename T>
__global__ void scatter_rows_drop_slab_kernel(const float* __restrict__ slab, int nsplit, int64_t stride, const int* __restrict__ inv, int H,
                                              T* __restrict__ out, DropParams d, const float* __restrict__ scale_dev) {
  const int row = blockIdx.x, j = inv[row];
  const float sc = scale_dev != nullptr ? *scale_dev : 1.0f;
  for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) {
    floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
    if (j >= 0) {
      const float* p = slab + (int64_t)j * H + c;
      v = *(const floatx4*)p;
      for (int s = 1; s < nsplit; ++s) v += *(const floatx4*)(p + s * stride);
      v *= drop_mult4(d.seed, d.thresh, d.scale, (uint32_t)row * (uint32_t)H + (uint32_t)c) * sc;
    }
    store4<T>(out + (int64_t)row * H + c, v);
  }
}
template <typename T> int scatter_rows_drop_slab(hipStream_t st, const float* slab, int nsplit, int64_t stride, const int* inv, int rows, int H, T* out,
                                                 DropParams d, const float* scale_dev) {
  if ((H & 3) || nsplit < 1) return RL_ERR_ARG;
  hipLaunchKernelGGL((scatter_rows_drop_slab_kernel<T>), dim3(rows), dim3(192), 0, st, slab, nsplit, stride, inv, H, out, d, scale_dev);
  return RL_LAUNCH_CHECK();
}
template int scatter_rows_drop_slab<bf16_t>(hipStream_t, const float*, int, int64_t, const int*, int, int, bf16_t*, DropParams, const float*);
template int scatter_rows_drop_slab<float>(hipStream_t, const float*, int, int64_t, const int*, int, int, float*, DropParams, const float*);
template int scatter_rows_drop<bf16_t>(hipStream_t, const bf16_t*, const int*, int, int, bf16_t*, DropParams, const float*);
template int scatter_rows_drop<float>(hipStream_t, const float*, const int*, int, int, float*, DropParams, const float*);

// x *= *scale_dev (n % 4 == 0): the dense-classifier fallback of the incoming loss gradient (engine.hip stage_head)
template <typename T>
__global__ void __launch_bounds__(256) scale_dev_kernel(T* __restrict__ x, int64_t n4, const float* __restrict__ scale_dev) {
  const float sc = *scale_dev;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) store4<T>(x + 4 * i, load4<T>(x + 4 * i) * sc);
}
template <typename T> int scale_by_dev(hipStream_t st, T* x, int64_t n, const float* scale_dev) {
  if ((n & 3) || scale_dev == nullptr) return RL_ERR_ARG;
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL((scale_dev_kernel<T>), dim3((int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048)), dim3(256), 0, st, x, n4, scale_dev);
  return RL_LAUNCH_CHECK();
}
template int scale_by_dev<bf16_t>(hipStream_t, bf16_t*, int64_t, const float*);
template int scale_by_dev<float>(hipStream_t, float*, int64_t, const float*);

// ---------------------------------------------------------------------------------------------
// Gate fusion (models.py:840-850).  The [T, 4H] concat is never materialised.
// ---------------------------------------------------------------------------------------------
// masked mean of the bert states of a sentence (models.py:842-843).  grid (H / 256, B), 256 threads = 64 column quads x 4 row lanes:
// a thread walks S / 4 rows, the four lanes meet in LDS (one thread per column quad walking all S rows serially took 53 us for
// 12.6 MB; this form ~10).  Row lane order and the final 4-way sum are fixed: deterministic.
template <typename T>
__global__ void __launch_bounds__(256) gate_mean_kernel(GateArgs<T> a) {
  __shared__ floatx4 part[4][64];
  __shared__ float mpart[4];
  const int q = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + q) * 4;
  const int b = blockIdx.y;
  floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
  float ms = 0.f;
  for (int s = rl; s < a.S; s += 4) {
    const float m = (float)a.masks[b * a.S + s];
    ms += m;
    // (select, not multiply: a masked row of a live-row step holds whatever an earlier step left there - 0 * NaN must not get in)
    if (c < a.H && m != 0.f) acc += load4<T>(a.bert + ((int64_t)b * a.S + s) * a.H + c) * m;
  }
  part[rl][q] = acc;
  if (q == 0) mpart[rl] = ms;
  __syncthreads();
  if (rl == 0) {
    const float msum = (mpart[0] + mpart[1]) + (mpart[2] + mpart[3]);
    if (c < a.H) *(floatx4*)(a.mean + (int64_t)b * a.H + c) = ((part[0][q] + part[1][q]) + (part[2][q] + part[3][q])) / msum;
    if (c == 0) a.msum[b] = msum;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) gate_fwd_kernel(GateArgs<T> a) {   // one wave per token
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.B * a.S) return;
  const int H = a.H, b = row / a.S;
  floatx4 xb[LN_MAXV], xp[LN_MAXV], xr[LN_MAXV];
  float z[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      xb[i] = load4<T>(a.bert + (int64_t)row * H + c);
      xp[i] = load4<T>(a.pho + (int64_t)row * H + c);
      xr[i] = load4<T>(a.res + (int64_t)row * H + c);
      const floatx4 xm = *(const floatx4*)(a.mean + (int64_t)b * H + c);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float* w = a.W + (int64_t)k * 4 * H + c;
        const floatx4 w0 = *(const floatx4*)w, w1 = *(const floatx4*)(w + H), w2 = *(const floatx4*)(w + 2 * H),
                      w3 = *(const floatx4*)(w + 3 * H);
#pragma unroll
        for (int j = 0; j < 4; ++j) z[k] += w0[j] * xb[i][j] + w1[j] * xp[i][j] + w2[j] * xr[i][j] + w3[j] * xm[j];
      }
    }
  }
  float gk[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) gk[k] = sigmoidf_(wave_sum(z[k]) + a.bias[k]);
  if (lane == 0) { a.g[row * 4 + 0] = gk[0]; a.g[row * 4 + 1] = gk[1]; a.g[row * 4 + 2] = gk[2]; a.g[row * 4 + 3] = 0.f; }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) store4<T>(a.fused + (int64_t)row * H + c, xb[i] * gk[0] + xp[i] * gk[1] + xr[i] * gk[2]);
  }
}
template <typename T> int gate_fwd(hipStream_t st, const GateArgs<T>& a) {
  if ((a.H & 3) || a.H > LN_MAXV * 256) return RL_ERR_ARG;
  hipLaunchKernelGGL((gate_mean_kernel<T>), dim3((a.H / 4 + 63) / 64, a.B), dim3(256), 0, st, a);
  hipLaunchKernelGGL((gate_fwd_kernel<T>), dim3((a.B * a.S + 3) / 4), dim3(256), 0, st, a);
  return RL_LAUNCH_CHECK();
}
template int gate_fwd<bf16_t>(hipStream_t, const GateArgs<bf16_t>&);
template int gate_fwd<float>(hipStream_t, const GateArgs<float>&);

// backward, step 1 (per token): dz_k and the direct parts of dbert/dpho/dres
template <typename T>
__global__ void __launch_bounds__(256) gate_bwd_token_kernel(GateArgs<T> a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.B * a.S) return;
  const int H = a.H;
  if (a.row_live != nullptr && a.row_live[row] == 0) {      // padding row: d fused is an exact zero, so is everything derived from it
    if (lane == 0) *(floatx4*)(a.dz + (int64_t)row * 4) = floatx4{0.f, 0.f, 0.f, 0.f};
    const floatx4 z4 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) { store4<T>(a.dbert + (int64_t)row * H + c, z4); store4<T>(a.dpho + (int64_t)row * H + c, z4); store4<T>(a.dres + (int64_t)row * H + c, z4); }
    }
    return;
  }
  floatx4 df[LN_MAXV];
  float dg[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      df[i] = load4<T>(a.dfused + (int64_t)row * H + c);
      const floatx4 xb = load4<T>(a.bert + (int64_t)row * H + c), xp = load4<T>(a.pho + (int64_t)row * H + c),
                    xr = load4<T>(a.res + (int64_t)row * H + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) { dg[0] += df[i][j] * xb[j]; dg[1] += df[i][j] * xp[j]; dg[2] += df[i][j] * xr[j]; }
    }
  }
  float gk[3], dz[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    gk[k] = a.g[row * 4 + k];
    dz[k] = wave_sum(dg[k]) * gk[k] * (1.0f - gk[k]);
  }
  if (lane == 0) { a.dz[row * 4 + 0] = dz[0]; a.dz[row * 4 + 1] = dz[1]; a.dz[row * 4 + 2] = dz[2]; a.dz[row * 4 + 3] = 0.f; }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      floatx4 ob = df[i] * gk[0], op = df[i] * gk[1], orr = df[i] * gk[2];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float* w = a.W + (int64_t)k * 4 * H + c;
        ob += *(const floatx4*)w * dz[k];
        op += *(const floatx4*)(w + H) * dz[k];
        orr += *(const floatx4*)(w + 2 * H) * dz[k];
      }
      store4<T>(a.dbert + (int64_t)row * H + c, ob);
      store4<T>(a.dpho + (int64_t)row * H + c, op);
      store4<T>(a.dres + (int64_t)row * H + c, orr);
    }
  }
}
// step 2 (per sentence): d(mean) -> spread over the masked tokens of dbert, and dW[:, 3H:4H] += sum_s dz * mean.
// grid (H / 256, B), 256 threads = 64 column quads x 4 row lanes (the row walk over S is split four ways).
template <typename T>
__global__ void __launch_bounds__(256) gate_bwd_mean_kernel(GateArgs<T> a) {
  __shared__ float zpart[4][4];
  const int q = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + q) * 4;
  const int b = blockIdx.y;
  const int H = a.H;
  if (q < 3) {                                   // lane q of row lane rl: partial sum of dz_q over its rows
    float z = 0.f;
    for (int s = rl; s < a.S; s += 4) z += a.dz[((int64_t)b * a.S + s) * 4 + q];
    zpart[rl][q] = z;
  }
  __syncthreads();
  float zs[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) zs[k] = (zpart[0][k] + zpart[1][k]) + (zpart[2][k] + zpart[3][k]);
  if (c >= H) return;
  floatx4 dm = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k) dm += *(const floatx4*)(a.W + (int64_t)k * 4 * H + 3 * H + c) * zs[k];
  if (rl == 0) {
    const floatx4 mean = *(const floatx4*)(a.mean + (int64_t)b * H + c);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(a.dW + (int64_t)k * 4 * H + 3 * H + c + j, zs[k] * mean[j]);
    if (c == 0)
#pragma unroll
      for (int k = 0; k < 3; ++k) atomicAdd(a.dbias + k, zs[k]);
  }
  dm = dm / a.msum[b];
  for (int s = rl; s < a.S; s += 4) {
    const int64_t m = a.masks[b * a.S + s];
    if (m != 0) {
      T* p = a.dbert + ((int64_t)b * a.S + s) * H + c;
      store4<T>(p, load4<T>(p) + dm * (float)m);
    }
  }
}
// step 3: dW[k, src H + c] += sum_t dz_k[t] * X_src(t, c) for the three sources and the three gates in ONE pass over bert / pho / res
// (was nine column reductions, each re-reading one source: 190 us of launches at the join of the three branches).
// grid (H / 128, row chunks), 256 threads = 32 column quads x 8 row lanes; nine float4 accumulators per thread; atomics at the end.
template <typename T>
__global__ void __launch_bounds__(256) gate_dw_kernel(GateArgs<T> a, int rows_per_block) {
  __shared__ floatx4 red[8][32];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cx) * 4;
  const int H = a.H, T_ = a.B * a.S;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(T_, r0 + rows_per_block);
  floatx4 acc[3][3];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[s][k] = floatx4{0.f, 0.f, 0.f, 0.f};
  if (c < H) {
    for (int r = r0 + ry; r < r1; r += 8) {
      const floatx4 dz = *(const floatx4*)(a.dz + (int64_t)r * 4);
      if (dz[0] == 0.f && dz[1] == 0.f && dz[2] == 0.f) continue;      // padding rows (and any other all-zero row): nothing to add, nothing read
      const floatx4 x[3] = {load4<T>(a.bert + (int64_t)r * H + c), load4<T>(a.pho + (int64_t)r * H + c), load4<T>(a.res + (int64_t)r * H + c)};
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[s][k] += x[s] * dz[k];
    }
  }
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      __syncthreads();
      red[ry][cx] = acc[s][k];
      __syncthreads();
      if (ry == 0 && c < H) {
        floatx4 v = red[0][cx];
#pragma unroll
        for (int j = 1; j < 8; ++j) v += red[j][cx];
        float* o = a.dW + (int64_t)k * 4 * H + (int64_t)s * H + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(o + j, v[j]);
      }
    }
}
template <typename T> int gate_bwd(hipStream_t st, const GateArgs<T>& a) {
  if ((a.H & 3) || a.H > LN_MAXV * 256) return RL_ERR_ARG;
  const int T_ = a.B * a.S;
  hipLaunchKernelGGL((gate_bwd_token_kernel<T>), dim3((T_ + 3) / 4), dim3(256), 0, st, a);
  hipLaunchKernelGGL((gate_bwd_mean_kernel<T>), dim3((a.H / 4 + 63) / 64, a.B), dim3(256), 0, st, a);
  const int gx = (a.H + 127) / 128;
  int gy = 1024 / gx;
  if (gy > (T_ + 31) / 32) gy = (T_ + 31) / 32;
  if (gy < 1) gy = 1;
  const int rows_per_block = (T_ + gy - 1) / gy;
  hipLaunchKernelGGL((gate_dw_kernel<T>), dim3(gx, gy), dim3(256), 0, st, a, rows_per_block);
  return RL_LAUNCH_CHECK();
}
template int gate_bwd<bf16_t>(hipStream_t, const GateArgs<bf16_t>&);
template int gate_bwd<float>(hipStream_t, const GateArgs<float>&);

// ---------------------------------------------------------------------------------------------
// Glyph dedup: distinct token ids of the batch in order of first occurrence (deterministic).
// ---------------------------------------------------------------------------------------------
__global__ void gu_clear_kernel(int* first, int V, float* counts, int T_) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < V) first[i] = 0x7fffffff;
  if (i < T_) counts[i] = 0.f;
}
__global__ void gu_first_kernel(const int64_t* __restrict__ ids, int T_, int* first) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < T_) atomicMin(first + (int)ids[t], t);
}
// single workgroup: exclusive scan of is_first over t -> slot of every first occurrence
__global__ void __launch_bounds__(1024) gu_scan_kernel(const int64_t* __restrict__ ids, int T_, const int* __restrict__ first,
                                                       int* slot_of_t, int64_t* uniq_ids, int* bounds, HwList hw) {
  __shared__ int wsum[16];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < T_; base += 1024) {
    const int t = base + threadIdx.x;
    const int flag = (t < T_ && first[(int)ids[t]] == t) ? 1 : 0;
    int v = flag;                                   // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(v, o, 64); if (lane >= o) v += n; }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    const int excl = off + v - flag;
    if (t < T_) {
      slot_of_t[t] = flag ? excl : -1;
      if (flag) uniq_ids[excl] = ids[t];
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = off + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int U = carry;
    bounds[0] = U;
    for (int k = 0; k < hw.n; ++k) bounds[1 + k] = U * hw.v[k];
  }
}
__global__ void gu_final_kernel(const int64_t* __restrict__ ids, int T_, const int* __restrict__ first, const int* __restrict__ slot_of_t,
                                int* inv, float* counts) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T_) return;
  const int s = slot_of_t[first[(int)ids[t]]];
  inv[t] = s;
  atomicAdd(counts + s, 1.0f);
}
static int g_glyph_dedup = 1;
void set_glyph_dedup(int on) { g_glyph_dedup = on; }
__global__ void gu_identity_kernel(const int64_t* __restrict__ ids, int T_, int64_t* uniq_ids, float* counts, int* inv, int* bounds, HwList hw) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < T_) { uniq_ids[t] = ids[t]; counts[t] = 1.0f; inv[t] = t; }
  if (t == 0) { bounds[0] = T_; for (int k = 0; k < hw.n; ++k) bounds[1 + k] = T_ * hw.v[k]; }
}
int glyph_unique(hipStream_t st, const int64_t* ids, int T_, int V, int* first_scratch, int* flag_scratch, int64_t* uniq_ids,
                 float* counts, int* inv, int* bounds, HwList hw) {
  if (!g_glyph_dedup) {
    hipLaunchKernelGGL(gu_identity_kernel, dim3((T_ + 255) / 256), dim3(256), 0, st, ids, T_, uniq_ids, counts, inv, bounds, hw);
    return RL_LAUNCH_CHECK();
  }
  const int n = V > T_ ? V : T_;
  hipLaunchKernelGGL(gu_clear_kernel, dim3((n + 255) / 256), dim3(256), 0, st, first_scratch, V, counts, T_);
  hipLaunchKernelGGL(gu_first_kernel, dim3((T_ + 255) / 256), dim3(256), 0, st, ids, T_, first_scratch);
  hipLaunchKernelGGL(gu_scan_kernel, dim3(1), dim3(1024), 0, st, ids, T_, first_scratch, flag_scratch, uniq_ids, bounds, hw);
  hipLaunchKernelGGL(gu_final_kernel, dim3((T_ + 255) / 256), dim3(256), 0, st, ids, T_, first_scratch, flag_scratch, inv, counts);
  return RL_LAUNCH_CHECK();
}

template <typename T>
__global__ void segsum_scatter_kernel(const T* __restrict__ x, const int* __restrict__ inv, int T_, int C, float* acc) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int t = blockIdx.y;
  if (c >= C) return;
  const floatx4 v = load4<T>(x + (int64_t)t * C + c);
  if (quad_is_zero(v)) return;                       // padded tokens: exact zeros, all on the slot of token id 0
  float* o = acc + (int64_t)inv[t] * C + c;
#pragma unroll
  for (int j = 0; j < 4; ++j) atomicAdd(o + j, v[j]);
}
template <typename T>
__global__ void segsum_cast_kernel(const float* __restrict__ acc, T* __restrict__ out, int C, const int* __restrict__ nuniq) {
  const int64_t n = (int64_t)(*nuniq) * C;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4)
    store4<T>(out + i, *(const floatx4*)(acc + i));
}
template <typename T> int segment_sum(hipStream_t st, const T* x, const int* inv, int T_, int C, float* acc, T* out, const int* nuniq_dev) {
  if (C & 3) return RL_ERR_ARG;
  (void)hipMemsetAsync(acc, 0, (size_t)T_ * C * sizeof(float), st);
  hipLaunchKernelGGL((segsum_scatter_kernel<T>), dim3((C / 4 + 63) / 64, T_), dim3(64), 0, st, x, inv, T_, C, acc);
  hipLaunchKernelGGL((segsum_cast_kernel<T>), dim3(1024), dim3(256), 0, st, acc, out, C, nuniq_dev);
  return RL_LAUNCH_CHECK();
}
template int segment_sum<bf16_t>(hipStream_t, const bf16_t*, const int*, int, int, float*, bf16_t*, const int*);
template int segment_sum<float>(hipStream_t, const float*, const int*, int, int, float*, float*, const int*);

// dense[u] = table[ids[u]] for the U distinct glyphs of the batch (one contiguous NHWC image each): the block-1 convolutions and
// their weight gradients then address images by slot, without the id indirection in their gather loops
template <typename T>
__global__ void __launch_bounds__(256) gather_images_kernel(const T* __restrict__ table, const int64_t* __restrict__ ids,
                                                            const int* __restrict__ nuniq, int64_t elems, T* __restrict__ dense) {
  const int u = blockIdx.x;
  if (u >= *nuniq) return;
  const uint4* s = (const uint4*)(table + ids[u] * elems);
  uint4* d = (uint4*)(dense + (int64_t)u * elems);
  const int n16 = (int)(elems * sizeof(T) / 16);
  for (int i = threadIdx.x; i < n16; i += 256) d[i] = s[i];
}
template <typename T> int gather_images(hipStream_t st, const T* table, const int64_t* ids, const int* nuniq, int max_images, int64_t elems, T* dense) {
  if ((elems * sizeof(T)) & 15) return RL_ERR_ARG;
  hipLaunchKernelGGL((gather_images_kernel<T>), dim3(max_images), dim3(256), 0, st, table, ids, nuniq, elems, dense);
  return RL_LAUNCH_CHECK();
}
template int gather_images<bf16_t>(hipStream_t, const bf16_t*, const int64_t*, const int*, int, int64_t, bf16_t*);
template int gather_images<float>(hipStream_t, const float*, const int64_t*, const int*, int, int64_t, float*);

// out[t, :] = x[inv[t], :]  (per-distinct-glyph rows back to per-token rows; the glyph-only entry point, BASELINE configs[3])
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ x, const int* __restrict__ inv, int T_, int C, T* __restrict__ out) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int t = blockIdx.y;
  if (c >= C) return;
  store4<T>(out + (int64_t)t * C + c, load4<T>(x + (int64_t)inv[t] * C + c));
}
template <typename T> int gather_rows(hipStream_t st, const T* x, const int* inv, int T_, int C, T* out) {
  if (C & 3) return RL_ERR_ARG;
  hipLaunchKernelGGL((gather_rows_kernel<T>), dim3((C / 4 + 63) / 64, T_), dim3(64), 0, st, x, inv, T_, C, out);
  return RL_LAUNCH_CHECK();
}
template int gather_rows<bf16_t>(hipStream_t, const bf16_t*, const int*, int, int, bf16_t*);
template int gather_rows<float>(hipStream_t, const float*, const int*, int, int, float*);

// ---------------------------------------------------------------------------------------------
// Row-wise argmax: one 256-thread workgroup per row, 16-byte loads, (value, index) pairs reduced with the
// first-occurrence tie rule.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool amax_better(float v, int i, float bv, int bi) {
  const bool vn = v != v, bn = bv != bv;
  if (vn || bn) return vn && (!bn || i < bi);
  return v > bv || (v == bv && i < bi);
}
template <typename T>
__global__ void __launch_bounds__(256) argmax_kernel(const T* __restrict__ logits, int64_t ld, int V, int64_t* __restrict__ ids) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const T* row = logits + (int64_t)blockIdx.x * ld;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  const int v8 = ((ld & 7) == 0) ? (V & ~7) : 0;          // 8-wide body when every row start is 16-byte aligned
  for (int c = threadIdx.x * 8; c < v8; c += 256 * 8) {
    floatx4 a, b;
    load8<T>(row + c, a, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) if (amax_better(a[j], c + j, bv, bi)) { bv = a[j]; bi = c + j; }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (amax_better(b[j], c + 4 + j, bv, bi)) { bv = b[j]; bi = c + 4 + j; }
  }
  for (int c = v8 + threadIdx.x; c < V; c += 256) {
    const float v = to_f<T>(row[c]);
    if (amax_better(v, c, bv, bi)) { bv = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (amax_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) if (amax_better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
    ids[blockIdx.x] = bi;
  }
}
template <typename T> int argmax_rows(hipStream_t st, const T* logits, int64_t ld, int rows, int V, int64_t* ids) {
  if (rows <= 0) return RL_OK;
  if (V <= 0 || ld < V) return RL_ERR_ARG;
  hipLaunchKernelGGL((argmax_kernel<T>), dim3(rows), dim3(256), 0, st, logits, ld, V, ids);
  return RL_LAUNCH_CHECK();
}
template int argmax_rows<bf16_t>(hipStream_t, const bf16_t*, int64_t, int, int, int64_t*);
template int argmax_rows<float>(hipStream_t, const float*, int64_t, int, int, int64_t*);

}  // namespace rl
