// Ping-pong 8-wave NT GEMM for gfx950 (bf16 operands, K % 8 == 0, dense rows):  C[M,N] = A[M,K] . B[N,K]^T + fused epilogue.
//
// Why a second NT kernel: the LDS-DMA fill path delivers ~36-45 B/clk per CU whatever the tile shape, so the
// FLOPs one fetched byte feeds - the workgroup tile - decide how far the fill is from binding:
//   128 x 128 (gemm.hip, two 4-wave workgroups per CU)   64 FLOP/B
//   128 x 192 (this kernel, TWO workgroups per CU)        77 FLOP/B   <- every layer GEMM (N < 4096): the second workgroup's MFMAs run
//                                                                       under the first one's output write and fetch waits
//   256 x 192 / 256 x 256 (this kernel, one per CU)      110 / 128     <- the 21128-wide classifier
// Measured (tools/nt8_probe.cpp, profiles/round2_nt8_probe.log): MFMA, fill and output write are three comparable costs on the
// K = 768 shapes; with the fragment reads the LDS is ~100 % busy (reads queue behind the DMA writes), which is why dedicated
// loader waves (gemm_nt8ws_kernel below) change nothing, and the result sits level with hipBLASLt on the same shapes.
// A 512-thread workgroup, 2 waves per SIMD: waves 0-3 (group 0) and 4-7 (group 1) sit pairwise on the four SIMDs and
// run HALF A PHASE apart (group 1 executes one extra s_barrier up front), so on every SIMD one wave is in its MEMORY segment
// (fragment ds_reads + its share of the LDS-DMA fetches + the counted vmcnt wait) while its partner is in its MFMA segment
// (16 v_mfma_f32_16x16x32_bf16 on register-resident fragments, s_setprio 1).
//
// K-tile = 64 (one 128-byte LDS row per operand row, 16-byte chunks XOR-swizzled by row: KTile).  A K-tile is consumed in NPH
// phases: the HELD operand's fragments of the whole K-tile are read once (phase 0) and stay in registers, the STREAMED operand
// is read SQ 16-row tiles per phase.  The ring has NS K-tile stages; the 1-KiB pieces (8 rows) of a stage are fetched in the
// order the phases need them (held operand + streamed group 0 first) and are issued LEAD phases ahead of the phase that reads
// their tile's first fragment, a fixed number per phase, so every s_waitcnt vmcnt(N) in the loop is a compile-time count and
// never 0 in steady state.  Hazards (both groups, half a phase apart):
//   RAW  a piece is waited for (own vmcnt, then a barrier) in the phase BEFORE the one that reads it;
//   WAR  a region is re-filled >= 2 phases after the phase that read it last.
// Both are static_assert-ed on the schedule below.
#include "gemm_dev.h"
#include "nt8_cfg.h"
#include "prof.h"

namespace rl {

static int g_nt8_group_m = 0;
static int g_nt8_single_round = 0;        // 1: one-round outputs on the three-stage one-per-CU 128 x 192 shape (realise_set_nt8p key 2)
void set_nt8_single_round(int on) { g_nt8_single_round = on; }
void set_nt8_group_m(int g) { g_nt8_group_m = g; }

// Logical tile id -> (tile row, tile column).  group_m <= 1: row-major (a run of consecutive ids walks along N: its tiles share
// few A rows but touch every B column panel).  group_m > 1: ids walk down group_m tile rows before moving to the next tile
// column, so the ~64 tiles an XCD has in flight form a group_m x (64 / group_m) block whose A and B panels fit its 4 MiB L2.
__device__ __forceinline__ void tile_coords(int tile, int tiles_n, int ntiles, int group_m, int& tm, int& tn) {
  if (group_m <= 1) { tm = tile / tiles_n; tn = tile - tm * tiles_n; return; }
  const int tiles_m = ntiles / tiles_n;
  const int per_group = group_m * tiles_n;
  const int g = tile / per_group, r = tile - g * per_group;
  const int rows = min(group_m, tiles_m - g * group_m);
  tn = r / rows;
  tm = g * group_m + (r - tn * rows);
}

static int g_nt8_probe = 0;
void set_nt8_probe(int mode) { g_nt8_probe = mode; }
static int g_nt8_cu_pair = 0;
void set_nt8_cu_pair(int on) { g_nt8_cu_pair = on; }
static int g_nt8_l2_prefetch = 0;       // realise_set_nt8p key 8: bits 0-3 prefetch workgroups per XCD, 4-7 K-tiles ahead before pacing, 8-15 s_sleep per K-tile
void set_nt8_l2_prefetch(int v) { g_nt8_l2_prefetch = v; }
static int nt8_bias_first_on();

// ---- K4 epilogue: dropout(acc + bias) + residual, then the LayerNorm of the row, whose columns are spread over the N / BN workgroups
// of a row band (see EpiParams::ln_*).  Same numbers as the unfused pair (GEMM epilogue -> bf16 -> ln_fwd16) up to the order of the
// fp32 statistics: the statistics are taken from the bf16-rounded sums, two-pass inside a tile, combined across tiles by Chan's rule.
template <typename C>
__device__ __forceinline__ void nt8_ln_epilogue(char* smem, const EpiParams<bf16_t>& ep, floatx4 (&acc)[C::MT][C::NT], int M, int N, int m0, int n0,
                                                int tm, int tn, int tiles_n, int wm, int wn, int lane) {
  typedef bf16_t T;
  constexpr int BM = C::BM, BN = C::BN, PITCH = BN * 2 + 16;          // bf16 x tile in LDS, rows padded by 16 B
  static_assert(BM * PITCH + BM * 8 <= C::LDS && BM == 128 && BN % 64 == 0, "LN epilogue geometry");
  const int g = lane >> 4, l15 = lane & 15, tid = threadIdx.x;
  float* rowstat = (float*)(smem + BM * PITCH);                         // [BM][2]: mean, rstd
  // A. x = bf16(dropout(acc * alpha + bias) + residual), in the MFMA layout (a lane: 4 consecutive columns of a row) -> LDS
#pragma unroll
  for (int i = 0; i < C::MT; ++i) {
    const int rl = wm * C::RM + i * 16 + l15, row = m0 + rl;
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      const int cl = wn * C::RN + j * 16 + 4 * g, col = n0 + cl;
      floatx4 a = acc[i][j];
      if (ep.alpha != 1.0f) a *= ep.alpha;
      if (ep.bias != nullptr) a += *(const floatx4*)(ep.bias + col);
      a *= drop_mult4(ep.drop_seed, ep.drop_thresh, ep.drop_scale, (uint32_t)row * (uint32_t)N + (uint32_t)col);
      a += load4<T>(ep.aux + (int64_t)row * ep.ldaux + col);
      uint2 pk;
      pk.x = pack2bf(a[0], a[1]); pk.y = pack2bf(a[2], a[3]);
      *(uint2*)(smem + rl * PITCH + cl * 2) = pk;
    }
  }
  __syncthreads();
  // B. per-row statistics of this tile's BN columns: 4 threads per row (BM * 4 = 512 threads), BN / 4 columns each
  const int srow = tid >> 2, sq = tid & 3;
  constexpr int QC = BN / 4, QCH = QC / 8;
  float xs[QC];
  {
    const char* p = smem + srow * PITCH + sq * QC * 2;
#pragma unroll
    for (int k = 0; k < QCH; ++k) {
      const uint4 u = *(const uint4*)(p + 16 * k);
      floatx4 lo, hi;
      unpack8(u, lo, hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) { xs[8 * k + e] = lo[e]; xs[8 * k + 4 + e] = hi[e]; }
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < QC; ++k) sum += xs[k];
  sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64);
  const float mean_t = sum * (1.0f / (float)BN);
  float m2 = 0.f;
#pragma unroll
  for (int k = 0; k < QC; ++k) { const float d = xs[k] - mean_t; m2 += d * d; }
  m2 += __shfl_xor(m2, 1, 64); m2 += __shfl_xor(m2, 2, 64);
  // C. hand the partials to the other column tiles of the row band and collect theirs.  A slot is two 64-bit words {value, tag}:
  // each word is written by ONE device-scope atomic store and validates itself (tag = this launch's number; the previous content of
  // a slot is always the previous launch's, whose tag differs) - no fence, no flag, no second round trip: a reader simply polls the
  // slots it needs.  The wait is bounded: a tile that never arrives must not hang the GPU (ln_timeout tells the host).
  unsigned long long* part = (unsigned long long*)ep.ln_part + ((int64_t)(m0 + srow) * tiles_n) * 2;
  const unsigned long long tag = (unsigned long long)(uint32_t)ep.ln_target << 32;
  if (sq == 0) {
    __hip_atomic_store(part + tn * 2, tag | (unsigned long long)__float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(part + tn * 2 + 1, tag | (unsigned long long)__float_as_uint(m2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // D. combine (Chan's rule, n_t = BN each); the 4 threads of a row take tiles sq and sq + 4
  float s_a = 0.f, s_b = 0.f, pm2 = 0.f;
  const bool has_a = sq < tiles_n, has_b = sq + 4 < tiles_n;
  auto fetch = [&](int t, float& s_out, float& m_out) {
    unsigned long long w0, w1;
    int spins = 0;
    for (;;) {
      w0 = __hip_atomic_load(part + t * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      w1 = __hip_atomic_load(part + t * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (((w0 ^ tag) >> 32) == 0ull && ((w1 ^ tag) >> 32) == 0ull) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 20)) { if (ep.ln_timeout != nullptr) *ep.ln_timeout = 1; break; }
    }
    s_out = __uint_as_float((uint32_t)w0); m_out = __uint_as_float((uint32_t)w1);
  };
  if (ep.ln_flag != nullptr) {          // diagnostics (tools/repro_probe.py, LNDEBUG=1): no hand-off - every tile uses its own partial for all tiles
    if (has_a) { s_a = sum; pm2 = m2; }   // (wrong statistics, but what is left is deterministic unless phases A / B / E race)
    if (has_b) { s_b = sum; pm2 += m2; }
  } else {
    if (has_a) fetch(sq, s_a, pm2);
    if (has_b) { float m_b; fetch(sq + 4, s_b, m_b); pm2 += m_b; }
  }
  float tsum = s_a + s_b;
  tsum += __shfl_xor(tsum, 1, 64); tsum += __shfl_xor(tsum, 2, 64);
  const float mean = tsum / (float)N;
  const float da = s_a * (1.0f / (float)BN) - mean, db = s_b * (1.0f / (float)BN) - mean;
  float tot = pm2 + (has_a ? (float)BN * da * da : 0.f) + (has_b ? (float)BN * db * db : 0.f);      // sum_t [M2_t + n_t (mean_t - mean)^2]
  tot += __shfl_xor(tot, 1, 64); tot += __shfl_xor(tot, 2, 64);
  const float rstd = 1.0f / sqrtf(tot / (float)N + ep.ln_eps);
  if (sq == 0 && tn == 0 && ep.ln_rstd != nullptr) ep.ln_rstd[m0 + srow] = rstd;
  // E. normalise this tile's columns from the registers of phase B: the four threads of a row own 48 consecutive columns each = 96
  // contiguous bytes of xhat (written over the pre-LN buffer) and of y (16-byte stores; a row's four threads cover 384 contiguous bytes).
  // (Round 4's first form re-read the tile from LDS by 8-column items after a third barrier; tools/repro_probe.py showed one
  // aligned 16-lane group per ~2000 launches leaving that re-read with one wrong element - this form has no re-read to go wrong.)
  (void)rowstat;
  const int col0 = n0 + sq * QC;
  const int64_t o = (int64_t)(m0 + srow) * ep.ldo + col0;
#pragma unroll
  for (int k = 0; k < QCH; ++k) {
    const floatx4 g0 = *(const floatx4*)(ep.ln_gamma + col0 + 8 * k), g1 = *(const floatx4*)(ep.ln_gamma + col0 + 8 * k + 4);
    const floatx4 b0 = *(const floatx4*)(ep.ln_beta + col0 + 8 * k), b1 = *(const floatx4*)(ep.ln_beta + col0 + 8 * k + 4);
    floatx4 h0, h1;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h0[e] = (xs[8 * k + e] - mean) * rstd; h1[e] = (xs[8 * k + 4 + e] - mean) * rstd; }
    store8<T>(ep.out + o + 8 * k, h0, h1);
    store8<T>(ep.ln_y + o + 8 * k, h0 * g0 + b0, h1 * g1 + b1);
  }
}

// Row state of the GRU epilogue, fetched at kernel start so that the dependent chain perm -> pho_idx -> table row (three global
// latencies) and the h_prev rows run under the main loop instead of in front of every 16-row chunk of the epilogue.
struct GruRows { int tok[2]; int gi_off[2]; bool ends[2]; uint4 hp[2]; };
template <typename C>
__device__ __forceinline__ void nt8_gru_prefetch(const EpiParams<bf16_t>& ep, GruRows& gr, int mlive, int H, int m0, int tn, int wm, int wn, int lane) {
  const int r = lane >> 2, o = lane & 3;
  const int unit0 = tn * 64 + wn * 32 + o * 8;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int i = m0 + wm * C::RM + c * 16 + r;
    const bool ok = i < mlive;
    const int tok = ok ? ep.gru_perm[i] : 0;
    const int64_t v = ok ? ep.gru_pho_idx[(int64_t)tok * ep.gru_Tp + ep.gru_t] : 0;
    gr.tok[c] = tok;
    gr.gi_off[c] = (int)v * 3 * H + unit0;
    gr.ends[c] = ok && ep.gru_lens[i] == ep.gru_t + 1;
    gr.hp[c] = ok ? *(const uint4*)(ep.gru_hprev + (int64_t)i * H + unit0) : uint4{0u, 0u, 0u, 0u};
  }
}
template <typename C>
__device__ __forceinline__ void nt8_gru_epilogue(char* smem, const EpiParams<bf16_t>& ep, floatx4 (&acc)[C::MT][C::NT], const GruRows& gr, int mlive, int H,
                                                 int m0, int tn, int wave, int wm, int wn, int lane) {
  typedef bf16_t T;
  constexpr int RS = C::RS, ER = 16;
  static_assert(C::RN == 96 && C::RM == 32 && 8 * ER * RS * 4 <= C::LDS, "GRU epilogue geometry");
  const int g = lane >> 4, l15 = lane & 15;
  float* et = (float*)smem + wave * (ER * RS);
  const int r = lane >> 2, o = lane & 3;
  const int unit0 = tn * 64 + wn * 32 + o * 8;
#pragma unroll
  for (int c = 0; c < C::RM / ER; ++c) {
#pragma unroll
    for (int j = 0; j < C::NT; ++j) *(floatx4*)(et + l15 * RS + j * 16 + 4 * g) = acc[c][j];
    // (a wave reads back only what it wrote: no barrier, the LDS ops of a wave are in order)
    const int i = m0 + wm * C::RM + c * ER + r;
    if (i >= mlive) continue;
    float hr[8], hz[8], hn[8];
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
      *(floatx4*)&hr[e] = *(const floatx4*)(et + r * RS + o * 8 + e);
      *(floatx4*)&hz[e] = *(const floatx4*)(et + r * RS + 32 + o * 8 + e);
      *(floatx4*)&hn[e] = *(const floatx4*)(et + r * RS + 64 + o * 8 + e);
    }
    const float* b = ep.bias + unit0;
    const float* gi = ep.gru_table + gr.gi_off[c];
    float br[8], bz[8], bn[8], ir[8], iz[8], in[8];
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
      *(floatx4*)&br[e] = *(const floatx4*)(b + e); *(floatx4*)&bz[e] = *(const floatx4*)(b + H + e); *(floatx4*)&bn[e] = *(const floatx4*)(b + 2 * H + e);
      *(floatx4*)&ir[e] = *(const floatx4*)(gi + e); *(floatx4*)&iz[e] = *(const floatx4*)(gi + H + e); *(floatx4*)&in[e] = *(const floatx4*)(gi + 2 * H + e);
    }
    floatx4 p0, p1;
    unpack8(gr.hp[c], p0, p1);
    float rr[8], zz[8], nn[8], hh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // bias, then the bf16 round trip of the two-launch form (GEMM stores bf16(acc + b_hh), the gate kernel reads it back)
      const float xr = bf2f(f2bf(hr[e] + br[e])), xz = bf2f(f2bf(hz[e] + bz[e]));
      hn[e] = bf2f(f2bf(hn[e] + bn[e]));
      const float hp = e < 4 ? p0[e] : p1[e - 4];
      gru_unit<T>(ir[e], iz[e], in[e], xr, xz, hn[e], hp, rr[e], zz[e], nn[e], hh[e]);
    }
    auto st8 = [&](T* p, const float (&x)[8]) { store8<T>(p, floatx4{x[0], x[1], x[2], x[3]}, floatx4{x[4], x[5], x[6], x[7]}); };
    if (ep.gru_rzn != nullptr) {
      T* s = ep.gru_rzn + (int64_t)i * 3 * H + unit0;
      st8(s, rr); st8(s + H, zz); st8(s + 2 * H, nn);
    }
    if (ep.gru_gh != nullptr) st8(ep.gru_gh + (int64_t)i * 3 * H + 2 * H + unit0, hn);
    st8(ep.out + (int64_t)i * H + unit0, hh);
    if (gr.ends[c]) st8(ep.gru_out + (int64_t)gr.tok[c] * H + unit0, hh);
  }
}

// ---- L2 prefetch workgroups (round 6 probe, EpiParams::l2_prefetch).  All tiles of a launch walk K in step, so every operand line - A
// row r x K-tile k, shared by the column tiles of r's tile row; B row n x K-tile k, shared by the tile rows of an XCD - is asked for the
// first time by all of its sharers at once: every fill of every K-tile waits out a memory-side miss, and a miss holds one of the CU's ~90
// request slots 3-4 x as long as an L2 hit (section 6.7).  A narrow launch (N = 768: 21 tiles per XCD on 32 CUs) leaves surplus
// workgroups of the nominal grid on idle CUs of the same XCD; instead of leaving at once they touch - 4 bytes per 128-byte line - the
// lines their XCD's tiles will ask for, `ahead` K-tiles in front of them, so that the readers find the lines in the L2.  No effect on
// any result.  MEASURED (profiles/round6_ab.log): the N = 768, K = 768 launches 19.6 -> 18.4 us (-6 %), the K = 2304 / 3072 ones level
// with every pacing tried, the step 0.1-0.4 ms SLOWER (the prefetch workgroups sit on CUs the other streams' kernels were using): the
// memory-side part of the fill latency is the smaller part.  Probe build only.
template <typename C>
__device__ __forceinline__ void nt8_l2_prefetch(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int N, int K,
                                                int tiles_n, int ntl, int nlive, const EpiParams<bf16_t>& ep) {
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int q = ntl >> 3, r = ntl & 7;
  const int len = x < r ? q + 1 : q, base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  const int P = min(32 - len, ep.l2_prefetch & 15), p = j - len;           // this XCD's prefetch workgroups: local indices len .. len + P - 1
  if (len <= 0 || p < 0 || p >= P) return;
  const int tm0 = base / tiles_n, tm1 = (base + len - 1) / tiles_n;
  const int ncol = min(N, tiles_n * C::BN), nrow = (tm1 - tm0 + 1) * C::BM, L = ncol + nrow;
  const int sleep = (ep.l2_prefetch >> 8) & 255, ahead = max(1, (ep.l2_prefetch >> 4) & 15);
  const int nk = (K + 63) >> 6;
  const int l = p * 512 + (int)threadIdx.x;
  const char* src = nullptr;
  if (l < L && P * 512 >= L) {                                             // (one line per thread per K-tile; a launch too big for that is left alone)
    if (l < ncol) src = (const char*)(B + (int64_t)l * ldb);
    else {
      const int e = tm0 * C::BM + (l - ncol);
      const int row = e < nlive ? ep.live_list[e] : -1;
      if (row >= 0) src = (const char*)(A + (int64_t)row * lda);
    }
  }
  if (src == nullptr) return;
#if defined(__HIP_DEVICE_COMPILE__)
  for (int k = 0; k < nk; ++k) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(src + (int64_t)k * 128) : "memory");
    if (k >= ahead) {
      for (int z = 0; z < sleep; ++z) __builtin_amdgcn_s_sleep(4);         // (sleep x 256 clocks per K-tile)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// XEPI: 0 = the standard epilogues (EpiParams::mode), 1 = K4 (+ LayerNorm across the row band's tiles), 2 = K6 (GRU gate math on
// gate-interleaved B rows), 3 = the standard epilogues over a LIST of live 16-row blocks (EpiParams::live_list) - separate
// instantiations, so that the special forms' registers are not the layer GEMMs' problem; 4 = the same over a LIST of live ROWS
// (round 6, EpiParams::live_unit == 1: row-granular packing - a tile is ANY 128 listed rows, so no tile row is spent on the padding
// rows that complete a sentence's last 16-row block; the per-lane fetch offset of a piece and the output row of an epilogue item
// come from the list, nothing changes in the main loop)
template <typename C, int PROBE, bool KTAIL = false, int XEPI = 0>
__global__ void __launch_bounds__(512, 2 * C::WGS)
gemm_nt8_kernel(const bf16_t* __restrict__ A_, int64_t lda, const bf16_t* __restrict__ B_, int64_t ldb, int M, int N, int K, int tiles_n,
                int ntiles, int group_m, EpiParams<bf16_t> ep) {
  typedef bf16_t T;
  typedef MmaBF16 Mma;
  constexpr int NPW = C::NPW, NPH = C::NPH, NS = C::NS, LEAD = C::LEAD, SQ = C::SQ, HT = C::HT, MT = C::MT, NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave / C::WN, wn = wave - wm * C::WN;
  // split-K launches (ep.ksplit > 1): ksplit x ntiles workgroups, split-major so that the tiles of one K-range - which share its
  // operand columns - sit behind the same L2s; the workgroup's K-range becomes its whole problem (operands advanced, K shortened)
  const bf16_t* __restrict__ A = A_;
  const bf16_t* __restrict__ B = B_;
  int tile, split = 0;
  if (!KTAIL && ep.ksplit > 1) {
    const int logical = xcd_remap(blockIdx.x, ntiles * ep.ksplit);
    split = logical / ntiles; tile = logical - split * ntiles;
    const int per = (((K + 63) >> 6) + ep.ksplit - 1) / ep.ksplit;
    const int k0 = split * per * 64;
    A += k0; B += k0; K = min(K - k0, per * 64);
  } else if constexpr (XEPI == 3 || XEPI == 4) {
    tile = 0;
  } else {
    tile = ep.cu_pair ? xcd_remap_paired(blockIdx.x, ntiles) : xcd_remap(blockIdx.x, ntiles);
  }
  int tm, tn;
  int nlive = 0;                                            // XEPI 3: live 16-row blocks; tile row tm owns list entries tm * BM / 16 ..
  constexpr bool LIVE = XEPI == 3 || XEPI == 4;
  constexpr int LUNIT = XEPI == 4 ? 1 : 16;                 // rows per list entry
  if constexpr (LIVE) {
    // the launch is sized for every block being live; the XCD split is made over the tiles that exist (made over the nominal count,
    // the surplus tiles - the tail of the logical order - would all sit on the last XCDs and leave them idle)
    nlive = *ep.live_count;
    const int tml = (nlive + C::BM / LUNIT - 1) / (C::BM / LUNIT);      // live tile rows
    const int ntl = tml * tiles_n;
    if (ep.xcd_gc > 1) {
      // 2-D XCD split (EpiParams::xcd_gc): XCD x = blockIdx % 8 (as dispatched today; another placement changes speed only) works on row
      // group x / gc x column group x % gc, row-major inside it
      const int gc = ep.xcd_gc, gr = 8 / gc, x = blockIdx.x & 7;
      int j = blockIdx.x >> 3;
      const int rg = x / gc, cg = x - rg * gc, ncl = tiles_n / gc;
      const int r0 = (rg * tml) / gr, r1 = ((rg + 1) * tml) / gr;
      if (j >= (r1 - r0) * ncl) return;                     // (the whole workgroup leaves before any barrier)
      if (ep.cu_pair) j = cu_pair_local(j, (r1 - r0) * ncl);
      const int lr = j / ncl;
      tm = r0 + lr; tn = cg * ncl + (j - lr * ncl);
      tile = tm * tiles_n + tn;
    } else {
      if ((int)blockIdx.x >= ntl) {                         // (the whole workgroup leaves before any barrier)
#if RL_PROBES
        if constexpr (XEPI == 4) { if (ep.l2_prefetch) nt8_l2_prefetch<C>(A, lda, B, ldb, N, K, tiles_n, ntl, nlive, ep); }
#endif
        return;
      }
      tile = ep.cu_pair ? xcd_remap_paired(blockIdx.x, ntl) : xcd_remap(blockIdx.x, ntl);
      tile_coords(tile, tiles_n, ntl, group_m, tm, tn);
    }
  } else {
    tile_coords(tile, tiles_n, ntiles, group_m, tm, tn);
  }
  const int m0 = tm * C::BM, n0 = tn * C::BN;
  const int mlive = ep.m_dev != nullptr ? *ep.m_dev : M;
  if (!LIVE && m0 >= mlive) return;                         // device-side live-row count: the whole workgroup leaves before any barrier
  const int mzero = ep.m_exact ? mlive : M;                 // A rows at or beyond it read as zeros
  GruRows gru_rows;
  if constexpr (XEPI == 2) nt8_gru_prefetch<C>(ep, gru_rows, mlive, N / 3, m0, tn, wave / C::WN, wave % C::WN, lane);
  const int nk = (K + 63) >> 6;
  const int ktail = KTAIL ? (K & 63) : 0;   // elements of a ragged last K-tile (multiple of 8); KTAIL = false: K % 64 == 0

  // ---- this wave's pieces: LDS offset inside a stage (scalar) and per-lane source offset (row clamped into the matrix: rows
  //      beyond M / N only feed accumulators that the epilogue never stores)
  const int lrow = lane >> 3;
  const int kchunk_b = (((lane & 7) ^ lrow) << 4);            // byte offset of the 16-byte chunk this lane fetches (source-side swizzle)
  int lo[NPW];
  uint32_t go[NPW];
#pragma unroll
  for (int s = 0; s < NPW; ++s) {
    const int p = s * 8 + wave;
    int row, is_b;
    if (s < C::HPW) { row = p * 8; is_b = C::HOLD_B ? 1 : 0; }
    else {
      const int pp = p - C::HP, q = pp / C::GP, rem = pp - q * C::GP, slice = rem / (SQ * 2), j = rem - slice * (SQ * 2);
      row = slice * C::SR + q * SQ * 16 + j * 8; is_b = C::HOLD_B ? 0 : 1;
    }
    lo[s] = (is_b ? C::A_BYTES : 0) + row * 128;
    int grow = is_b ? min(n0 + row + lrow, N - 1) : min(m0 + row + lrow, M - 1);
    if constexpr (XEPI == 2) {
      if (is_b) {                                 // gate-interleaved column tile: [r | z | n] x 32 units for each of the two wave columns
        const int c = row + lrow, half = c / 96, cc = c - half * 96, gate = cc >> 5;
        grow = gate * (N / 3) + tn * 64 + half * 32 + (cc & 31);
      }
    }
    bool park = !LIVE && !is_b && m0 + row + lrow >= mzero;
    if constexpr (XEPI == 4) {
      if (!is_b) {                                  // the lane's row of the piece is the list's entry (tile row tm owns entries tm * BM ..)
        const int j = tm * C::BM + row + lrow;
        const int r = j < nlive ? ep.live_list[j] : -1;
        grow = r < 0 ? 0 : r;
        park = r < 0;                               // list exhausted inside the last tile: zeros
      }
    }
    if constexpr (XEPI == 3) {
      if (!is_b) {                                  // a piece is 8 rows of ONE 16-row block (row % 8 == 0): wave-uniform list entry
        const int j = tm * (C::BM / 16) + (row >> 4);
        const int blk = j < nlive ? ep.live_list[j] : -1;
        grow = (blk < 0 ? 0 : blk) * 16 + ((row + lrow) & 15);
        park = blk < 0;                             // list exhausted inside the last tile: zeros
      }
    }
    go[s] = (uint32_t)((int64_t)grow * (is_b ? ldb : lda) * 2 + kchunk_b);
    if (park) go[s] = 0xFFFFFF00u;                  // beyond num_records: the buffer range check returns zeros
  }
  // Fetches are raw-buffer LDS-DMA loads: resource descriptor + K-tile byte offset in SGPRs, the per-lane row/chunk offset in
  // ONE 32-bit VGPR per piece - no vector ALU work per fetch (a 64-bit flat address costs two v_lshl_add_u64 each, which showed
  // up as +40 % on the fetch-only probe).  A ragged last K-tile (K % 64 != 0) parks the chunks past K on an offset beyond
  // num_records: the buffer range check returns zeros for them.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(((int64_t)(M - 1) * lda + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(((int64_t)(N - 1) * ldb + K) * 2), 0x00020000);
  const bool lane_past_k = KTAIL && ktail != 0 && kchunk_b >= 2 * ktail;
  auto issue = [&](auto s_c, int stage, int ktile) {
    constexpr int s = decltype(s_c)::value;
    if constexpr (PROBE == 2) return;
    constexpr bool is_b = (s < C::HPW) ? C::HOLD_B : !C::HOLD_B;
    uint32_t voff = go[s];
    if constexpr (KTAIL) { if (ktail != 0 && ktile == nk - 1) voff = lane_past_k ? 0xFFFFFF00u : voff; }
    // (round 6: the B fetches marked non-temporal - aux = 2 - so that a co-resident workgroup's A lines outlive them in the L1: the family
    // 6.5 -> 7.2 ms/step; the weight tiles live on their L2 / L1 residency.  profiles/round6_ab.log)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(is_b ? rsB : rsA, (__attribute__((address_space(3))) void*)(smem + stage * C::STAGE + lo[s]), 16,
                                             voff, ktile * 128, 0, 0);
  };

  // ---- fragment addressing: lane part (row l15 of a 16-row tile, chunk ks*4+g swizzled by the row) + wave slice
  int fa[2], fb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int lane_sw = l15 * 128 + ((((ks << 2) + g) ^ (l15 & 7)) << 4);
    fa[ks] = wm * C::RM * 128 + lane_sw;
    fb[ks] = C::A_BYTES + wn * C::RN * 128 + lane_sw;
  }
  floatx4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  bf16x8_t hf[HT][2], sf[SQ][2];

  // ---- prologue: the pieces the steady state would have issued before phase 0
  static_for<C::PRO_TILES>([&](auto dt_c) {
    constexpr int dt = decltype(dt_c)::value;
    if (dt < nk) {
      static_for<NPW>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (C::in_prologue(dt, s)) issue(s_c, dt % NS, dt);
      });
    }
  });
  if (nk >= C::PRO_TILES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::VM_PRO) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }     // group 1 runs half a phase behind

  auto phase = [&](auto par_c, auto q_c, int t) {
    constexpr int PAR = decltype(par_c)::value, q = decltype(q_c)::value;
    constexpr int dt2 = (q + LEAD) / NPH, q2 = (q + LEAD) % NPH, SBASE = PAR * C::STAGE;
    // ---------------- memory segment
    constexpr int dtw = (q + LEAD - (C::ISSUE_C ? 1 : 0)) / NPH;       // tile of the last issue slot before this phase's wait
    const bool do_issue = t + dt2 < nk;
    auto issue_all = [&]() {
      if (do_issue) {
        static_for<NPW>([&](auto s_c) {
          constexpr int s = decltype(s_c)::value;
          if constexpr (s >= C::cum(q2) && s < C::cum(q2 + 1)) issue(s_c, (PAR + dt2) % NS, t + dt2);
        });
      }
    };
    if constexpr (C::ISSUE_AT == 2) { issue_all(); __builtin_amdgcn_sched_barrier(0); }
    if constexpr (PROBE != 3) {
      if constexpr (q == 0) {
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            hf[h][ks] = *(const bf16x8_t*)(smem + SBASE + h * 2048 + (C::HOLD_B ? fb[ks] : fa[ks]));
      }
#pragma unroll
      for (int i = 0; i < SQ; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          sf[i][ks] = *(const bf16x8_t*)(smem + SBASE + (q * SQ + i) * 2048 + (C::HOLD_B ? fa[ks] : fb[ks]));
    }
    if constexpr (C::ISSUE_AT == 0) { __builtin_amdgcn_sched_barrier(0); issue_all(); }
    if (t + dtw < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::vm(q)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- MFMA segment (ISSUE_C: this phase's fetches are spread between the MFMAs)
    constexpr int NM = 2 * SQ * HT, NI = C::cum(q2 + 1) - C::cum(q2);
    if constexpr (PROBE != 3) __builtin_amdgcn_s_setprio(1);
    static_for<NM>([&](auto m_c) {
      constexpr int m = decltype(m_c)::value, ks = m / (SQ * HT), i = (m / HT) % SQ, h = m % HT;
      if constexpr (PROBE != 3) {
        if constexpr (C::HOLD_B) acc[q * SQ + i][h] = Mma::mma(hf[h][ks], sf[i][ks], acc[q * SQ + i][h]);
        else acc[h][q * SQ + i] = Mma::mma(sf[i][ks], hf[h][ks], acc[h][q * SQ + i]);
      }
      if constexpr (C::ISSUE_C) {
        static_for<NI>([&](auto j_c) {
          constexpr int j = decltype(j_c)::value;
          if constexpr (m + 1 == ((j + 1) * NM) / (NI + 1)) {
            __builtin_amdgcn_sched_barrier(0);
            if (do_issue) issue(IC<C::cum(q2) + j>{}, (PAR + dt2) % NS, t + dt2);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      }
    });
    if constexpr (PROBE != 3) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int tb = 0; tb < nk; tb += NS) {
    static_for<NS>([&](auto par_c) {
      constexpr int PAR = decltype(par_c)::value;
      if (tb + PAR < nk) static_for<NPH>([&](auto q_c) { phase(par_c, q_c, tb + PAR); });
    });
  }
  if (grp == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }     // barrier census: group 1 took one extra up front

  if constexpr (XEPI == 1) {             // (the ring is dead: both groups passed the loop's last barrier)
    nt8_ln_epilogue<C>(smem, ep, acc, M, N, m0, n0, tm, tn, tiles_n, wm, wn, lane);
    return;
  }
  if constexpr (XEPI == 2) {
    nt8_gru_epilogue<C>(smem, ep, acc, gru_rows, mlive, N / 3, m0, tn, wave, wm, wn, lane);
    return;
  }
  // ---------------- epilogue: per-wave fp32 transpose through LDS, 8 consecutive columns (16 B of bf16) per lane
  constexpr int RS = C::RS, ER = C::ER, ITEMS = C::RN / 8, NIT = ER * ITEMS / 64;
  float* et = (float*)smem + wave * (ER * RS);
  const int row_w = m0 + wm * C::RM, col_w = n0 + wn * C::RN;
  // Round 5 (EpiParams::bias_first, realise_set_nt8p key 4): alpha and the bias go into the accumulators BEFORE the transposes.  In the
  // MFMA layout a lane holds four consecutive columns of one row per tile, so NT float4 of bias serve all its accumulators - loaded
  // once, together, while nothing else is pending - instead of two float4 per item behind that item's predecessors' stores (hipcc
  // waits with vmcnt(0) whenever loads and stores are pending together: every item of a wave paid a bias latency plus the drain of
  // the stores before it; the bias-only epilogues - qkv, FFN-up - now run their items without a single wait).  Same (acc * alpha) +
  // bias per element as epilogue8: same bits.
  EpiParams<bf16_t> epx = ep;
  // row-granular list: the output row of every item this lane will store (NCH chunks x NIT items), requested here - before the first
  // store, so that no tracked load meets a pending store later -; -1 beyond the list
  constexpr int NCH_ = C::RM / ER;
  int rid[XEPI == 4 ? NCH_ : 1][XEPI == 4 ? NIT : 1];
  if constexpr (XEPI == 4) {
#pragma unroll
    for (int c = 0; c < NCH_; ++c)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e = lane + 64 * it, r = e / ITEMS;
        const int jr = tm * C::BM + wm * C::RM + c * ER + r;
        rid[c][it] = jr < nlive ? ep.live_list[jr] : -1;
      }
  }
  if constexpr (!KTAIL) {
    if (ep.bias_first && ep.slab == nullptr) {
      if (ep.alpha != 1.0f) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] *= ep.alpha;
      }
      if (ep.bias != nullptr) {
        floatx4 bq[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bq[j] = *(const floatx4*)(ep.bias + min(col_w + j * 16 + 4 * g, N - 4));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] += bq[j];
      }
      epx.alpha = 1.0f; epx.bias = nullptr;
    }
  }
  // Round 5 (same knob): the residual / pre-activation / accumulated-output octets of ALL of the wave's items are requested here, before
  // the first store, from addresses clamped into the matrix (unconditional loads: nothing between them for the compiler to wait on);
  // epilogue8_pre then runs item by item on registers.  Two-per-CU shape only (RM / ER chunks x NIT items = 6 octets = 24 registers).
  if constexpr (!KTAIL && (XEPI == 0 || LIVE) && C::WGS == 2) {
    const bool use_aux = ep.mode == EPI_DROP_RESID || ep.mode == EPI_GELU_BWD;
    const bool use_old = ep.mode == EPI_STORE && ep.accumulate != 0;
    if (ep.bias_first && ep.slab == nullptr && ep.rm_hw_shift < 0 && (use_aux || use_old) && !(ep.mode == EPI_GELU_BWD && ep.accumulate) &&
        (!use_aux || ep.aux != nullptr) && (N % 8) == 0) {
      constexpr int NCH = C::RM / ER;
      const bf16_t* src = use_aux ? ep.aux : (const bf16_t*)ep.out;
      const int64_t ld = use_aux ? ep.ldaux : ep.ldo;
      int rcs[NCH];
      bool cok[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        rcs[c] = row_w + c * ER; cok[c] = true;
        if constexpr (XEPI == 3) {
          const int j = tm * (C::BM / 16) + ((wm * C::RM + c * ER) >> 4);
          cok[c] = j < nlive;
          rcs[c] = cok[c] ? ep.live_list[j] * 16 : 0;
        }
      }
      EpiParams<bf16_t> e3 = ep;
      e3.bias = nullptr; e3.alpha = 1.0f;                  // (both are in the accumulators already: bias_first)
      typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;      // (a register-class type: HIP's uint4 struct cannot be an asm operand)
      u32x4_t pre[NCH][NIT];
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int e = lane + 64 * it, r = e / ITEMS, c8 = e - r * ITEMS;
          int prow = rcs[c] + r;
          if constexpr (XEPI == 4) prow = max(rid[c][it], 0);
          pre[c][it] = *(const u32x4_t*)(src + (int64_t)min(prow, M - 1) * ld + min(col_w + c8 * 8, N - 8));
        }
      // first transpose under the loads' latency, then ONE unconditional wait for all six octets: left to their first uses - inside the
      // items' bounds-check branches - the compiler could not prove them landed on every path and re-waited, with stores pending, by vmcnt(0)
#pragma unroll
      for (int i = 0; i < ER / 16; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) *(floatx4*)(et + (i * 16 + l15) * RS + j * 16 + 4 * g) = acc[i][j];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(pre[c][it]));
#endif
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c > 0) {
#pragma unroll
          for (int i = 0; i < ER / 16; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) *(floatx4*)(et + (i * 16 + l15) * RS + j * 16 + 4 * g) = acc[c * (ER / 16) + i][j];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int e = lane + 64 * it, r = e / ITEMS, c8 = e - r * ITEMS;
          const floatx4 v0 = *(const floatx4*)(et + r * RS + c8 * 8), v1 = *(const floatx4*)(et + r * RS + c8 * 8 + 4);
          if constexpr (XEPI == 4) {
            if (rid[c][it] >= 0) epilogue8_pre(e3, M, N, rid[c][it], col_w + c8 * 8, v0, v1, __builtin_bit_cast(uint4, pre[c][it]));
          } else {
            if (cok[c]) epilogue8_pre(e3, M, N, rcs[c] + r, col_w + c8 * 8, v0, v1, __builtin_bit_cast(uint4, pre[c][it]));
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int c = 0; c < C::RM / ER; ++c) {
    int row_c = row_w + c * ER;
    if constexpr (XEPI == 3) {                      // the chunk's 16 rows are one listed block (or lie beyond the list: nothing to store)
      static_assert(!LIVE || ER == 16, "live-row form: one 16-row block per epilogue chunk");
      const int j = tm * (C::BM / 16) + ((wm * C::RM + c * ER) >> 4);
      if (j >= nlive) continue;
      row_c = ep.live_list[j] * 16;
    }
#pragma unroll
    for (int i = 0; i < ER / 16; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) *(floatx4*)(et + (i * 16 + l15) * RS + j * 16 + 4 * g) = acc[c * (ER / 16) + i][j];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = lane + 64 * it, r = e / ITEMS, c8 = e - r * ITEMS;
      const floatx4 v0 = *(const floatx4*)(et + r * RS + c8 * 8), v1 = *(const floatx4*)(et + r * RS + c8 * 8 + 4);
      if constexpr (XEPI == 4) {
        if (rid[c][it] >= 0) epilogue8<T>(epx, M, N, rid[c][it], col_w + c8 * 8, v0, v1);
      } else if (!KTAIL && ep.slab != nullptr) {
        const int row = row_c + r, col = col_w + c8 * 8;
        if (row < M && col < N) {
          float* o = ep.slab + (int64_t)split * ep.slab_stride + (int64_t)row * N + col;
          *(floatx4*)o = v0; *(floatx4*)(o + 4) = v1;
        }
      } else {
        epilogue8<T>(epx, M, N, row_c + r, col_w + c8 * 8, v0, v1);
      }
    }
  }
}


template <typename C, bool KTAIL = false, int XEPI = 0>
static int launch_nt8_cfg(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K,
                          const EpiParams<bf16_t>& ep) {
  const int tiles_m = (M + C::BM - 1) / C::BM, tiles_n = (N + C::BN - 1) / C::BN, ntiles = tiles_m * tiles_n;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt8_kernel<C, 0, KTAIL, XEPI>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
#if RL_PROBES
    if constexpr (!KTAIL) {
      (void)hipFuncSetAttribute((const void*)gemm_nt8_kernel<C, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
      (void)hipFuncSetAttribute((const void*)gemm_nt8_kernel<C, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    }
#endif
    attr_set = true;
  }
  ProfScope ps(st, PK_GEMM_NT, 2.0 * M * N * K);
  if (ep.m_dev != nullptr) prof_set_exec(ep.m_dev, 2.0 * N * K, C::BM, M);      // tiles that start at or beyond the count leave at once
  if constexpr (XEPI == 3) prof_set_exec(ep.live_count, 2.0 * N * K * 16.0, C::BM / 16, M / 16);      // (counted in 16-row blocks)
  if constexpr (XEPI == 4) prof_set_exec(ep.live_count, 2.0 * N * K, C::BM, M);                       // (counted in rows, whole tiles)
#if RL_PROBES
  if constexpr (!KTAIL) {
    if (g_nt8_probe == 2) { RL_LAUNCH((gemm_nt8_kernel<C, 2>), dim3(ntiles), dim3(512), C::LDS, st, A, lda, B, ldb, M, N, K, tiles_n, ntiles, g_nt8_group_m, ep); return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH; }
    if (g_nt8_probe == 3) { RL_LAUNCH((gemm_nt8_kernel<C, 3>), dim3(ntiles), dim3(512), C::LDS, st, A, lda, B, ldb, M, N, K, tiles_n, ntiles, g_nt8_group_m, ep); return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH; }
  }
#endif
  EpiParams<bf16_t> epk = ep;
  epk.bias_first = nt8_bias_first_on() && (N % 4) == 0 && N >= 4;
  epk.cu_pair = (g_nt8_cu_pair && C::WGS == 2 && (tiles_n % 2) == 0 && ep.ksplit <= 1) ? 1 : 0;
  epk.l2_prefetch = (XEPI == 4 && ep.xcd_gc <= 1) ? g_nt8_l2_prefetch : 0;
  int grid = ntiles * (KTAIL || ep.ksplit < 1 ? 1 : ep.ksplit);
  if constexpr (XEPI == 3 || XEPI == 4) { if (ep.xcd_gc > 1) grid = (tiles_m + 8 / ep.xcd_gc) * tiles_n; }      // every row group rounded up to whole tile rows
  RL_LAUNCH((gemm_nt8_kernel<C, 0, KTAIL, XEPI>), dim3(grid), dim3(512), C::LDS, st, A, lda, B, ldb, M, N, K, tiles_n, ntiles, g_nt8_group_m, epk);
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}


#if RL_PROBES
// =================================================================================================
// Warp-specialised form: 8 consumer waves + 4 LOADER waves (768 threads, 3 waves per SIMD, <= 168 VGPRs).
// tools/clock_probe.cpp measured why: a wave that both fetches and multiplies serialises the two in its in-order stream (LDS-DMA
// issue back-pressure and the vmcnt waits sit between its MFMAs; the 8-wave kernel's K-tile time is the SUM of its fetch-only and
// MFMA-only times), while MFMA waves next to dedicated fetch waves run both at once: 1510 TF + 45 B/clk/CU of fill on the same
// CUs against 1675 TF / 43 B/clk alone.  The consumers keep the two-group ping-pong (fragment reads of one group under the MFMAs of
// the other); the loaders walk the same phase sequence as group 0 with nothing but the phase's fetches and the counted wait in it,
// so the RAW / WAR phase rules of the schedule above hold unchanged (the loader's wait + barrier precede every read).
// =================================================================================================
// The loader role of the warp-specialised kernel (a separate __device__ function: as in-kernel generic lambdas next to the consumer's
// the host pass of hipcc 7.2 silently failed to instantiate the kernel stub).
template <typename C, int PROBE>
__device__ __forceinline__ void nt8ws_loader(char* smem, const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                             int M, int N, int K, int m0, int n0, int nk, int lw, int lane) {
  constexpr int NPW = C::NPW, NPH = C::NPH, NS = C::NS, LEAD = C::LEAD, SQ = C::SQ;
  const int lrow = lane >> 3;
  const int kchunk_b = (((lane & 7) ^ lrow) << 4);
  int lo[NPW];
  uint32_t go[NPW];
#pragma unroll
  for (int s = 0; s < NPW; ++s) {
    const int p = s * 4 + lw;
    int row, is_b;
    if (s < C::HPW) { row = p * 8; is_b = C::HOLD_B ? 1 : 0; }
    else {
      const int pp = p - C::HP, q = pp / C::GP, rem = pp - q * C::GP, slice = rem / (SQ * 2), j = rem - slice * (SQ * 2);
      row = slice * C::SR + q * SQ * 16 + j * 8; is_b = C::HOLD_B ? 0 : 1;
    }
    lo[s] = (is_b ? C::A_BYTES : 0) + row * 128;
    int grow = is_b ? min(n0 + row + lrow, N - 1) : min(m0 + row + lrow, M - 1);
    go[s] = (uint32_t)((int64_t)grow * (is_b ? ldb : lda) * 2 + kchunk_b);
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(((int64_t)(M - 1) * lda + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(((int64_t)(N - 1) * ldb + K) * 2), 0x00020000);
  auto issue = [&](auto s_c, int stage, int ktile) {
    constexpr int s = decltype(s_c)::value;
    if constexpr (PROBE == 2) return;
    constexpr bool is_b = (s < C::HPW) ? C::HOLD_B : !C::HOLD_B;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(is_b ? rsB : rsA, (__attribute__((address_space(3))) void*)(smem + stage * C::STAGE + lo[s]), 16,
                                             go[s], ktile * 128, 0, 0);
  };
  static_for<C::PRO_TILES>([&](auto dt_c) {
    constexpr int dt = decltype(dt_c)::value;
    if (dt < nk) {
      static_for<NPW>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (C::in_prologue(dt, s)) issue(s_c, dt % NS, dt);
      });
    }
  });
  if (nk >= C::PRO_TILES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::VM_PRO) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  auto lphase = [&](auto par_c, auto q_c, int t) {
    constexpr int PAR = decltype(par_c)::value, q = decltype(q_c)::value;
    constexpr int dt2 = (q + LEAD) / NPH, q2 = (q + LEAD) % NPH;
    if (t + dt2 < nk) {
      static_for<NPW>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (s >= C::cum(q2) && s < C::cum(q2 + 1)) issue(s_c, (PAR + dt2) % NS, t + dt2);
      });
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::vm(q)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  for (int tb = 0; tb < nk; tb += NS) {
    static_for<NS>([&](auto par_c) {
      constexpr int PAR = decltype(par_c)::value;
      if (tb + PAR < nk) static_for<NPH>([&](auto q_c) { lphase(par_c, q_c, tb + PAR); });
    });
  }
  __builtin_amdgcn_s_barrier();            // barrier census of consumer group 0
}

template <typename C, int PROBE>
__global__ void __launch_bounds__(768)
gemm_nt8ws_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int M, int N, int K, int tiles_n,
                  int ntiles, int group_m, EpiParams<bf16_t> ep) {
  typedef bf16_t T;
  typedef MmaBF16 Mma;
  static_assert(C::FW == 4 && C::ISSUE_AT == 0, "loader-wave configuration");
  constexpr int NPW = C::NPW, NPH = C::NPH, NS = C::NS, LEAD = C::LEAD, SQ = C::SQ, HT = C::HT, MT = C::MT, NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, ntiles);
  int tm, tn;
  tile_coords(tile, tiles_n, ntiles, group_m, tm, tn);
  const int m0 = tm * C::BM, n0 = tn * C::BN;
  const int nk = K >> 6;

  if (wave >= 8) {          // loader wave
    nt8ws_loader<C, PROBE>(smem, A, lda, B, ldb, M, N, K, m0, n0, nk, wave - 8, lane);
    return;
  }

  // -------------------------------------------------------------------------------------------- consumer wave
  const int g = lane >> 4, l15 = lane & 15;
  const int grp = wave >> 2;
  const int wm = wave / C::WN, wn = wave - wm * C::WN;
  int fa[2], fb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int lane_sw = l15 * 128 + ((((ks << 2) + g) ^ (l15 & 7)) << 4);
    fa[ks] = wm * C::RM * 128 + lane_sw;
    fb[ks] = C::A_BYTES + wn * C::RN * 128 + lane_sw;
  }
  floatx4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  bf16x8_t hf[HT][2], sf[SQ][2];
  if constexpr (PROBE == 4) {                // fetch + MFMA without the fragment reads: registers hold zeros
#pragma unroll
    for (int h = 0; h < HT; ++h) { hf[h][0] = bf16x8_t{}; hf[h][1] = bf16x8_t{}; }
#pragma unroll
    for (int i = 0; i < SQ; ++i) { sf[i][0] = bf16x8_t{}; sf[i][1] = bf16x8_t{}; }
  }
  __builtin_amdgcn_s_barrier();              // tile 0 landed (the loaders waited for it)
  asm volatile("" ::: "memory");
  if (grp == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }     // group 1 runs half a phase behind

  auto phase = [&](auto par_c, auto q_c) {
    constexpr int PAR = decltype(par_c)::value, q = decltype(q_c)::value, SBASE = PAR * C::STAGE;
    if constexpr (PROBE != 3 && PROBE != 4) {
      if constexpr (q == 0) {
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            hf[h][ks] = *(const bf16x8_t*)(smem + SBASE + h * 2048 + (C::HOLD_B ? fb[ks] : fa[ks]));
      }
#pragma unroll
      for (int i = 0; i < SQ; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          sf[i][ks] = *(const bf16x8_t*)(smem + SBASE + (q * SQ + i) * 2048 + (C::HOLD_B ? fa[ks] : fb[ks]));
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PROBE == 5) {                // fetch + fragment reads without the MFMAs
#pragma unroll
      for (int i = 0; i < SQ; ++i) { asm volatile("" ::"v"(sf[i][0]), "v"(sf[i][1])); }
      if constexpr (q == 0) {
#pragma unroll
        for (int h = 0; h < HT; ++h) asm volatile("" ::"v"(hf[h][0]), "v"(hf[h][1]));
      }
    }
    if constexpr (PROBE != 3 && PROBE != 5) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < SQ; ++i)
#pragma unroll
          for (int h = 0; h < HT; ++h) {
            if constexpr (C::HOLD_B) acc[q * SQ + i][h] = Mma::mma(hf[h][ks], sf[i][ks], acc[q * SQ + i][h]);
            else acc[h][q * SQ + i] = Mma::mma(sf[i][ks], hf[h][ks], acc[h][q * SQ + i]);
          }
      __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int tb = 0; tb < nk; tb += NS) {
    static_for<NS>([&](auto par_c) {
      constexpr int PAR = decltype(par_c)::value;
      if (tb + PAR < nk) static_for<NPH>([&](auto q_c) { phase(par_c, q_c); });
    });
  }
  if (grp == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

  constexpr int RS = C::RS, ER = C::ER, ITEMS = C::RN / 8, NIT = ER * ITEMS / 64;
  float* et = (float*)smem + wave * (ER * RS);
  const int row_w = m0 + wm * C::RM, col_w = n0 + wn * C::RN;
#pragma unroll
  for (int c = 0; c < C::RM / ER; ++c) {
#pragma unroll
    for (int i = 0; i < ER / 16; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) *(floatx4*)(et + (i * 16 + l15) * RS + j * 16 + 4 * g) = acc[c * (ER / 16) + i][j];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = lane + 64 * it, r = e / ITEMS, c8 = e - r * ITEMS;
      const floatx4 v0 = *(const floatx4*)(et + r * RS + c8 * 8), v1 = *(const floatx4*)(et + r * RS + c8 * 8 + 4);
      epilogue8<T>(ep, M, N, row_w + c * ER + r, col_w + c8 * 8, v0, v1);
    }
  }
}

template <typename C>
static int launch_nt8ws_cfg(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K,
                            const EpiParams<bf16_t>& ep) {
  const int tiles_m = (M + C::BM - 1) / C::BM, tiles_n = (N + C::BN - 1) / C::BN, ntiles = tiles_m * tiles_n;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt8ws_kernel<C, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt8ws_kernel<C, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt8ws_kernel<C, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt8ws_kernel<C, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt8ws_kernel<C, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    attr_set = true;
  }
  ProfScope ps(st, PK_GEMM_NT, 2.0 * M * N * K);
  if (g_nt8_probe == 2) RL_LAUNCH((gemm_nt8ws_kernel<C, 2>), dim3(ntiles), dim3(768), C::LDS, st, A, lda, B, ldb, M, N, K, tiles_n, ntiles, g_nt8_group_m, ep);
  else if (g_nt8_probe == 3) RL_LAUNCH((gemm_nt8ws_kernel<C, 3>), dim3(ntiles), dim3(768), C::LDS, st, A, lda, B, ldb, M, N, K, tiles_n, ntiles, g_nt8_group_m, ep);
  else if (g_nt8_probe == 4) RL_LAUNCH((gemm_nt8ws_kernel<C, 4>), dim3(ntiles), dim3(768), C::LDS, st, A, lda, B, ldb, M, N, K, tiles_n, ntiles, g_nt8_group_m, ep);
  else if (g_nt8_probe == 5) RL_LAUNCH((gemm_nt8ws_kernel<C, 5>), dim3(ntiles), dim3(768), C::LDS, st, A, lda, B, ldb, M, N, K, tiles_n, ntiles, g_nt8_group_m, ep);
  else RL_LAUNCH((gemm_nt8ws_kernel<C, 0>), dim3(ntiles), dim3(768), C::LDS, st, A, lda, B, ldb, M, N, K, tiles_n, ntiles, g_nt8_group_m, ep);
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}

#endif  // RL_PROBES

//             BM   BN  WM WN hold_B SQ NS LEAD issue_in_MFMA_segment
typedef Nt8Cfg<256, 256, 2, 4, true, 2, 2, 5> Cfg256x256;      // wave 128 x 64, 4 phases of 16 MFMAs
typedef Nt8Cfg<256, 192, 4, 2, false, 2, 2, 4> Cfg256x192;     // wave  64 x 96, 3 phases of 16 MFMAs
typedef Nt8Cfg<256, 128, 4, 2, true, 2, 3, 4> Cfg256x128;      // wave  64 x 64, 2 phases of 16 MFMAs, 3 stages
typedef Nt8Cfg<128, 192, 2, 4, true, 2, 3, 4> Cfg128x192;      // wave  64 x 48, 2 phases of 12 MFMAs, 3 stages
typedef Nt8Cfg<256, 256, 2, 4, true, 2, 2, 6, 1> Cfg256x256c;
typedef Nt8Cfg<256, 192, 4, 2, false, 2, 2, 5, 1> Cfg256x192c;
typedef Nt8Cfg<256, 128, 4, 2, true, 2, 3, 5, 1> Cfg256x128c;
typedef Nt8Cfg<128, 192, 2, 4, true, 2, 3, 5, 1> Cfg128x192c;
typedef Nt8Cfg<256, 256, 2, 4, true, 2, 2, 5, 2> Cfg256x256f;
typedef Nt8Cfg<256, 192, 4, 2, false, 2, 2, 4, 2> Cfg256x192f;
typedef Nt8Cfg<256, 128, 4, 2, true, 2, 3, 4, 2> Cfg256x128f;
typedef Nt8Cfg<128, 192, 2, 4, true, 2, 3, 4, 2> Cfg128x192f;
typedef Nt8Cfg<128, 192, 2, 4, true, 2, 2, 2, 0, 8, 2> Cfg128x192p;       // two workgroups per CU (80 KB each): one's epilogue and fetch
typedef Nt8Cfg<128, 192, 4, 2, false, 2, 2, 4, 0, 8, 2> Cfg128x192q;      //   waits run under the other's MFMAs
typedef Nt8Cfg<256, 192, 4, 2, false, 2, 2, 4, 0, 4> Cfg256x192w;     // 8 consumer + 4 loader waves
typedef Nt8Cfg<256, 128, 4, 2, true, 2, 3, 4, 0, 4> Cfg256x128w;
typedef Nt8Cfg<128, 192, 2, 4, true, 2, 3, 4, 0, 4> Cfg128x192w;

bool nt8_supported(int M, int N, int K, const EpiParams<bf16_t>& ep, int64_t lda, int64_t ldb) {
  if (ep.mode == EPI_AFFINE || ep.col_scale != nullptr) return false;      // (the 4-wave kernels' epilogues carry it)
  return (K % 8) == 0 && K >= 64 && (N % 8) == 0 && (ep.ldo % 8) == 0 && (ep.aux == nullptr || (ep.ldaux % 8) == 0) &&
         (lda % 8) == 0 && (ldb % 8) == 0 && M >= 1 && N >= 8 &&
         (int64_t)M * lda * 2 < 0xFFFFFF00ll && (int64_t)N * ldb * 2 < 0xFFFFFF00ll;
}

// K4: C = dropout(A . B^T + bias) + aux, then LayerNorm over the row (EpiParams::ln_*; ln_target = this launch's tag, != the previous
// launch's on the same ln_part buffer; ln_flag is not used by this form).  RL_ERR_ARG when the shape does not fit the
// fused form (the caller then runs the GEMM and the LayerNorm as two launches).
int gemm_nt8_ln(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, const EpiParams<bf16_t>& ep) {
  if (ep.mode != EPI_DROP_RESID || ep.ln_y == nullptr || ep.ln_gamma == nullptr || ep.ln_beta == nullptr || ep.ln_part == nullptr ||
      ep.ln_target == 0 || ep.aux == nullptr || ep.out == nullptr || ep.accumulate || ep.m_dev != nullptr || ep.slab != nullptr ||
      (M % 128) != 0 || (N % 192) != 0 || N / 192 > 8 || (K % 64) != 0 || ep.ldo != N || (ep.ldaux % 4) != 0 || ep.rm_hw_shift >= 0 ||
      (int64_t)M * N >= (1ll << 32) || !nt8_supported(M, N, K, ep, lda, ldb))
    return RL_ERR_ARG;
  // every workgroup of a row band must be resident at the same time: the bands' tiles are adjacent in the dispatch order and the
  // whole launch (M / 128 * N / 192 tiles) has to fit the chip's 512 two-per-CU slots
  if ((int64_t)(M / 128) * (N / 192) > 512) return RL_ERR_ARG;
  return launch_nt8_cfg<Cfg128x192q, false, 1>(st, A, lda, B, ldb, M, N, K, ep);
}

int gemm_nt8_gru(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, const EpiParams<bf16_t>& ep) {
  const int H = N / 3;
  if (ep.gru_table == nullptr || ep.gru_pho_idx == nullptr || ep.gru_perm == nullptr || ep.gru_lens == nullptr || ep.gru_hprev == nullptr ||
      ep.out == nullptr || ep.gru_out == nullptr || ep.bias == nullptr || ep.mode != EPI_STORE || ep.accumulate || ep.slab != nullptr ||
      ep.ln_y != nullptr || N != 3 * H || (H % 64) != 0 || (K % 64) != 0 || M < 1 || !nt8_supported(M, N, K, ep, lda, ldb))
    return RL_ERR_ARG;
  return launch_nt8_cfg<Cfg128x192q, false, 2>(st, A, lda, B, ldb, M, N, K, ep);
}

// the M dimension as a list of live 16-row blocks (EpiParams::live_list / live_count): see gemm.h
static int g_nt8_bias_first = 1;        // alpha / bias into the accumulators before the epilogue's transposes (realise_set_nt8p key 4; 0: per item, the round-4 form)
void set_nt8_epi_pre(int on) { g_nt8_bias_first = on; }
static int nt8_bias_first_on() { return g_nt8_bias_first; }
static int g_nt8_live_gc = 0;
static int g_nt8_live_big = 0;
void set_nt8_live_big(int v) { g_nt8_live_big = v; }
void set_nt8_live_gc(int gc) { g_nt8_live_gc = (gc == 1 || gc == 2 || gc == 4 || gc == 8) ? gc : 0; }
int gemm_nt8_live(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, const EpiParams<bf16_t>& ep) {
  if (ep.live_list == nullptr || ep.live_count == nullptr || ep.m_dev != nullptr || ep.slab != nullptr || ep.rm_hw_shift >= 0 ||
      ep.ln_y != nullptr || ep.gru_table != nullptr || (K % 64) != 0 || (M % 16) != 0 || M < 16 || !nt8_supported(M, N, K, ep, lda, ldb))
    return RL_ERR_ARG;
  EpiParams<bf16_t> e2 = ep;
  // XCD split.  One row band per XCD streams the WHOLE weight panel through each of the eight L2s; a panel beyond ~2 MB (N x K x 2 B:
  // 3.5 MB qkv, 4.7 MB FFN-up) does not stay resident next to the A rows and is re-read per wave of workgroups.  With column groups
  // the slice an XCD needs (panel / gc) stays resident and the A rows are read by gc XCDs: traffic gc x A + (8 / gc) x W - for the
  // wide K = 768 outputs (A 9 MB) two groups are the minimum, for the N = 768 outputs (A 27-36 MB at K = 2304 / 3072) one is.
  const int tiles_n = (N + 191) / 192;
  int gc = 1;
  if ((int64_t)N * K * 2 > (int64_t)(2 << 20) && (int64_t)M * K < (int64_t)N * K * 4) gc = 2;
  if (g_nt8_live_gc) gc = g_nt8_live_gc;
  while (gc > 1 && (tiles_n % gc) != 0) gc >>= 1;
  e2.xcd_gc = gc;
#if RL_PROBES
  if (ep.live_unit == 1 && g_nt8_live_big && N >= (g_nt8_live_big == 2 ? 2304 : 3072) && (N % 256) == 0) {
    // (round 6 measurement knob, probe build only, realise_set_nt8p(5, v): the wide outputs of a row-list launch on 256 x 256 one-per-CU
    // tiles - at the bench's ~5.3 k live rows FFN-up is 21 x 12 = 252 tiles, ONE round of the chip, where 128 x 192 tiles make 672 =
    // 1.3 rounds.  Measured: bit-identical, family 6.27 -> 6.70 ms/step, step 15.0 -> 15.3 ms: the one-per-CU tile exposes
    // its prologue and epilogue and keeps the other streams' kernels off the CU)
    int g2 = gc;
    const int tn2 = N / 256;
    while (g2 > 1 && (tn2 % g2) != 0) g2 >>= 1;
    e2.xcd_gc = g2;
    return launch_nt8_cfg<Cfg256x256, false, 4>(st, A, lda, B, ldb, M, N, K, e2);
  }
#endif
  if (ep.live_unit == 1) return launch_nt8_cfg<Cfg128x192q, false, 4>(st, A, lda, B, ldb, M, N, K, e2);      // list of rows
  if (ep.live_unit != 16) return RL_ERR_ARG;
  return launch_nt8_cfg<Cfg128x192q, false, 3>(st, A, lda, B, ldb, M, N, K, e2);
}

int gemm_nt8_splitk(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, int nsplit,
                    float* slab, int64_t slab_stride, const int* m_dev) {
  EpiParams<bf16_t> ep;
  ep.mode = EPI_STORE; ep.ldo = N; ep.m_dev = m_dev; ep.slab = slab; ep.slab_stride = slab_stride; ep.ksplit = nsplit;
  const int nk = K / 64;
  if (slab == nullptr || (K % 64) != 0 || nsplit < 1 || nsplit > 16 || (N % 8) != 0 || (lda % 8) != 0 || (ldb % 8) != 0 || M < 1 ||
      (int64_t)M * lda * 2 >= 0xFFFFFF00ll || (int64_t)N * ldb * 2 >= 0xFFFFFF00ll || slab_stride < (int64_t)M * N ||
      (nsplit - 1) * ((nk + nsplit - 1) / nsplit) >= nk)          // every K-range holds at least one K-tile
    return RL_ERR_ARG;
  return launch_nt8_cfg<Cfg128x192q>(st, A, lda, B, ldb, M, N, K, ep);
}

// tile: 0 = heuristic, 1 = 256x256, 2 = 256x192, 3 = 256x128, 4 = 128x192, 5 / 6 = 128x192 two workgroups per CU;
// +10: fetches issued inside the MFMA segments; 32..34: loader-wave kernels
int gemm_nt8(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K,
             const EpiParams<bf16_t>& ep, int tile) {
  if (!nt8_supported(M, N, K, ep, lda, ldb)) return RL_ERR_ARG;
  if (K % 64) return launch_nt8_cfg<Cfg128x192, true>(st, A, lda, B, ldb, M, N, K, ep);     // ragged K: one tile shape carries the tail code
  const int flavour = (tile / 10) * 10;      // 10..14: fetches issued between the MFMAs; 20..24: at the head of the memory segment
  tile -= flavour;
  if (tile == 0 && N < 4096) {
    // outputs up to a few thousand columns: the two-workgroups-per-CU shape.  Its smaller tile fetches 43 % more bytes per flop
    // than 256x192, but the second workgroup's MFMAs run under the first one's output write (a third of a K = 768 GEMM's time with
    // one workgroup per CU) and under its fetch waits: qkv 39.6 -> 35.0 us, ffn1+GELU 56.1 -> 51.3, attn-out 14.6 -> 14.3, the
    // K = 2304 / 3072 shapes equal (tools/nt8_probe.cpp ws, profiles/round2_nt8_probe.log)
    tile = 6;
    // Knob (off): outputs of at most one 128 x 192 tile per CU (N = 768: attention-output, FFN-down, the data gradients of qkv / FFN-up)
    // on the three-stage one-per-CU shape.  Alone, with operands coming from HBM, that shape is 4-8 % faster (23.1 vs 24.0 us at
    // K = 768 with dropout + residual, 48.5 vs 51.8 at K = 3072, 33.3 vs 36.3 at K = 2304: tools/nt8_probe.cpp cold); inside a step
    // it is 0.3-0.4 ms SLOWER (18.8 vs 18.45 ms, two A/B pairs on one box): its 120 KB of LDS keep the weight-gradient and branch
    // kernels of the other streams off the CU, which the 80 KB two-per-CU shape lets in.
    if (g_nt8_single_round && (long)((M + 127) / 128) * ((N + 191) / 192) <= 256) tile = 4;
  }
#if !RL_PROBES
  if (tile == 0) tile = 2;               // wide outputs the persistent kernel does not take: 256 x 192, one workgroup per CU
#endif
  if (tile == 0) {
    // (probe build) chip fill: rounds of 256 one-per-CU workgroups; among the shapes pick the least (rounds x MFMA time of one tile), ties to
    // the larger tile (fewer fetched bytes per flop).  The classifier (N = 21128) lands on 256x192: 303 us against 340 for the
    // two-per-CU shape, which is fetch-bound there.
    struct Cand { int id, bm, bn; } cands[4] = {{1, 256, 256}, {2, 256, 192}, {3, 256, 128}, {4, 128, 192}};
    double best = 1e30;
    for (const Cand& c : cands) {
      const long tiles = (long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
      const long rounds = (tiles + 255) / 256;
      const double fetch_pen = 1.0 + 24.0 / (2.0 * c.bm * c.bn / (double)(c.bm + c.bn));   // 256x256 -> 1.19, 128x192 -> 1.31
      const double cost = (double)rounds * c.bm * c.bn * fetch_pen;
      if (cost < best) { best = cost; tile = c.id; }
    }
  }
  switch (tile + flavour) {
    case 2: return launch_nt8_cfg<Cfg256x192>(st, A, lda, B, ldb, M, N, K, ep);
    case 6: return launch_nt8_cfg<Cfg128x192q>(st, A, lda, B, ldb, M, N, K, ep);
    case 4: return launch_nt8_cfg<Cfg128x192>(st, A, lda, B, ldb, M, N, K, ep);
#if RL_PROBES
    case 1: return launch_nt8_cfg<Cfg256x256>(st, A, lda, B, ldb, M, N, K, ep);
    case 3: return launch_nt8_cfg<Cfg256x128>(st, A, lda, B, ldb, M, N, K, ep);
    case 11: return launch_nt8_cfg<Cfg256x256c>(st, A, lda, B, ldb, M, N, K, ep);
    case 12: return launch_nt8_cfg<Cfg256x192c>(st, A, lda, B, ldb, M, N, K, ep);
    case 13: return launch_nt8_cfg<Cfg256x128c>(st, A, lda, B, ldb, M, N, K, ep);
    case 14: return launch_nt8_cfg<Cfg128x192c>(st, A, lda, B, ldb, M, N, K, ep);
    case 21: return launch_nt8_cfg<Cfg256x256f>(st, A, lda, B, ldb, M, N, K, ep);
    case 22: return launch_nt8_cfg<Cfg256x192f>(st, A, lda, B, ldb, M, N, K, ep);
    case 23: return launch_nt8_cfg<Cfg256x128f>(st, A, lda, B, ldb, M, N, K, ep);
    case 24: return launch_nt8_cfg<Cfg128x192f>(st, A, lda, B, ldb, M, N, K, ep);
    case 5: return launch_nt8_cfg<Cfg128x192p>(st, A, lda, B, ldb, M, N, K, ep);
    case 32: return launch_nt8ws_cfg<Cfg256x192w>(st, A, lda, B, ldb, M, N, K, ep);
    case 33: return launch_nt8ws_cfg<Cfg256x128w>(st, A, lda, B, ldb, M, N, K, ep);
    case 34: return launch_nt8ws_cfg<Cfg128x192w>(st, A, lda, B, ldb, M, N, K, ep);
#endif
    default: return RL_ERR_ARG;
  }
}

}  // namespace rl
