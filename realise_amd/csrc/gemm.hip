// MFMA GEMM kernels for gfx950 (CDNA4):
//   gemm_nt : C[M,N]  = A[M,K] . B[N,K]^T, fused epilogues   (Linear fwd, dgrad on W^T shadows,
//             classifier, GRU recurrent step, implicit-im2col conv fwd / dgrad)      K2 K4 K5 K6 K8 K12
//   gemm_tn : C[I,J] += sum_p A[p,i] B[p,j]  (weight gradients, K14)
// One code path for both numerics modes: T = bf16 (v_mfma_f32_16x16x32_bf16, speed mode) and
// T = float (v_mfma_f32_16x16x4_f32, exact-fp32 parity mode).
// Round 2: dense bf16 NT GEMMs with M >= 1024, N >= 256 go to the 8-wave kernels of gemm_nt8.hip (launch_nt routes); the four weight
// gradients of a transformer layer go out as one grouped TN launch (gemm_tn_group); block-1 64-channel convolutions go to the
// LDS-resident kernels (conv_c64_nt.hip, conv_wgrad_c64.hip); stride-2 data gradients run per parity class (ConvLoader::par).
//
// Structure (both kernels): 256-thread workgroup = WA x WB waves of a 64x64 output each (4x4 MFMA
// tiles per wave).  Operand tiles go HBM -> LDS directly with global_load_lds_dwordx4 (no VGPR round
// trip, no ds_write pass); the LDS image of one wave-instruction is lane-linear (1 KiB), so the XOR
// swizzle that makes the operand reads bank-conflict free is applied to the per-lane SOURCE address,
// and out-of-range chunks (row / K tails, conv zero padding) read a 16-byte zero page.  Two LDS
// stages, one barrier per K-tile: the loads of tile t+1 are in flight while tile t is multiplied.
// Operands are passed to the MFMA swapped (B-fragment first) so each lane ends up with 4 CONSECUTIVE
// OUTPUT COLUMNS of one row: epilogue loads/stores are 8-16 B per lane.
#include "gemm_dev.h"
#include "prof.h"

namespace rl {

// =================================================================================================
// NT.  WM x WN waves: (2,2) -> 128x128 tile; (4,1) -> 256x64 tile for the 64-channel glyph convs.
// K-tile = one 128-byte LDS row per operand row (64 bf16 / 32 f32), chunk c of row r stored at chunk
// (c ^ (r & 7)) (KTile): conflict-free ds_read_b128 operand reads.
// =================================================================================================
// Per-wave state of a gathered A operand: NA 8-row pieces per K-tile, one Row32 per piece, one Tap32 per lane.
template <typename L, int NA> struct ConvRows {
  __device__ __forceinline__ void init(const L&, int, int) {}
  __device__ __forceinline__ void issue(const L&, char*) {}
};
template <typename T, int NA> struct ConvRows<ConvLoader<T>, NA> {
  typename ConvLoader<T>::Row32 r[NA];
  typename ConvLoader<T>::Tap32 q;
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ void init(const ConvLoader<T>& la, int row0, int kchunk) {
#pragma unroll
    for (int j = 0; j < NA; ++j) r[j] = la.prepare32(row0 + j * 8);
    q = la.tap32(kchunk);
    rs = __builtin_amdgcn_make_buffer_rsrc((void*)la.src, 0, (int)ConvLoader<T>::RECORDS, 0x00020000);
  }
  __device__ __forceinline__ void issue(const ConvLoader<T>& la, char* lds_wave_base) {      // fetch this K-tile's pieces, step to the next
#pragma unroll
    for (int j = 0; j < NA; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds_wave_base + j * 1024), 16,
                                               la.voff32(r[j], q), 0, 0, 0);
    la.advance32(q, Geo<T>::BK);
  }
};

template <typename T, typename ALoader, int WM, int WN, int NSTAGE, int NF, bool SPREAD>
__global__ void __launch_bounds__(64 * WM * WN, (WM * WN == 4 && NSTAGE == 2) ? 2 : 1)
gemm_nt_kernel(ALoader la, DenseLoader<T> lb, int M, int N, int K, int tiles_n, int ntiles, EpiParams<T> ep) {
  typedef typename MmaOf<T>::type Mma;
  typedef Geo<T> G;
  constexpr int BM_ = 64 * WM, BN_ = 16 * NF * WN;       // each wave: 64 rows x (16*NF) columns
  constexpr int A_BYTES = BM_ * 128, B_BYTES = BN_ * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int NW = WM * WN;
  constexpr int NA = BM_ / (8 * NW), NB = BN_ / (8 * NW);   // 1-KiB (8-row) wave-instructions per wave per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform (SGPR): LDS bases stay scalar
  const int wm = wave / WN, wn = wave - wm * WN;
  // A device-side row bound (glyph dedup) shrinks the tile grid: the XCD remap is applied to the LIVE tiles only, so the
  // surplus workgroups exit at once and the live ones still spread over all 8 XCDs.
  la.clamp_rows();
  const int live = ((la.rows + BM_ - 1) / BM_) * tiles_n;
  if ((int)blockIdx.x >= live) return;
  const int tile = xcd_remap(blockIdx.x, live < ntiles ? live : ntiles);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM_, n0 = tn * BN_;
  const void* zero = (const void*)g_zero16;
  const int lrow = lane >> 3;                        // row inside the 8-row group written by one instruction
  const int kchunk = ((lane & 7) ^ lrow) * G::VEC;   // logical K offset of the chunk this lane fetches

  constexpr bool kDenseA = sizeof(typename ALoader::KPos) == sizeof(typename DenseLoader<T>::KPos);
  typename ALoader::Ctx actx[kDenseA ? NA : 1];
  typename DenseLoader<T>::Ctx bctx[NB];
  const int m0f = (RL_PROBES && ep.probe == 1) ? 0 : m0, n0f = (RL_PROBES && ep.probe == 1) ? 0 : n0;
  if constexpr (kDenseA) {
#pragma unroll
    for (int j = 0; j < NA; ++j) actx[j] = la.prepare(m0f + (wave * NA + j) * 8 + lrow);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) bctx[j] = lb.prepare(n0f + (wave * NB + j) * 8 + lrow);
  typename ALoader::KPos aq = la.kpos(kchunk);
  typename DenseLoader<T>::KPos bq = lb.kpos(kchunk);
  // gathered (implicit-im2col) A operand: 32-bit buffer addressing, see ConvLoader::Row32 / Tap32
  ConvRows<ALoader, NA> crow;
  if constexpr (!kDenseA) crow.init(la, m0f + wave * NA * 8 + lrow, kchunk);
  const int nk = (K + G::BK - 1) / G::BK;
  // Fast addressing for the K-tiles that lie completely inside K: each lane keeps one 64-bit source pointer per
  // wave-instruction and just adds the K-tile stride (0 for lanes parked on the zero page).  A ragged last tile
  // (K % BK != 0, e.g. the classifier's data gradient with K = 21128) falls back to checked addressing for that tile only.
  const int nfast = K / G::BK;
  const bool fast = nfast > 0, fast_all = (K % G::BK) == 0;
  int kt_issue = 0;
  const char* pa[NA];
  const char* pb[NB];
  int inca[NA], incb[NB];
  if (fast) {
    if constexpr (kDenseA) {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const void* p0 = la.addr(actx[j], aq, zero);
        pa[j] = (const char*)p0; inca[j] = (p0 != zero) ? G::BK * (int)sizeof(T) : 0;
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const void* p0 = lb.addr(bctx[j], bq, zero);
      pb[j] = (const char*)p0; incb[j] = (p0 != zero) ? G::BK * (int)sizeof(T) : 0;
    }
  }

  floatx4 acc[4][NF];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int stage) {                      // fetch the next K-tile, then advance the positions
    if (RL_PROBES && ep.probe == 2) return;
    char* base = smem + stage * STAGE;
    const bool f = kt_issue < nfast;                 // this tile is addressed by pointer stepping
    if constexpr (!kDenseA) {
      crow.issue(la, base + wave * NA * 1024);
    } else if (f) {
#pragma unroll
      for (int j = 0; j < NA; ++j) { glds16(pa[j], base + (wave * NA + j) * 1024); pa[j] += inca[j]; }
    } else {
      if (fast && kt_issue == nfast) aq = la.kpos(kchunk + nfast * G::BK);
#pragma unroll
      for (int j = 0; j < NA; ++j) glds16(la.addr(actx[j], aq, zero), base + (wave * NA + j) * 1024);
      la.advance(aq, G::BK);
    }
    if (f) {
#pragma unroll
      for (int j = 0; j < NB; ++j) { glds16(pb[j], base + A_BYTES + (wave * NB + j) * 1024); pb[j] += incb[j]; }
    } else {
      if (fast && kt_issue == nfast) bq = lb.kpos(kchunk + nfast * G::BK);
#pragma unroll
      for (int j = 0; j < NB; ++j) glds16(lb.addr(bctx[j], bq, zero), base + A_BYTES + (wave * NB + j) * 1024);
      lb.advance(bq, G::BK);
    }
    ++kt_issue;
  };

  // One 1-KiB piece of the next K-tile (fast addressing only): lets the main loop spread the NA + NB fetch instructions
  // between its MFMA groups instead of queueing them back to back behind the barrier.
  auto issue_piece = [&](int stage, int j) {
    char* base = smem + stage * STAGE;
    if constexpr (kDenseA) {
#pragma unroll
      for (int q = 0; q < NA; ++q)
        if (q == j) { glds16(pa[q], base + (wave * NA + q) * 1024); pa[q] += inca[q]; }
    }
#pragma unroll
    for (int q = 0; q < NB; ++q)
      if (NA + q == j) { glds16(pb[q], base + A_BYTES + (wave * NB + q) * 1024); pb[q] += incb[q]; }
  };

  // NSTAGE-deep ring: tiles kt .. kt+NSTAGE-2 are in flight while tile kt is awaited.  LDS-DMA completion is
  // tracked by vmcnt, so the wait is a COUNTED vmcnt (leave the younger tiles outstanding) followed by a raw
  // s_barrier (a __syncthreads() here would drain vmcnt to 0 and serialise the pipeline).
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue(s);
  for (int kt = 0; kt < nk; ++kt) {
    if (NSTAGE == 2 || kt + 1 >= nk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (NSTAGE == 3 || kt + 2 >= nk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");          // one younger tile stays in flight
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NA + NB)) : "memory");    // two younger tiles stay in flight
    }
    __builtin_amdgcn_s_barrier();                     // tile kt landed for every wave; stage (kt-1) % NSTAGE is free
    asm volatile("" ::: "memory");
    const bool more = kt + NSTAGE - 1 < nk;
    const int nstage = (kt + NSTAGE - 1) % NSTAGE;
    constexpr bool kSpread = SPREAD && kDenseA && sizeof(T) == 2;
    const bool spread = kSpread && fast_all && (!RL_PROBES || ep.probe == 0);
    if (more && !spread) issue(nstage);
    const char* As = smem + (kt % NSTAGE) * STAGE;
    const char* Bs = As + A_BYTES;
    if (RL_PROBES && ep.probe == 3) continue;
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks) {
      typename Mma::Frag a[4], b[NF];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = ktile_frag<T, G::BK>(As, wm * 64 + i * 16 + l15, ks, g);
#pragma unroll
      for (int j = 0; j < NF; ++j) b[j] = ktile_frag<T, G::BK>(Bs, wn * 16 * NF + j * 16 + l15, ks, g);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = Mma::mma(b[j], a[i], acc[i][j]);
        if constexpr (kSpread) {
          if (spread && more) {
            constexpr int SLOTS = 4 * G::KSTEPS, PER = (NA + NB + SLOTS - 1) / SLOTS;
#pragma unroll
            for (int u = 0; u < PER; ++u) issue_piece(nstage, (ks * 4 + i) * PER + u);
            __builtin_amdgcn_sched_barrier(0);          // keep the piece between this MFMA group and the next
          }
        }
      }
    }
  }
  if (ep.wide) {
    // Each wave transposes its 64 x 16NF result through a private, padded fp32 LDS tile (row stride 16NF + 4 floats:
    // the 8-lane groups of ds_write_b128 land on 8 distinct 4-bank sets) and leaves with 8 consecutive columns per lane.
    constexpr int RS = 16 * NF + 4, ITEMS = 2 * NF;            // floats per row ; 8-column items per row
    __syncthreads();                                           // every wave is done reading the last K-tile
    float* et = (float*)smem + wave * (64 * RS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) *(floatx4*)(et + (i * 16 + l15) * RS + j * 16 + 4 * g) = acc[i][j];
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
      const int e = lane + 64 * t, r = e / ITEMS, c8 = e - r * ITEMS;
      const floatx4 v0 = *(const floatx4*)(et + r * RS + c8 * 8), v1 = *(const floatx4*)(et + r * RS + c8 * 8 + 4);
      epilogue8<T>(ep, M, N, m0 + wm * 64 + r, n0 + wn * 16 * NF + c8 * 8, v0, v1);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j)
      epilogue4<T>(ep, M, N, m0 + wm * 64 + i * 16 + l15, n0 + wn * 16 * NF + j * 16 + 4 * g, acc[i][j]);
}

static int g_tn_jmajor = 0;      // (measured: family 2.585 against 2.545 ms/step i-major, three pairs - fewer HBM bytes, not faster: off)
void set_tn_jmajor(int on) { g_tn_jmajor = on; }
static int g_nt_n96 = 1;             // allow 128x96 tiles (tuning / A-B knob)
void set_nt_allow_n96(int on) { g_nt_n96 = on; }
static int g_nt_probe = 0, g_nt_wide = 1, g_nt_variant = 0;
static int g_conv_c64 = 1;          // block-1 conv2 (64 channels, 16x16 maps) through the LDS-resident kernels conv_c64_nt.hip / conv_wgrad_c64.hip
void set_conv_c64(int on) { g_conv_c64 = on; }
void set_nt_variant(int v) { g_nt_variant = (RL_PROBES || v == 0 || v == 9 || v == 12 || v == 14 || v == 16 || v >= 50) ? v : 0; }     // production: 0, 9 (4-wave), 12 / 14 / 16 (the three shipped 8-wave tiles), 50 / 51 (persistent on / off)
static int g_tn_probe = 0, g_tn_split = 0;
void set_tn_split(int n) { g_tn_split = n; }
void set_tn_probe(int mode) { g_tn_probe = RL_PROBES ? mode : 0; }
void set_nt_wide_epilogue(int on) { g_nt_wide = on; }
void set_nt_probe(int mode) { g_nt_probe = RL_PROBES ? mode : 0; set_nt8_probe(RL_PROBES ? mode : 0); }


#if RL_PROBES
// =================================================================================================
// Phase-shifted 8-wave NT kernel (EXPERIMENTAL, variant 8 of realise_set_nt_variant): 256 x 128 tile, waves
// 0-3 (group A) and 4-7 (group B) share the SIMDs pairwise and run half a K-tile out of phase - one group's MFMA phase
// (32 MFMAs on register-resident fragments) covers the other's memory phase (fragment reads of a whole K-tile + its share
// of the LDS-DMA fetches of tile t+2).  bf16, dense operands, K % 64 == 0.
// =================================================================================================
__global__ void __launch_bounds__(512, 1)
gemm_nt_pp_kernel(DenseLoader<bf16_t> la, DenseLoader<bf16_t> lb, int M, int N, int K, int tiles_n, int ntiles, EpiParams<bf16_t> ep) {
  typedef bf16_t T;
  typedef MmaBF16 Mma;
  constexpr int BM_ = 256, BN_ = 128, BK = 64, NST = 3;
  constexpr int A_BYTES = BM_ * 128, B_BYTES = BN_ * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int NA = 4, NB = 2;                       // 1-KiB pieces per wave per tile (8 waves)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wm = wave & 3, wn = grp;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM_, n0 = tn * BN_;
  const void* zero = (const void*)g_zero16;
  const int lrow = lane >> 3;
  const int kchunk = ((lane & 7) ^ lrow) * 8;
  const char* pa[NA];
  const char* pb[NB];
  int inca[NA], incb[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const void* p0 = la.addr(la.prepare(m0 + (wave * NA + j) * 8 + lrow), la.kpos(kchunk), zero);
    pa[j] = (const char*)p0; inca[j] = (p0 != zero) ? BK * 2 : 0;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const void* p0 = lb.addr(lb.prepare(n0 + (wave * NB + j) * 8 + lrow), lb.kpos(kchunk), zero);
    pb[j] = (const char*)p0; incb[j] = (p0 != zero) ? BK * 2 : 0;
  }
  floatx4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  bf16x8_t fa[2][4], fb[2][4];

  auto issue_tile = [&](int stage) {
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < NA; ++j) { glds16(pa[j], base + (wave * NA + j) * 1024); pa[j] += inca[j]; }
#pragma unroll
    for (int j = 0; j < NB; ++j) { glds16(pb[j], base + A_BYTES + (wave * NB + j) * 1024); pb[j] += incb[j]; }
  };
  auto read_frags = [&](int stage) {
    const char* As = smem + stage * STAGE;
    const char* Bs = As + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[ks][i] = ktile_frag<T, 64>(As, wm * 64 + i * 16 + l15, ks, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[ks][j] = ktile_frag<T, 64>(Bs, wn * 64 + j * 16 + l15, ks, g);
    }
  };
  auto mma_all = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mma::mma(fb[ks][j], fa[ks][i], acc[i][j]);
  };
#define RL_PP_BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

  const int nk = K / BK;
  issue_tile(0);
  if (nk > 1) { issue_tile(1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RL_PP_BARRIER();                                   // tile 0 landed for every wave
  if (grp == 0) {                                    // half 1(0): group A's first memory phase
    read_frags(0);
    if (2 < nk) issue_tile(2 % NST);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  RL_PP_BARRIER();
  for (int t = 0; t < nk; ++t) {
    // half 2(t): A multiplies tile t, B reads its fragments of tile t and fetches its share of tile t+2
    if (grp == 0) {
      mma_all();
    } else {
      read_frags(t % NST);
      if (t + 2 < nk) issue_tile((t + 2) % NST);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // own pieces of tile t+1 done (t+2 may fly)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RL_PP_BARRIER();
    // half 1(t+1): A reads tile t+1 and fetches its share of tile t+3, B multiplies tile t
    if (grp == 0) {
      if (t + 1 < nk) {
        read_frags((t + 1) % NST);
        if (t + 3 < nk) issue_tile((t + 3) % NST);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {
      mma_all();
    }
    RL_PP_BARRIER();
  }
#undef RL_PP_BARRIER
  if (ep.wide) {
    constexpr int RS = 68, ITEMS = 8;
    __syncthreads();
    float* et = (float*)smem + wave * (64 * RS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) *(floatx4*)(et + (i * 16 + l15) * RS + j * 16 + 4 * g) = acc[i][j];
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
      const int e = lane + 64 * t, r = e / ITEMS, c8 = e - r * ITEMS;
      const floatx4 v0 = *(const floatx4*)(et + r * RS + c8 * 8), v1 = *(const floatx4*)(et + r * RS + c8 * 8 + 4);
      epilogue8<T>(ep, M, N, m0 + wm * 64 + r, n0 + wn * 64 + c8 * 8, v0, v1);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) epilogue4<T>(ep, M, N, m0 + wm * 64 + i * 16 + l15, n0 + wn * 64 + j * 16 + 4 * g, acc[i][j]);
}
static int launch_nt_pp(hipStream_t st, const DenseLoader<bf16_t>& la, const DenseLoader<bf16_t>& lb, int M, int N, int K,
                        const EpiParams<bf16_t>& ep) {
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 127) / 128, ntiles = tiles_m * tiles_n;
  const size_t lds = 3 * (size_t)(256 + 128) * 128;      // 144 KB ring >= 8 x 64 x 68 x 4 epilogue tiles
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  ProfScope ps(st, PK_GEMM_NT, 2.0 * M * N * K);
  EpiParams<bf16_t> epp = ep;
  epp.wide = g_nt_wide && (N % 8 == 0) && (ep.ldo % 8 == 0) && (ep.aux == nullptr || ep.ldaux % 8 == 0);
  RL_LAUNCH(gemm_nt_pp_kernel, dim3(ntiles), dim3(512), lds, st, la, lb, M, N, K, tiles_n, ntiles, epp);
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}

#endif  // RL_PROBES

template <typename T, typename ALoader, int WM, int WN, int NSTAGE = 2, int NF = 4, bool SPREAD = false>
static int launch_nt_tile(hipStream_t st, const ALoader& la, const DenseLoader<T>& lb, int M, int N, int K, const EpiParams<T>& ep) {
  constexpr int BM_ = 64 * WM, BN_ = 16 * NF * WN;
  const int tiles_m = (M + BM_ - 1) / BM_, tiles_n = (N + BN_ - 1) / BN_;
  const int ntiles = tiles_m * tiles_n;
  const size_t ring = NSTAGE * (size_t)(BM_ + BN_) * 128, etile = (size_t)WM * WN * 64 * (16 * NF + 4) * 4;
  // (NF = 6, round 6 probe: 128 x 192 on FOUR waves of 64 x 96 - the ring alone, 80 KB, so that two workgroups share a CU as the 8-wave
  // 128 x 192 kernel's do; its epilogue then stores straight from the accumulator layout)
  const size_t lds = (NF == 6 || ring > etile) ? ring : etile;        // the epilogue re-uses the ring as per-wave transpose tiles
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, ALoader, WM, WN, NSTAGE, NF, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  ProfScope ps(st, sizeof(typename ALoader::KPos) == sizeof(typename DenseLoader<T>::KPos) ? PK_GEMM_NT : PK_CONV_NT, 2.0 * M * N * K);
  if constexpr (sizeof(typename ALoader::KPos) == sizeof(typename DenseLoader<T>::KPos))      // (the conv families are scaled by the live-glyph fraction in bench.py)
    if (la.rows_dev != nullptr) prof_set_exec(la.rows_dev, 2.0 * N * K, BM_, M);   // device-side row bound: surplus tiles exit at once
  EpiParams<T> epp = ep;
  epp.probe = g_nt_probe;
  epp.wide = NF != 6 && g_nt_wide && (N % 8 == 0) && (ep.ldo % 8 == 0) && (ep.aux == nullptr || ep.ldaux % 8 == 0);
  RL_LAUNCH((gemm_nt_kernel<T, ALoader, WM, WN, NSTAGE, NF, SPREAD>), dim3(ntiles), dim3(64 * WM * WN), lds, st, la, lb, M, N, K, tiles_n, ntiles, epp);
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}

template <typename T, typename ALoader>
static int launch_nt(hipStream_t st, const ALoader& la, const T* B, int64_t ldb, int M, int N, int K, const EpiParams<T>& ep) {
  if (M <= 0 || N <= 0 || K <= 0) return RL_OK;
  if ((N & 3) || (K % Geo<T>::VEC) || (ldb % Geo<T>::VEC)) return RL_ERR_ARG;
  DenseLoader<T> lb{B, ldb, N, K};
  if (N <= 64) return launch_nt_tile<T, ALoader, 4, 1>(st, la, lb, M, N, K, ep);
  // Tile width by chip balance: 256 CUs x 2 resident workgroups.  The column count per tile (128 or 96) is chosen to
  // minimise  width x f(tiles per CU)  with f(1) = 1 and f(n) = 0.75 n (two co-resident tiles overlap each other's
  // stalls): e.g. M = 8192, N = 768 -> 8 x 96 columns = 512 tiles, two per CU on every CU, instead of 384 tiles of
  // 128 columns (half the CUs run two, half run one): +25 % measured (tools/nt_probe.cpp, n96 knob).
  auto cost = [&](int bn) {
    const long tiles = (long)((M + 127) / 128) * ((N + bn - 1) / bn);
    const long tpc = (tiles + 255) / 256;
    return bn * (tpc == 1 ? 1.0 : 0.75 * (double)tpc);
  };
  if constexpr (sizeof(typename ALoader::KPos) == sizeof(typename DenseLoader<T>::KPos) && sizeof(T) == 2) {
    // Production path for the big dense GEMMs: the ping-pong 8-wave kernel (gemm_nt8.hip), tile by chip fill.  Variants 10..34
    // force one of its tiles / issue flavours, variant 9 forces the 4-wave kernel below (tools/nt8_probe.cpp).
    // Wide outputs (qkv, FFN-up + GELU, FFN-down data gradient + GELU', the classifier: >= 1.5 tiles of 256 x 192 per CU): the persistent
    // kernel, whose register epilogue and next-tile prologue overlap what the one-tile kernels leave exposed (gemm_nt8p.hip).
    // Variant 50 forces it wherever it is supported, variant 51 keeps it off.
    if (la.rows_dev == nullptr && g_nt_probe == 0 && (g_nt_variant == 50 || ((g_nt_variant == 0 || (g_nt_variant == 53 && ep.mode != EPI_GELU_BWD)) && M >= 1024 && (long)((M + 255) / 256) * ((N + 191) / 192) >= 384)) &&
        nt8p_supported(M, N, K, ep, la.ld, ldb))
      return gemm_nt8p(st, la.base, la.ld, B, ldb, M, N, K, ep);
    if (ep.out_f32 != nullptr) return RL_ERR_ARG;        // (the fp32 copy exists in the persistent kernel's epilogue only: the caller casts)
    if (la.rows_dev == nullptr && g_nt_probe != 1 && ((g_nt_variant == 0 && M >= 1024 && N >= 256) || (g_nt_variant >= 10 && g_nt_variant <= 44) || (g_nt_variant >= 50 && g_nt_variant < 60))) {
      if (nt8_supported(M, N, K, ep, la.ld, ldb)) return gemm_nt8(st, la.base, la.ld, B, ldb, M, N, K, ep, (g_nt_variant >= 10 && g_nt_variant < 50) ? g_nt_variant - 10 : 0);
    }
    // a device-side row bound on a wide output (round 6: the training classifier over the loss rows): the persistent kernel, which
    // shrinks its tile walk to the live tile rows on the device
    if (la.rows_dev != nullptr && ep.m_dev == nullptr && g_nt_probe == 0 && g_nt_variant == 0 && M >= 1024 &&
        (long)((M + 255) / 256) * ((N + 191) / 192) >= 384 && nt8p_supported(M, N, K, ep, la.ld, ldb)) {
      EpiParams<T> e2 = ep;
      e2.m_dev = la.rows_dev;
      return gemm_nt8p(st, la.base, la.ld, B, ldb, M, N, K, e2);
    }
    // a device-side row bound (the GRU steps of a device-built batch: nominal M = B*S, the alive count lives on the device): the
    // 8-wave kernel with exact row masking - tiles beyond the count leave at once, rows beyond it inside the last live tile read zeros
    if (la.rows_dev != nullptr && ep.m_dev == nullptr && g_nt_probe == 0 && g_nt_variant == 0 && M >= 1024 && N >= 256 && (K % 64) == 0 &&
        nt8_supported(M, N, K, ep, la.ld, ldb)) {
      EpiParams<T> e2 = ep;
      e2.m_dev = la.rows_dev; e2.m_exact = 1;
      return gemm_nt8(st, la.base, la.ld, B, ldb, M, N, K, e2, 0);
    }
#if RL_PROBES
    switch (g_nt_variant) {       // experimental tile shapes (tools/nt_probe.cpp)
      case 1: return launch_nt_tile<T, ALoader, 2, 4, 2, 3>(st, la, lb, M, N, K, ep);     // 128 x 192, 8 waves, 2 stages
      case 2: return launch_nt_tile<T, ALoader, 2, 4, 3, 3>(st, la, lb, M, N, K, ep);     // 128 x 192, 3 stages
      case 3: return launch_nt_tile<T, ALoader, 4, 2, 2, 4>(st, la, lb, M, N, K, ep);     // 256 x 128, 2 stages
      case 4: return launch_nt_tile<T, ALoader, 4, 2, 3, 4>(st, la, lb, M, N, K, ep);     // 256 x 128, 3 stages
      case 5: if (g_nt_n96 && cost(96) < cost(128)) return launch_nt_tile<T, ALoader, 2, 2, 2, 3, true>(st, la, lb, M, N, K, ep);
              return launch_nt_tile<T, ALoader, 2, 2, 2, 4, true>(st, la, lb, M, N, K, ep);   // production tiles, spread fetch issue
      case 6: return launch_nt_tile<T, ALoader, 4, 2, 3, 4, true>(st, la, lb, M, N, K, ep);     // 256 x 128, 3 stages, spread
      case 7: return launch_nt_tile<T, ALoader, 2, 4, 3, 3, true>(st, la, lb, M, N, K, ep);     // 128 x 192, 3 stages, spread
      // round 6: 128 x 192 on four waves of 64 x 96, two workgroups per CU - 37 % fewer fragment bytes read
      // from LDS per K-tile than the 8-wave 128 x 192 kernel (80 against 128 KB); 60 = fetches behind the barrier, 61 = spread
      case 60: return launch_nt_tile<T, ALoader, 2, 2, 2, 6>(st, la, lb, M, N, K, ep);
      case 61: return launch_nt_tile<T, ALoader, 2, 2, 2, 6, true>(st, la, lb, M, N, K, ep);
      case 8: if ((K % 64) == 0 && la.rows_dev == nullptr) return launch_nt_pp(st, la, lb, M, N, K, ep);     // phase-shifted 256 x 128
              break;
      default: break;
    }
#endif
    // very wide outputs (the 21128-column classifier: > 20 rounds of 128x128 tiles): 8-wave 256x128 tiles cut the
    // operand traffic per flop by a third; three stages and the fetches spread between the MFMA groups keep the single
    // resident workgroup fed (730 vs 644 TF, tools/nt_probe.cpp).  Below ~8 rounds the 4-wave tiles win.
    if ((long)((M + 127) / 128) * ((N + 127) / 128) >= 4096 && (K % Geo<T>::BK) == 0)
      return launch_nt_tile<T, ALoader, 4, 2, 3, 4, true>(st, la, lb, M, N, K, ep);
  }
  if (g_nt_n96 && cost(96) < cost(128)) return launch_nt_tile<T, ALoader, 2, 2, 2, 3>(st, la, lb, M, N, K, ep);
  return launch_nt_tile<T, ALoader, 2, 2>(st, la, lb, M, N, K, ep);
}

template <typename T>
int gemm_nt(hipStream_t st, const T* A, int64_t lda, const T* B, int64_t ldb, int M, int N, int K, const EpiParams<T>& ep,
            const int* rows_dev) {
  if (lda % Geo<T>::VEC) return RL_ERR_ARG;
  if (ep.out_f32 != nullptr && sizeof(T) != 2) return RL_ERR_ARG;
  if (ep.col_scale != nullptr && ep.mode != EPI_AFFINE) return RL_ERR_ARG;
  DenseLoader<T> la{A, lda, M, K};
  la.rows_dev = rows_dev;
  return launch_nt<T, DenseLoader<T>>(st, la, B, ldb, M, N, K, ep);
}
template <typename T>
int gemm_nt_conv(hipStream_t st, const ConvLoader<T>& la_, const T* B, int64_t ldb, int M, int N, int K, const EpiParams<T>& ep) {
  if (la_.C % Geo<T>::VEC || ep.out_f32 != nullptr || (ep.col_scale != nullptr && ep.mode != EPI_AFFINE)) return RL_ERR_ARG;
  ConvLoader<T> la = la_;
  la.finalize();
  if (la.K != K || !la.span_ok()) return RL_ERR_ARG;
  if (la.par >= 0 && (la.par > 3 || la.mode != 1 || la.stride != 2 || la.hw_shift < 2 || la.w_shift < 1 || la.img_index != nullptr)) return RL_ERR_ARG;
  if constexpr (sizeof(T) == 2) {
    // 64 -> 64 channels, 3x3 / stride 1 / pad 1 on 16x16 maps (glyph ResNet block 1), plain store: the LDS-resident image + weights kernel
    if (g_conv_c64 && g_nt_probe == 0 && la.par < 0 && la.C == 64 && la.KH == 3 && la.KW == 3 && la.stride == 1 && la.pad == 1 && la.mode <= 1 &&
        la.Hr == 16 && la.Wr == 16 && la.Hs == 16 && la.Ws == 16 && la.img_index == nullptr && N == 64 && K == 576 && ldb == 576 &&
        M == la.rows && (M % 256) == 0 && (int64_t)M * 128 < 0xFFFFFE00ll && ep.out2 == nullptr && ep.alpha == 1.0f && ep.ldo == 64 && ep.rm_hw_shift < 0) {
      if (ep.mode == EPI_STORE && ep.accumulate == 0 && ep.bias == nullptr && ep.col_scale == nullptr)
        return conv_c64_nt(st, la.src, B, ep.out, M, la.rows_dev, la.mode);
      if (ep.mode == EPI_AFFINE && ep.col_scale != nullptr && ep.bias != nullptr && (ep.aux == nullptr || ep.ldaux == 64))
        return conv_c64_nt(st, la.src, B, ep.out, M, la.rows_dev, la.mode, ep.col_scale, ep.bias, ep.aux, ep.relu);
    }
  }
  return launch_nt<T, ConvLoader<T>>(st, la, B, ldb, M, N, K, ep);
}
// hipcc (ROCm 7.2) drops the implicit instantiation of the default-tile conv kernel once the kernel body holds the ConvRows
// helper (the host object keeps an undefined reference): instantiate the gathered-operand kernels explicitly.
#define RL_INST_NT_CONV(T, WM, WN, NS, NF) \
  template __global__ void gemm_nt_kernel<T, ConvLoader<T>, WM, WN, NS, NF, false>(ConvLoader<T>, DenseLoader<T>, int, int, int, int, int, EpiParams<T>);
RL_INST_NT_CONV(bf16_t, 2, 2, 2, 4) RL_INST_NT_CONV(bf16_t, 2, 2, 2, 3) RL_INST_NT_CONV(bf16_t, 4, 1, 2, 4)
RL_INST_NT_CONV(float, 2, 2, 2, 4) RL_INST_NT_CONV(float, 2, 2, 2, 3) RL_INST_NT_CONV(float, 4, 1, 2, 4)
#undef RL_INST_NT_CONV
template int gemm_nt<bf16_t>(hipStream_t, const bf16_t*, int64_t, const bf16_t*, int64_t, int, int, int, const EpiParams<bf16_t>&, const int*);
template int gemm_nt<float>(hipStream_t, const float*, int64_t, const float*, int64_t, int, int, int, const EpiParams<float>&, const int*);
template int gemm_nt_conv<bf16_t>(hipStream_t, const ConvLoader<bf16_t>&, const bf16_t*, int64_t, int, int, int, const EpiParams<bf16_t>&);
template int gemm_nt_conv<float>(hipStream_t, const ConvLoader<float>&, const float*, int64_t, int, int, int, const EpiParams<float>&);

// =================================================================================================
// TN: C[I,J] += sum_p A[p,i] * B[p,j].  Operand tiles are staged in their natural layout
// ([p][feature], feature contiguous) and transposed on the LDS read: ds_read_b64_tr_b16 for bf16
// (TR = true; 16-bit gathers otherwise), one ds_read_b32 per operand for the fp32 atom.  The
// reduction dimension is split over workgroups to fill the chip; partial results go to dense fp32
// slabs that a second kernel folds into the gradient (no atomics; a single split accumulates in
// place).  WI x WJ waves: (2,2) -> 128x128 outputs; (1,4) -> 64x256 for the 64-channel glyph convs.
// =================================================================================================

// Per-wave state of a gathered B operand of the TN kernel: NB pieces per reduction tile, the lane's (tap, channel) per piece is
// fixed for the whole kernel, only the reduction row p moves.
template <typename L, int NB> struct ConvTaps {
  __device__ __forceinline__ void set(const L&, int, int) {}
  __device__ __forceinline__ void init(const L&) {}
  __device__ __forceinline__ void fetch(const L&, int, int, bool, char*) {}
};
template <typename L, typename T, int NB> struct ConvTapsImpl {      // L = ConvLoader<T> or ConvLoaderDirect<T> (whose prepare32 loads nothing)
  typename ConvLoader<T>::Tap32 t[NB];
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ void set(const L& lb, int q, int col) {
#pragma unroll
    for (int i = 0; i < NB; ++i) if (i == q) t[i] = lb.tap32(col);
  }
  __device__ __forceinline__ void init(const L& lb) {
    rs = __builtin_amdgcn_make_buffer_rsrc((void*)lb.src, 0, (int)ConvLoader<T>::RECORDS, 0x00020000);
  }
  __device__ __forceinline__ void fetch(const L& lb, int q, int p, bool live, char* lds_wave_piece) {
    uint32_t voff = ConvLoader<T>::OOB;
#pragma unroll
    for (int i = 0; i < NB; ++i) if (i == q && live) voff = lb.voff32(lb.prepare32(p), t[i]);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_piece, 16, voff, 0, 0, 0);
  }
};
template <typename T, int NB> struct ConvTaps<ConvLoader<T>, NB> : ConvTapsImpl<ConvLoader<T>, T, NB> {};
template <typename T, int NB> struct ConvTaps<ConvLoaderDirect<T>, NB> : ConvTapsImpl<ConvLoaderDirect<T>, T, NB> {};

// One (split, tile) of a TN problem; `logical` = split * ntiles + tile.
// NST LDS stages of BP = TnGeo::BP / BPD reduction rows each (default: 2 stages of full tiles; the grouped kernel runs 4 stages of half
// tiles in the same 64 KB: three tiles in flight per workgroup instead of one).
template <typename T, typename BLoader, bool TR, int WI, int WJ, int NST = 2, int BPD = 1>
__device__ __forceinline__ void tn_tile_body(const T* __restrict__ A, int64_t lda, BLoader lb, int P, int I, int J, int tiles_j, int ntiles,
                                             int nsplit, int pchunk, int how, TnEpi ep, int logical, int jmajor = 0) {
  typedef typename MmaOf<T>::type Mma;
  typedef TnGeo<T> G;
  constexpr int BP = G::BP / BPD, KST = G::KSTEPS / BPD;
  constexpr int BI = 64 * WI, BJ = 64 * WJ;
  constexpr int RPA = BI * (int)sizeof(T), RPB = BJ * (int)sizeof(T);         // row pitches (bytes)
  constexpr int A_BYTES = BP * RPA, B_BYTES = BP * RPB, STAGE = A_BYTES + B_BYTES;
  constexpr int NA = A_BYTES / 4096, NB = B_BYTES / 4096;                     // 1-KiB instructions per wave per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave / WJ, wj = wave - wi * WJ;
  const int split = logical / ntiles, tile = logical - split * ntiles;
  // Tile order inside a problem.  The XCD remap hands every XCD a contiguous run of tiles; i-major (default) a run shares few dY panels
  // and streams the WHOLE B operand through its L2, j-major the other way round.  The grouped launch picks the order that streams the
  // smaller operand whole (round 6: FFN-down's weight gradient, I = 768, J = 3072, re-read its 33 MB `post` operand on three XCDs).
  int ti, tj;
  if (jmajor) { const int tiles_i = ntiles / tiles_j; tj = tile / tiles_i; ti = tile - tj * tiles_i; }
  else { ti = tile / tiles_j; tj = tile - ti * tiles_j; }
  const int i0 = ti * BI, j0 = tj * BJ;
  lb.clamp_rows();
  const int Pe = min(P, lb.rows);                      // device-side row bound (glyph dedup): re-split the live rows evenly
  if (Pe < P) pchunk = (((Pe + nsplit - 1) / nsplit + BP - 1) / BP) * BP;
  const int p_begin = split * pchunk;
  const int p_end = max(p_begin, min(Pe, p_begin + pchunk));
  const void* zero = (const void*)g_zero16;
  // live-tile list (unsplit dense reductions): the t-th tile of the loop is block tile_list[t] of the rows
  constexpr bool kDenseB0 = sizeof(typename BLoader::KPos) == sizeof(typename DenseLoader<T>::KPos);
  const bool listed = kDenseB0 && ep.tile_list != nullptr && nsplit == 1;
  // 16-row form of the list (bf16, 128 x 128 tiles: wave w fetches rows 16w .. 16w + 15 of both operand tiles): a reduction tile is
  // any FOUR live 16-row blocks, wave w takes block list[4 t + w] - 68 % instead of 84 % of the rows of a SIGHAN-shaped batch
  const bool sub16 = listed && ep.list_rows == 16 && sizeof(T) == 2 && WI == 2 && WJ == 2 && BPD == 1;

  // per-lane chunk coordinates are the same for every reduction tile: only the row base moves
  constexpr bool kDenseB = sizeof(typename BLoader::KPos) == sizeof(typename DenseLoader<T>::KPos);
  int apl[NA], bpl[NB];
  const T* acol[NA];
  typename BLoader::KPos bq[NB];
  bool bok[NB];
  ConvTaps<BLoader, NB> ctap;          // gathered B operand: the lane's tap per piece is fixed, the row moves (32-bit buffer addressing)
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int off = (wave * NA + q) * 1024 + lane * 16;
    apl[q] = off / RPA;
    const int col = i0 + ((off % RPA) ^ tn_swz<T, RPA>(apl[q])) / (int)sizeof(T);
    acol[q] = col < I ? A + col : nullptr;
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int off = (wave * NB + q) * 1024 + lane * 16;
    bpl[q] = off / RPB;
    const int col = j0 + ((off % RPB) ^ tn_swz<T, RPB>(bpl[q])) / (int)sizeof(T);
    bok[q] = col < J;
    bq[q] = lb.kpos(bok[q] ? col : 0);
    if constexpr (!kDenseB) ctap.set(lb, q, bok[q] ? col : 0);
  }
  if constexpr (!kDenseB) ctap.init(lb);

  floatx4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  // fused bias gradient: waves owning distinct i-ranges of the tj == 0 tiles also multiply their A fragments by ones
  const bool do_colsum = (ep.colsum != nullptr) && (tj == 0) && (wj == 0);
  floatx4 csum[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) csum[f] = floatx4{0.f, 0.f, 0.f, 0.f};
  typename Mma::Frag ones;
  if constexpr (sizeof(T) == 2) {
    typedef __attribute__((ext_vector_type(8))) short short8_t;
    const short8_t o = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    ones = __builtin_bit_cast(bf16x8_t, o);
  } else {
    ones = 1.0f;
  }

  // Fast addressing for full reduction tiles: one 64-bit source pointer per wave-instruction, advanced by a
  // constant (0 for lanes parked on the zero page); the checked path handles a ragged last tile.
  const char* pa[NA];
  const char* pb[NB];
  int64_t inca[NA], incb[NB];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    pa[q] = acol[q] != nullptr ? (const char*)(acol[q] + (int64_t)(p_begin + (sub16 ? apl[q] - 16 * wave : apl[q])) * lda) : (const char*)zero;
    inca[q] = acol[q] != nullptr ? (int64_t)BP * lda * (int64_t)sizeof(T) : 0;
  }
  if constexpr (kDenseB) {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const typename BLoader::Ctx c0 = lb.prepare(0);
      const typename BLoader::Ctx c1 = lb.prepare(1);
      const char* r0 = (const char*)lb.addr(c0, bq[q], zero);
      const char* r1 = (const char*)lb.addr(c1, bq[q], zero);
      const int64_t ldb_bytes = r1 - r0;                                       // row pitch of the dense B operand
      pb[q] = bok[q] ? r0 + (int64_t)(p_begin + (sub16 ? bpl[q] - 16 * wave : bpl[q])) * ldb_bytes : (const char*)zero;
      incb[q] = bok[q] ? (int64_t)BP * ldb_bytes : 0;
    }
  }
  int ptr_tile = p_begin / BP;           // the reduction tile pa / pb point at (live-tile list mode)
  int ptr_sub = 0;                       // sub16: the 16-row block this wave's pointers stand on
  bool sub_ok = true;                    // sub16: this wave has a block in the tile being issued (the last tile may hold fewer than four)
  auto issue = [&](int pt, int stage) {
    if (RL_PROBES && ep.probe == 2) return;
    char* base = smem + stage * STAGE;
    const bool full = sub16 || pt + BP <= p_end;
    if (sub16) {
      // pt = this wave's block index (wave-uniform) or -1.  Blocks are mostly 4 apart (consecutive tiles of live rows): the running
      // pointers advance by one tile after every issue and only the difference to that is applied here
      sub_ok = pt >= 0;
      const int d = sub_ok ? pt - ptr_sub : 0;
      if (d != 0) {
#pragma unroll
        for (int q = 0; q < NA; ++q) pa[q] += (inca[q] >> 2) * (int64_t)d;
        if constexpr (kDenseB) {
#pragma unroll
          for (int q = 0; q < NB; ++q) pb[q] += (incb[q] >> 2) * (int64_t)d;
        }
      }
      ptr_sub = (sub_ok ? pt : ptr_sub) + 4;
    } else
    // listed tiles are not consecutive: the pointers are rebuilt from the tile's first row (pa / pb hold the addresses of row 0 then)
    if (listed) {
      // listed tiles are mostly consecutive: the running pointers only jump (a wave-uniform number of tiles) over a run of dead blocks
      const int d = pt / BP - ptr_tile;
      if (d != 0) {
#pragma unroll
        for (int q = 0; q < NA; ++q) pa[q] += inca[q] * (int64_t)d;
        if constexpr (kDenseB) {
#pragma unroll
          for (int q = 0; q < NB; ++q) pb[q] += incb[q] * (int64_t)d;
        }
      }
      ptr_tile = pt / BP + 1;
    }
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const void* src = pa[q];
      if (!full && pt + apl[q] >= p_end) src = zero;
      if (sub16 && !sub_ok) src = zero;
      glds16(src, base + (wave * NA + q) * 1024);
      pa[q] += inca[q];
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const void* src = zero;
      if constexpr (kDenseB) {
        src = pb[q];
        if (!full && pt + bpl[q] >= p_end) src = zero;
        if (sub16 && !sub_ok) src = zero;
        pb[q] += incb[q];
      } else {
        const int p = pt + bpl[q];
        ctap.fetch(lb, q, p, p < p_end && bok[q], base + A_BYTES + (wave * NB + q) * 1024);
        continue;
      }
      glds16(src, base + A_BYTES + (wave * NB + q) * 1024);
    }
  };

  constexpr int NL = NA + NB;                          // fetch instructions per wave per tile
  const int n_list = listed ? *ep.n_tiles : 0;
  const int nt = sub16 ? min((n_list + 3) >> 2, (p_end - p_begin) / BP) : listed ? min(n_list, (p_end - p_begin) / BP) : (p_end - p_begin + BP - 1) / BP;
  int issued = 0;
  // The live-tile list is copied to LDS once (TN_LIST_LDS bytes behind the stages) and read from there, one broadcast ds_read per tile.
  // Round 4 kept 64 entries in a register and re-loaded it every 16 tiles INSIDE the loop; that load sat in a conditional block, and at the
  // merge hipcc's waitcnt pass could no longer tell whether the register (shared with a fetch pointer by the allocator) still had a load
  // pending: it put an s_waitcnt vmcnt(0) between the first and the second LDS-DMA fetch of EVERY tile (found in the ISA in round 5) -
  // every K-tile of every weight-gradient launch paid a full memory latency with one fetch in flight.  LDS reads are counted by lgkmcnt:
  // nothing the fetch queue's counter has to be drained for.
  const int n_ent = sub16 ? min(n_list, 4 * nt) : nt;
  int* lds_list = (int*)(smem + NST * STAGE);
  if (listed) {
    for (int e = tid; e < n_ent; e += 256) lds_list[e] = ep.tile_list[e];
    __syncthreads();
  }
  auto issue_next = [&]() {
    if (issued < nt) {
      int pt = p_begin + issued * BP;
      if (sub16) {
        const int e = 4 * issued + wave;
        pt = e < n_ent ? __builtin_amdgcn_readfirstlane(lds_list[e]) : -1;
      } else if (listed) {
        pt = __builtin_amdgcn_readfirstlane(lds_list[issued]) * BP;
      }
      issue(pt, issued % NST);
      ++issued;
    }
  };
#pragma unroll
  for (int q = 0; q < NST - 1; ++q) issue_next();
  for (int t = 0; t < nt; ++t) {
    // tile t has landed once at most the (issued - 1 - t) younger tiles' fetches are outstanding (counted vmcnt; raw barrier:
    // __syncthreads() would drain vmcnt to 0)
    const int younger = issued - 1 - t;
    if (NST == 2 || younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (NST == 3 || younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_next();                                      // into the stage every wave finished reading before this barrier
    const int cur = t % NST;
    const char* At = smem + cur * STAGE;
    const char* Bt = At + A_BYTES;
    if (RL_PROBES && ep.probe == 3) continue;
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
      typename Mma::Frag a[4], b[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int ca = wi * 64 + f * 16, cb = wj * 64 + f * 16;
        if constexpr (sizeof(T) == 2) {
          a[f] = tn_frag_bf16<TR, RPA>(At, ks, ca, l15, g);
          b[f] = tn_frag_bf16<TR, RPB>(Bt, ks, cb, l15, g);
        } else {
          const int p = ks * 4 + g;
          a[f] = *(const float*)(At + p * RPA + (((ca + l15) * 4) ^ tn_swz<float, RPA>(p)));
          b[f] = *(const float*)(Bt + p * RPB + (((cb + l15) * 4) ^ tn_swz<float, RPB>(p)));
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mma::mma(b[j], a[i], acc[i][j]);
      if (do_colsum) {
#pragma unroll
        for (int f = 0; f < 4; ++f) csum[f] = Mma::mma(ones, a[f], csum[f]);   // every row of the result = sum_p A[p, i]
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      tn_epilogue4(ep, how, split, I, J, i0 + wi * 64 + i * 16 + l15, j0 + wj * 64 + j * 16 + 4 * g, acc[i][j]);
  if (do_colsum && g == 0) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int i = i0 + wi * 64 + f * 16 + l15;
      if (i < I) atomicAdd(ep.colsum + i, csum[f][0] * tn_alpha(ep));
    }
  }
}

template <typename T, typename BLoader, bool TR, int WI, int WJ>
__global__ void __launch_bounds__(256, 2)
gemm_tn_kernel(const T* __restrict__ A, int64_t lda, BLoader lb, int P, int I, int J, int tiles_j, int ntiles,
               int nsplit, int pchunk, int how, TnEpi ep) {
  // 1-D grid of (split, tile) pairs: the XCD remap hands each XCD a contiguous run of them, so the tiles of one
  // reduction split (which share A / B row panels) sit behind one L2.
  tn_tile_body<T, BLoader, TR, WI, WJ>(A, lda, lb, P, I, J, tiles_j, ntiles, nsplit, pchunk, how, ep, xcd_remap(blockIdx.x, ntiles * nsplit));
}

// Grouped form: up to TN_GROUP_MAX dense problems that share the reduction length P in ONE launch, each workgroup owning one
// 128x128 tile of one problem over the whole reduction (no split: nothing to fold).  The four weight gradients of a transformer
// layer are 432 such tiles - one round of the chip's 512 workgroup slots - where launched one by one each needs a 3-4 way
// reduction split (slab write + fold pass) to fill the chip: 8 launches and ~75 MB of slab traffic per layer become 1 launch.
template <typename T, bool TR, int NST, int BPD>
__global__ void __launch_bounds__(256, 2)
gemm_tn_group_kernel(TnGroup<T> grp, int P, int pchunk) {
  const int logical = xcd_remap(blockIdx.x, grp.total_tiles);
  int k = 0;
#pragma unroll
  for (int i = 1; i < TN_GROUP_MAX; ++i) if (i < grp.n && logical >= grp.p[i].tile_begin) k = i;
  const TnGroupProblem<T>& pr = grp.p[k];
  DenseLoader<T> lb{pr.B, pr.ldb, P, pr.J};
  TnEpi ep;
  ep.out = pr.out; ep.ldo = pr.ldo; ep.colsum = pr.colsum; ep.alpha = grp.alpha; ep.probe = grp.probe; ep.overwrite = grp.overwrite;
  ep.tile_list = grp.tile_list; ep.n_tiles = grp.n_tiles; ep.list_rows = grp.list_rows;
  tn_tile_body<T, DenseLoader<T>, TR, 2, 2, NST, BPD>(pr.A, pr.lda, lb, P, pr.I, pr.J, pr.tiles_j, pr.ntiles, 1, pchunk, TN_OUT_DIRECT, ep,
                                                      logical - pr.tile_begin, pr.jmajor);
}

// out(mapped) += alpha * sum_s slab[s][i][j], in a fixed order (bitwise reproducible).
// A block is EB items x SB split lanes (EB * SB = 256): lane group s sums the splits s, s + SB, ... of its item, the partial sums meet
// in LDS and lane group 0 adds them in order.  SB > 1 is what keeps the pass short when a small gradient was split hundreds of ways
// to fill the chip (the 64-channel glyph convolutions: 4608 outputs x 512 splits took 120 us with one thread per output walking all
// the slabs; tools/conv_tn_probe.py).
// PLAIN: an item is one float4 of a row.  CONVW: an item is one (output channel, input channel) pair, all KH*KW taps - the slab is
// read along ci (coalesced) per tap and the [Co][Ci][KH][KW] gradient is written as KH*KW consecutive floats per thread.
template <int SB>
__global__ void __launch_bounds__(256) tn_fold_plain_kernel(TnEpi ep, int nsplit, int I, int J) {
  constexpr int EB = 256 / SB;
  __shared__ floatx4 part[SB > 1 ? SB : 1][EB];
  const int64_t n4 = (int64_t)I * J / 4, plane = (int64_t)I * J;
  const int le = threadIdx.x % EB, ls = threadIdx.x / EB;
  for (int64_t e0 = (int64_t)blockIdx.x * EB; e0 < n4; e0 += (int64_t)gridDim.x * EB) {
    const int64_t e = e0 + le;
    floatx4 s = floatx4{0.f, 0.f, 0.f, 0.f};
    if (e < n4)
      for (int k = ls; k < nsplit; k += SB) s += *(const floatx4*)(ep.slab + k * plane + e * 4);
    if constexpr (SB > 1) {
      part[ls][le] = s;
      __syncthreads();
      if (ls == 0) {
#pragma unroll
        for (int k = 1; k < SB; ++k) s += part[k][le];
      }
    }
    if (ls == 0 && e < n4) {
      s *= tn_alpha(ep);
      const int i = (int)((e * 4) / J), j = (int)((e * 4) - (int64_t)i * J);
      if (ep.mode == TN_PLAIN && (ep.ldo & 3) == 0) {
        floatx4* o = (floatx4*)(ep.out + (int64_t)i * ep.ldo + j);
        *o = *o + s;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t idx = tn_out_index(ep, i, j + r);
          if (idx >= 0) ep.out[idx] += s[r];
        }
      }
    }
    if constexpr (SB > 1) __syncthreads();
  }
}

template <int SB>
__global__ void __launch_bounds__(256) tn_fold_convw_kernel(TnEpi ep, int nsplit, int I, int J) {
  constexpr int EB = 256 / SB;
  __shared__ float part[SB > 1 ? SB : 1][EB];
  const int64_t items = (int64_t)I * ep.Cin, plane = (int64_t)I * J;
  const int le = threadIdx.x % EB, ls = threadIdx.x / EB;
  for (int64_t e0 = (int64_t)blockIdx.x * EB; e0 < items; e0 += (int64_t)gridDim.x * EB) {
    const int64_t e = e0 + le;
    const int i = (int)(e / ep.Cin), ci = (int)(e - (int64_t)i * ep.Cin);
    const int ntaps = J / ep.Cpad;          // the taps the J columns cover (all KH*KW, or the one live tap of a 1x1 map)
    for (int tap = 0; tap < ntaps; ++tap) {
      float s = 0.f;
      if (e < items) {
        const float* src = ep.slab + (int64_t)i * J + tap * ep.Cpad + ci;
        for (int k = ls; k < nsplit; k += SB) s += src[k * plane];
      }
      if constexpr (SB > 1) {
        part[ls][le] = s;
        __syncthreads();
        if (ls == 0) {
#pragma unroll
          for (int k = 1; k < SB; ++k) s += part[k][le];
        }
        __syncthreads();
      }
      if (ls == 0 && e < items) ep.out[e * ep.KHW + tap + ep.tap0] += s * tn_alpha(ep);
    }
  }
}

void tn_fold_launch(hipStream_t st, const TnEpi& ep, int nsplit, int I, int J) {
  // the per-(co, ci) form pays off once there are enough pairs to fill blocks; tiny gradients (64 x 3 x 9) go through the float4 form
  const bool convw = ep.mode != TN_PLAIN && (int64_t)I * ep.Cin >= 16384;
  const int64_t items = convw ? (int64_t)I * ep.Cin : (int64_t)I * J / 4;
  // split lanes: as many as keep >= ~1024 blocks' worth of items busy, at most 16 and at most the split count
  int sb = 1;
  while (sb < 16 && sb * 2 <= nsplit && items * sb < 256 * 1024) sb *= 2;
  int blocks = (int)((items + 256 / sb - 1) / (256 / sb));
  if (blocks > 4096) blocks = 4096;
#define RL_FOLD(SBV) \
  if (convw) hipLaunchKernelGGL(tn_fold_convw_kernel<SBV>, dim3(blocks), dim3(256), 0, st, ep, nsplit, I, J); \
  else hipLaunchKernelGGL(tn_fold_plain_kernel<SBV>, dim3(blocks), dim3(256), 0, st, ep, nsplit, I, J);
  switch (sb) {
    case 1: RL_FOLD(1) break;
    case 2: RL_FOLD(2) break;
    case 4: RL_FOLD(4) break;
    case 8: RL_FOLD(8) break;
    default: RL_FOLD(16) break;
  }
#undef RL_FOLD
}

static int g_tn_tr = 1;   // ds_read_b64_tr_b16 verified on MI355X (tests/test_kernels_gpu.py::test_gemm_tn)
void set_tn_transpose_read(int use_tr) { g_tn_tr = use_tr; }

template <typename T, typename BLoader, int WI, int WJ>
static int launch_tn_tile(hipStream_t st, const T* A, int64_t lda, const BLoader& lb, int P, int I, int J, const TnEpi& ep_) {
  typedef TnGeo<T> G;
  TnEpi ep = ep_;
  ep.probe = g_tn_probe;
  constexpr int BI = 64 * WI, BJ = 64 * WJ;
  const int tiles_i = (I + BI - 1) / BI, tiles_j = (J + BJ - 1) / BJ, ntiles = tiles_i * tiles_j;
  // Reduction split: the chip holds 512 workgroups (256 CUs x 2).  Pick the split count that minimises
  //   rounds(split) x rows per workgroup + split x (slab write + fold read, in row-times)
  // - a split that spills a few workgroups into a second round costs a whole extra pass (e.g. 144 tiles: 3 splits =
  // 432 workgroups beat 4 splits = 576 by 25 %, tools/nt_probe.cpp sweep).
  const int max_split = (P + 4 * G::BP - 1) / (4 * G::BP);
  int64_t cap = max_split;
  if (ep.slab != nullptr && ep.slab_elems / ((int64_t)I * J) < cap) cap = ep.slab_elems / ((int64_t)I * J);
  if (cap < 1) cap = 1;
  int nsplit = 1, pchunk = ((P + G::BP - 1) / G::BP) * G::BP;
  {
    double best = 1e30;
    const double slab_rows = 1e-4 * (double)I * (double)J;
    for (int ns = 1; ns <= (int)cap; ++ns) {
      if (g_tn_split > 0 && ns != (g_tn_split < (int)cap ? g_tn_split : (int)cap)) continue;
      const int pc = (((P + ns - 1) / ns + G::BP - 1) / G::BP) * G::BP;
      const int ne = (P + pc - 1) / pc;
      const double rounds = (double)(((int64_t)ntiles * ne + 511) / 512);
      const double c = rounds * pc + (ne > 1 ? ne * slab_rows : 0.0);
      if (c < best) { best = c; nsplit = ne; pchunk = pc; }
    }
  }
  const int how = nsplit == 1 ? TN_OUT_DIRECT : (ep.slab != nullptr ? TN_OUT_SLAB : TN_OUT_ATOMIC);
  const int list_lds = ep.tile_list != nullptr ? tn_list_lds_bytes((int64_t)P / (ep.list_rows ? ep.list_rows : G::BP)) : 0;
  if (list_lds < 0) return RL_ERR_ARG;
  const size_t lds = 2 * (size_t)G::BP * (BI + BJ) * sizeof(T) + (size_t)list_lds;
  const int lds_attr = (int)(2 * (size_t)G::BP * (BI + BJ) * sizeof(T)) + TN_LIST_LDS_MAX;
  dim3 grid(ntiles * nsplit);
  {
    ProfScope ps(st, sizeof(typename BLoader::KPos) == sizeof(typename DenseLoader<T>::KPos) ? PK_GEMM_TN : PK_CONV_TN, 2.0 * P * I * J);
    if constexpr (sizeof(typename BLoader::KPos) == sizeof(typename DenseLoader<T>::KPos))
      if (lb.rows_dev != nullptr) prof_set_exec(lb.rows_dev, 2.0 * I * J, G::BP, P);       // the reduction stops at the device-side row bound
    if (sizeof(T) == 2 && g_tn_tr) {
      static bool a1 = false;
      if (!a1) { (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<T, BLoader, true, WI, WJ>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr); a1 = true; }
      RL_LAUNCH((gemm_tn_kernel<T, BLoader, true, WI, WJ>), grid, dim3(256), lds, st, A, lda, lb, P, I, J, tiles_j, ntiles, nsplit, pchunk, how, ep);
    } else {
      static bool a2 = false;
      if (!a2) { (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<T, BLoader, false, WI, WJ>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr); a2 = true; }
      RL_LAUNCH((gemm_tn_kernel<T, BLoader, false, WI, WJ>), grid, dim3(256), lds, st, A, lda, lb, P, I, J, tiles_j, ntiles, nsplit, pchunk, how, ep);
    }
    if (how == TN_OUT_SLAB && ep.probe != 4) tn_fold_launch(st, ep, nsplit, I, J);
  }
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}

template <typename T, typename BLoader>
static int launch_tn(hipStream_t st, const T* A, int64_t lda, const BLoader& lb, int P, int I, int J, const TnEpi& ep) {
  if (P <= 0 || I <= 0 || J <= 0) return RL_OK;
  typedef TnGeo<T> G;
  if ((lda % G::VEC) || (I % G::VEC) || (J % G::VEC)) return RL_ERR_ARG;
  if (I <= 64) return launch_tn_tile<T, BLoader, 1, 4>(st, A, lda, lb, P, I, J, ep);
  return launch_tn_tile<T, BLoader, 2, 2>(st, A, lda, lb, P, I, J, ep);
}

static int g_tn_variant = 0;      // 0 production (the 4-wave kernel), 8 the experimental 8-wave ping-pong kernel (gemm_tn8.hip: correct, 7-16 % slower)
void set_tn_variant(int v) { g_tn_variant = RL_PROBES ? v : 0; }

static int g_tn_group8 = 0;          // grouped weight gradients on the 8-wave 256 x 128 kernel (gemm_tn8_group): measured, not faster (see gemm_tn8.hip)
void set_tn_group8(int on) { g_tn_group8 = on; }
static int g_tn_group_ring = 0;      // measured: 4 x 32-row stages 3.76 ms/step vs 3.39 for 2 x 64-row stages (more barriers, smaller DMA batches)
void set_tn_group_ring(int on) { g_tn_group_ring = on; }

template <typename T>
int gemm_tn_group(hipStream_t st, int n, const TnGroupProblem<T>* probs, int P, float alpha, int overwrite, const int* tile_list,
                  const int* n_tiles, int list_rows) {
  typedef TnGeo<T> G;
  if (n < 1 || n > TN_GROUP_MAX || P <= 0) return RL_ERR_ARG;
  if constexpr (sizeof(T) == 2) {
    if (g_tn_group8 && g_tn_tr && !g_tn_group_ring && g_tn_probe == 0 && (tile_list == nullptr || list_rows == 16)) {
      const int rc = gemm_tn8_group(st, n, probs, P, alpha, overwrite, tile_list, n_tiles, list_rows);
      if (rc != RL_ERR_ARG) return rc;
    }
  }
  TnGroup<T> grp;
  grp.n = n; grp.alpha = alpha; grp.probe = g_tn_probe; grp.overwrite = overwrite;
  if (list_rows != 0 && list_rows != G::BP && !(list_rows == 16 && sizeof(T) == 2)) return RL_ERR_ARG;
  // A caller that hands over a live-block list may rely on it (a live-row step leaves stale values in the rows of unlisted blocks):
  // the list is honoured or the call fails - it is never silently dropped.  The four-stage ring has no list form, so a listed call
  // runs on the default two-stage ring whatever the knob says.
  const bool listed_call = tile_list != nullptr;
  if (listed_call && (n_tiles == nullptr || (P % G::BP) != 0)) return RL_ERR_ARG;
  const bool ring4 = g_tn_group_ring && !listed_call;
  if (listed_call) { grp.tile_list = tile_list; grp.n_tiles = n_tiles; grp.list_rows = list_rows ? list_rows : G::BP; }
  int total = 0;
  double flops = 0.0;
  for (int k = 0; k < n; ++k) {
    TnGroupProblem<T> pr = probs[k];
    if (pr.I <= 0 || pr.J <= 0 || (pr.lda % G::VEC) || (pr.ldb % G::VEC) || (pr.I % G::VEC) || (pr.J % G::VEC) || (pr.ldo & 3)) return RL_ERR_ARG;
    pr.tiles_j = (pr.J + 127) / 128;
    pr.ntiles = ((pr.I + 127) / 128) * pr.tiles_j;
    pr.jmajor = (g_tn_jmajor && pr.J > pr.I) ? 1 : 0;
    pr.tile_begin = total;
    total += pr.ntiles;
    flops += 2.0 * P * pr.I * pr.J;
    grp.p[k] = pr;
  }
  grp.total_tiles = total;
  const int pchunk = ((P + G::BP - 1) / G::BP) * G::BP;
  // (+ the live-block list: sized from the entry count, at most TN_LIST_MAX_ENTRIES)
  const int list_lds = grp.tile_list != nullptr ? tn_list_lds_bytes((int64_t)P / grp.list_rows) : 0;
  if (list_lds < 0) return RL_ERR_ARG;
  const size_t lds = 2 * (size_t)G::BP * 256 * sizeof(T) + (size_t)list_lds;
  const int lds_attr = (int)(2 * (size_t)G::BP * 256 * sizeof(T)) + TN_LIST_LDS_MAX;
  ProfScope ps(st, PK_GEMM_TN, flops);
  if (grp.tile_list != nullptr) prof_set_exec(grp.n_tiles, flops / P * grp.list_rows, sizeof(T) == 2 && grp.list_rows == 16 ? 4 : 1, P / grp.list_rows);   // live blocks only
  // ring: the 2 full stages of the single-problem kernel (default) or 4 stages of half tiles (three tiles in flight per workgroup,
  // g_tn_group_ring = 1: built to test the fetch-latency hypothesis - 11 % slower)
#define RL_TN_GROUP(TRV, NSTV, BPDV) do { \
    static bool attr = false; \
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_tn_group_kernel<T, TRV, NSTV, BPDV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr); attr = true; } \
    RL_LAUNCH((gemm_tn_group_kernel<T, TRV, NSTV, BPDV>), dim3(total), dim3(256), lds, st, grp, P, pchunk); } while (0)
  const bool tr = sizeof(T) == 2 && g_tn_tr;
  if (ring4) { if (tr) RL_TN_GROUP(true, 4, 2); else RL_TN_GROUP(false, 4, 2); }
  else { if (tr) RL_TN_GROUP(true, 2, 1); else RL_TN_GROUP(false, 2, 1); }
#undef RL_TN_GROUP
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}
template int gemm_tn_group<bf16_t>(hipStream_t, int, const TnGroupProblem<bf16_t>*, int, float, int, const int*, const int*, int);
template int gemm_tn_group<float>(hipStream_t, int, const TnGroupProblem<float>*, int, float, int, const int*, const int*, int);

template <typename T>
int gemm_tn(hipStream_t st, const T* A, int64_t lda, const T* B, int64_t ldb, int P, int I, int J, const TnEpi& ep,
            const int* rows_dev) {
  if (ldb % TnGeo<T>::VEC) return RL_ERR_ARG;
  if constexpr (sizeof(T) == 2) {
#if RL_PROBES
    if (rows_dev == nullptr && g_tn_variant == 8 && g_tn_probe == 0 && tn8_supported(lda, ldb, P, I, J, ep))
      return gemm_tn8(st, A, lda, B, ldb, P, I, J, ep, g_tn_split);
#endif
  }
  DenseLoader<T> lb{B, ldb, P, J};
  lb.rows_dev = rows_dev;
  return launch_tn<T, DenseLoader<T>>(st, A, lda, lb, P, I, J, ep);
}
template <typename T>
int gemm_tn_conv(hipStream_t st, const T* A, int64_t lda, const ConvLoader<T>& lb_, int P, int I, int J, const TnEpi& ep) {
  if (lb_.C % TnGeo<T>::VEC) return RL_ERR_ARG;
  ConvLoader<T> lb = lb_;
  lb.finalize();
  if (lb.K != J || !lb.span_ok()) return RL_ERR_ARG;
  if constexpr (sizeof(T) == 2) {
    // 64 -> 64 channels, 3x3 / stride 1 / pad 1 on 16x16 maps (glyph ResNet block 1): the LDS-resident-input kernel
    if (g_conv_c64 && g_tn_probe == 0 && lb.C == 64 && lb.KH == 3 && lb.KW == 3 && lb.stride == 1 && lb.pad == 1 && lb.mode == 0 && lb.Hr == 16 &&
        lb.Wr == 16 && lb.Hs == 16 && lb.Ws == 16 && lb.img_index == nullptr && I == 64 && lda == 64 && P == lb.rows && (P % 256) == 0 &&
        ep.mode == TN_CONVW && ep.Cin == 64 && ep.Cpad == 64 && ep.KHW == 9 && ep.tap0 == 0 && ep.alpha == 1.0f && ep.alpha_dev == nullptr && ep.slab != nullptr &&
        ep.slab_elems >= 64 * 576 && (int64_t)P * 128 < 0xFFFFFE00ll)
      return conv_wgrad_c64(st, A, lb.src, P, lb.rows_dev, ep);
  }
  if (lb.img_index == nullptr) return launch_tn<T, ConvLoaderDirect<T>>(st, A, lda, ConvLoaderDirect<T>(lb), P, I, J, ep);      // (every engine call)
  return launch_tn<T, ConvLoader<T>>(st, A, lda, lb, P, I, J, ep);
}
template int gemm_tn<bf16_t>(hipStream_t, const bf16_t*, int64_t, const bf16_t*, int64_t, int, int, int, const TnEpi&, const int*);
template int gemm_tn<float>(hipStream_t, const float*, int64_t, const float*, int64_t, int, int, int, const TnEpi&, const int*);
template int gemm_tn_conv<bf16_t>(hipStream_t, const bf16_t*, int64_t, const ConvLoader<bf16_t>&, int, int, int, const TnEpi&);
template int gemm_tn_conv<float>(hipStream_t, const float*, int64_t, const ConvLoader<float>&, int, int, int, const TnEpi&);

}  // namespace rl
