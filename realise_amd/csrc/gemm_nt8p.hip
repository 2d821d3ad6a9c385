// Persistent ping-pong NT GEMM for gfx950 (bf16, K % 64 == 0):  C[M,N] = A[M,K] . B[N,K]^T + fused epilogue, one 512-thread
// workgroup per CU that walks over several 256 x 192 output tiles.
//
// Why: on the K = 768 shapes of the transformer stacks the non-persistent kernels spend a third
// of their time OUTSIDE the main loop - launch ramp, the first fetch's latency, and an output write nobody overlaps (a K = 64
// launch of the 8192 x 3072 shape takes 16-20 us against 44 for K = 768; inside a training step, where operands and outputs miss
// the Infinity Cache, 62).  The two-workgroups-per-CU 128 x 192 shape hides part of it by co-residency at the price of 43 % more
// fetched bytes and 60 % more fragment reads per flop.  This kernel keeps the big tile (64 x 96 per wave: 10 fragment reads per 24
// MFMAs) and gets the overlap from its own instruction stream:
//   * the epilogue touches NO LDS: the MFMA operands are swapped (acc = B-fragment x A-fragment) so that a lane holds four
//     consecutive output COLUMNS of one row; after the fused epilogue math two packed accumulator tiles exchange 16-lane rows
//     (v_permlane16_swap) and every lane owns 8 consecutive bf16 = one 16-byte store, 64 contiguous bytes per row per instruction;
//   * so the NEXT tile's first K-tiles are fetched into the (free) LDS ring before the epilogue math starts, and the stores drain
//     under the next tile's main loop.  Loads and stores retire in order on the vector-memory counter (gfx9), so the first LEAD - 1
//     phases of the next tile wait with vmcnt(N + S), S = the stores the epilogue issued behind the prologue fetches.
// Main loop, fetch schedule and hazard rules are those of gemm_nt8.hip (nt8_cfg.h); the outputs are bit-identical to it (same
// k-order inside every MFMA, same fp32 epilogue arithmetic).
#include "gemm_dev.h"
#include "nt8_cfg.h"
#include "prof.h"

namespace rl {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr int EPI_STORE_F32X = 8;      // template value only (not an EpiMode): EPI_STORE + EpiParams::out_f32
constexpr int EPI_GELU_NOPRE = 9;      // template value only: EPI_GELU without the pre-activation store (out2 == nullptr: no backward follows)
typedef __attribute__((ext_vector_type(4))) int i32x4;

// (the host pass of hipcc parses kernel bodies too and silently drops a kernel stub whose body holds AMDGPU register constraints
// it cannot place: the asm statements exist in the device pass only)
__device__ __forceinline__ u32x4 asm_buffer_load_b128(uint32_t voff, i32x4 rsrc) {
  u32x4 v = {0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
  // (s_nop 4: 5 wait states behind a VALU write - a spill restore - of the descriptor, which hipcc does not pad for asm text: tools/isa_hazard_scan.py)
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(v) : "v"(voff), "s"(rsrc) : "memory");
#endif
  return v;
}
__device__ __forceinline__ void asm_after_wait(u32x4& v) {      // pins the uses of an asm-loaded value behind the preceding s_waitcnt
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
}

// The body lives in a __device__ function: the host pass of hipcc 7.2 parses __global__ bodies and silently drops the stub of a
// kernel template whose body it cannot digest (device-only builtins inside lambdas); a __device__ function is never looked into.
template <typename C, int EPI>
__device__ __forceinline__ void nt8p_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int M_, int N,
                                          int K, int tiles_n, int ntiles_, const EpiParams<bf16_t>& ep, int rows_per_xcd_) {
  // Round 6: an optional device-side row count (EpiParams::m_dev; the training classifier over the rows that enter the loss - their
  // number lives on the device).  The launch is sized for the nominal M; every workgroup shrinks the problem to the live tile rows
  // before it derives its tile walk: rows at or beyond the count are neither read (they lie outside the A descriptor: zeros) nor
  // stored (outside the output descriptor: dropped), tiles beyond it do not exist.
  int M = M_, ntiles = ntiles_, rows_per_xcd = rows_per_xcd_;
  if (ep.m_dev != nullptr) {
    const int md = *ep.m_dev;
    if (md <= 0) return;
    if (md < M) {
      M = md;
      const int tml = (M + C::BM - 1) / C::BM;
      ntiles = tml * tiles_n;
      rows_per_xcd = (rows_per_xcd_ > 0 && (tml % 8) == 0) ? tml / 8 : 0;
    }
  }
  typedef MmaBF16 Mma;
  constexpr int NPW = C::NPW, NPH = C::NPH, NS = C::NS, LEAD = C::LEAD, SQ = C::SQ, HT = C::HT, MT = C::MT, NT = C::NT;
  static_assert(!C::HOLD_B && C::ISSUE_AT == 0 && C::FW == 8 && (NT % 2) == 0, "persistent kernel: hold-A schedule, fetches at the end of the memory segment");
  // EPI_STORE_F32X (this file only): EPI_STORE + the fp32 copy of every stored value (EpiParams::out_f32), one 16-byte store per
  // accumulator tile in the accumulator layout (a lane: 4 consecutive columns of one row)
  constexpr int NSTORE = MT * (NT / 2) * (EPI == EPI_GELU ? 2 : 1) + (EPI == EPI_STORE_F32X ? MT * NT : 0);      // 16-byte store instructions of one tile's epilogue, per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave / C::WN, wn = wave - wm * C::WN;
  const int nk = K >> 6;
  const int nwg = gridDim.x;
  // Tile walk.  rows_per_xcd > 0 (tile rows divisible by 8, grid a multiple of 8): XCD x (= blockIdx % 8 as dispatched today; a
  // different placement changes speed only) OWNS tile rows [x rpx, (x + 1) rpx) for the whole launch and walks over the tile
  // columns, rpx rows at a time: its A panels (rpx x BM rows) stay in its 4 MiB L2 for the whole kernel, every B panel is shared by
  // the rpx workgroups that run it together, and HBM sees every A byte once and every B byte once per XCD.  Otherwise: logical
  // tile ids in XCD-contiguous chunks, stride = grid.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = nwg >> 3;
  const int per_xcd = rows_per_xcd * tiles_n;
  auto tile_of = [&](int it, int& tm, int& tn) -> bool {       // it = this workgroup's it-th tile
    if (rows_per_xcd > 0) {
      const int idx = slot + it * nslots;
      if (idx >= per_xcd) return false;
      const int c = idx / rows_per_xcd;
      tm = xcd * rows_per_xcd + (idx - c * rows_per_xcd);
      tn = c + (xcd * tiles_n >> 3);                // every XCD starts its column walk an eighth of the way further: the eight L2s
      if (tn >= tiles_n) tn -= tiles_n;             //   do not ask the memory side for the same B panel at the same time
      return true;
    }
    const int t = xcd_remap(blockIdx.x, nwg) + it * nwg;
    if (t >= ntiles) return false;
    tm = t / tiles_n; tn = t - tm * tiles_n;
    return true;
  };

  const int lrow = lane >> 3;
  const int kchunk_b = (((lane & 7) ^ lrow) << 4);
  // a piece = 8 rows x 128 B: LDS offset inside a stage and source row offset are wave-uniform (SGPRs); the lane part of the source
  // address (row lrow of the piece, swizzled chunk) is ONE VGPR per operand, re-based per tile.  The piece's row offset is added
  // to it per fetch (one v_add), so rows beyond M / N fall outside num_records and come back as zeros - no clamping.
  int lo[NPW];
  uint32_t po[NPW];
#pragma unroll
  for (int s = 0; s < NPW; ++s) {
    const int p = s * 8 + wave;
    int row, is_b;
    if (s < C::HPW) { row = p * 8; is_b = 0; }
    else {
      const int pp = p - C::HP, q = pp / C::GP, rem = pp - q * C::GP, slice = rem / (SQ * 2), j = rem - slice * (SQ * 2);
      row = slice * C::SR + q * SQ * 16 + j * 8; is_b = 1;
    }
    lo[s] = (is_b ? C::A_BYTES : 0) + row * 128;
    po[s] = (uint32_t)((int64_t)row * (is_b ? ldb : lda) * 2);
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(((int64_t)(M - 1) * lda + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(((int64_t)(N - 1) * ldb + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)ep.out, 0, (int)(((int64_t)(M - 1) * ep.ldo + N) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == EPI_GELU ? ep.out2 : ep.out), 0, (int)(((int64_t)(M - 1) * ep.ldo + N) * 2), 0x00020000);
  // Epilogue operands (bias, saved pre-activations) are fetched by inline-asm buffer loads: next to LDS-DMA fetches in flight hipcc
  // waits vmcnt(0) before the first use of any load it can see, which would drain the next tile's prologue ahead of the epilogue
  // math; these loads are invisible to it and waited for by count (they are OLDER than the prologue fetches issued behind them).
  auto rsrc_words = [](const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    i32x4 r = {(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
    return r;
  };
  const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == EPI_STORE_F32X ? (void*)ep.out_f32 : (void*)ep.out), 0,
                                                                       EPI == EPI_STORE_F32X ? (int)(uint32_t)(((int64_t)(M - 1) * ep.ldo_f32 + N) * 4) : 0, 0x00020000);
  const bool has_bias = EPI != EPI_GELU_BWD && ep.bias != nullptr;
  const i32x4 wBias = rsrc_words(has_bias ? (const void*)ep.bias : (const void*)A, has_bias ? (uint32_t)N * 4u : 0u);
  const i32x4 wX = rsrc_words(EPI == EPI_GELU_BWD ? (const void*)ep.aux : (const void*)A,
                              EPI == EPI_GELU_BWD ? (uint32_t)(((int64_t)(M - 1) * ep.ldaux + N) * 2) : 0u);

  uint32_t goA = 0, goB = 0;
  auto set_tile = [&](int tm, int tn, int& m0, int& n0) {
    m0 = tm * C::BM; n0 = tn * C::BN;
    goA = (uint32_t)((int64_t)(m0 + lrow) * lda * 2 + kchunk_b);
    goB = (uint32_t)((int64_t)(n0 + lrow) * ldb * 2 + kchunk_b);
  };
  auto issue = [&](auto s_c, int stage, int ktile) {
    constexpr int s = decltype(s_c)::value;
    constexpr bool is_b = s >= C::HPW;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(is_b ? rsB : rsA, (__attribute__((address_space(3))) void*)(smem + stage * C::STAGE + lo[s]), 16,
                                             (is_b ? goB : goA) + po[s], ktile * 128, 0, 0);
  };
  auto prologue = [&]() {
    static_for<C::PRO_TILES>([&](auto dt_c) {
      constexpr int dt = decltype(dt_c)::value;
      static_for<NPW>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (C::in_prologue(dt, s)) issue(s_c, dt % NS, dt);
      });
    });
  };

  int fa[2], fb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int lane_sw = l15 * 128 + ((((ks << 2) + g) ^ (l15 & 7)) << 4);
    fa[ks] = wm * C::RM * 128 + lane_sw;
    fb[ks] = C::A_BYTES + wn * C::RN * 128 + lane_sw;
  }
  floatx4 acc[MT][NT];
  bf16x8_t hf[HT][2], sf[SQ][2];

  // one phase of the K loop (gemm_nt8.hip).  behind_stores: the previous tile's NSTORE epilogue stores sit between this tile's
  // prologue fetches and its in-loop fetches on the (in-order) vector-memory counter: while a phase's wait only covers prologue
  // fetches (Nt8Cfg::wait_is_prologue_only) the stores may stay in flight - the count grows by NSTORE; as soon as an awaited fetch
  // was issued inside the loop it is younger than the stores, which have then retired with it.
  auto phase = [&](auto par_c, auto q_c, int t, bool behind_stores) {
    constexpr int PAR = decltype(par_c)::value, q = decltype(q_c)::value;
    constexpr int dt2 = (q + LEAD) / NPH, q2 = (q + LEAD) % NPH, SBASE = PAR * C::STAGE;
    constexpr int dtw = (q + LEAD) / NPH;
    if constexpr (q == 0) {
#pragma unroll
      for (int h = 0; h < HT; ++h)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) hf[h][ks] = *(const bf16x8_t*)(smem + SBASE + h * 2048 + fa[ks]);
    }
#pragma unroll
    for (int i = 0; i < SQ; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) sf[i][ks] = *(const bf16x8_t*)(smem + SBASE + (q * SQ + i) * 2048 + fb[ks]);
    __builtin_amdgcn_sched_barrier(0);
    if (t + dt2 < nk) {
      static_for<NPW>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (s >= C::cum(q2) && s < C::cum(q2 + 1)) issue(s_c, (PAR + dt2) % NS, t + dt2);
      });
    }
    if (t + dtw < nk) {
      if constexpr (C::wait_is_prologue_only(PAR * NPH + q)) {
        if (behind_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::vm(q) + NSTORE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::vm(q)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::vm(q)) : "memory");
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NM = 2 * SQ * HT;
    __builtin_amdgcn_s_setprio(1);
    static_for<NM>([&](auto m_c) {
      constexpr int m = decltype(m_c)::value, ks = m / (SQ * HT), i = (m / HT) % SQ, h = m % HT;
      // operands swapped: lane (g, l15) accumulates C[row 16h + l15][columns 16(q SQ + i) + 4g .. + 3]
      acc[h][q * SQ + i] = Mma::mma(sf[i][ks], hf[h][ks], acc[h][q * SQ + i]);
    });
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  static_assert(!C::wait_is_prologue_only(NS * NPH), "store-aware waits must end inside the first pass over the ring");

  int m0, n0, it = 0;
  {
    int tm, tn;
    if (!tile_of(0, tm, tn)) return;
    set_tile(tm, tn, m0, n0);
  }
  prologue();
  bool have_prev = false;
  for (;;) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // everything the tile's first phase reads has landed; the previous tile's epilogue stores (issued behind the prologue) may still fly
    if (have_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::VM_PRO + NSTORE) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::VM_PRO) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }     // group 1 runs half a phase behind

    for (int tb = 0; tb < nk; tb += NS) {
      const bool behind = have_prev && tb == 0;
      static_for<NS>([&](auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        if (tb + PAR < nk) static_for<NPH>([&](auto q_c) { phase(par_c, q_c, tb + PAR, behind); });
      });
    }
    if (grp == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }     // barrier census; every LDS read of the tile is complete

    // ---------------- tile boundary: epilogue operands, then the next tile's prologue fetches, then the epilogue itself
    const int row_w = m0 + wm * C::RM + l15, col_w = n0 + wn * C::RN + 16 * (g & 1) + 8 * (g >> 1);      // store layout
    const int col_f = n0 + wn * C::RN + 4 * g;                                                              // accumulator layout
    u32x4 bias4[EPI != EPI_GELU_BWD ? NT : 1];
    if constexpr (EPI != EPI_GELU_BWD) {
      if (has_bias) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int c = col_f + 16 * j;
          bias4[j] = asm_buffer_load_b128(c < N ? (uint32_t)c * 4u : 0x7FFFFF00u, wBias);
        }
      }
    }
    u32x4 ax[EPI == EPI_GELU_BWD ? MT : 1][EPI == EPI_GELU_BWD ? NT / 2 : 1];
    if constexpr (EPI == EPI_GELU_BWD) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jp = 0; jp < NT / 2; ++jp) {
          const int r = row_w + 16 * i, c = col_w + 32 * jp;
          const uint32_t off = (r < M && c < N) ? (uint32_t)(((int64_t)r * ep.ldaux + c) * 2) : 0x7FFFFF00u;
          ax[i][jp] = asm_buffer_load_b128(off, wX);
        }
    }
    int ntm, ntn;
    const bool more = tile_of(it + 1, ntm, ntn);
    if (more) {
      set_tile(ntm, ntn, m0, n0);
      prologue();
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::pro_count()) : "memory");     // the epilogue operands are older than the prologue fetches
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (EPI != EPI_GELU_BWD) {
      if (has_bias) {
#pragma unroll
        for (int j = 0; j < NT; ++j) asm_after_wait(bias4[j]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jp = 0; jp < NT / 2; ++jp) asm_after_wait(ax[i][jp]);
    }

#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int jp = 0; jp < NT / 2; ++jp) {
        floatx4 v0 = acc[i][2 * jp], v1 = acc[i][2 * jp + 1];
        if constexpr (EPI != EPI_GELU_BWD) {
          if (has_bias) { v0 += __builtin_bit_cast(floatx4, bias4[2 * jp]); v1 += __builtin_bit_cast(floatx4, bias4[2 * jp + 1]); }
        }
        const int r = row_w + 16 * i, c = col_w + 32 * jp;
        const uint32_t off = (r < M && c < N) ? (uint32_t)(((int64_t)r * ep.ldo + c) * 2) : 0x7FFFFF00u;
        auto emit = [&](const __amdgpu_buffer_rsrc_t& rs, floatx4 a, floatx4 b) {
          const uint32_t x0 = pack2bf(a[0], a[1]), x1 = pack2bf(a[2], a[3]), y0 = pack2bf(b[0], b[1]), y1 = pack2bf(b[2], b[3]);
          const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
          const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
          __builtin_amdgcn_raw_buffer_store_b128(o, rs, off, 0, 0);
        };
        if constexpr (EPI == EPI_GELU || EPI == EPI_GELU_NOPRE) {
          if constexpr (EPI == EPI_GELU) emit(rsO2, v0, v1);     // pre-activation (what the backward's GELU' reads; not stored when no backward follows)
#pragma unroll
          for (int e = 0; e < 4; ++e) { v0[e] = gelu_fwd<bf16_t>(v0[e]); v1[e] = gelu_fwd<bf16_t>(v1[e]); }
        }
        if constexpr (EPI == EPI_GELU_BWD) {
          // the saved pre-activations arrive in the store layout: the same row exchange takes them back to the accumulator layout
          const auto s0 = __builtin_amdgcn_permlane16_swap(ax[i][jp][0], ax[i][jp][2], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(ax[i][jp][1], ax[i][jp][3], false, false);
          const uint32_t xs[4] = {s0[0], s1[0], s0[1], s1[1]};   // x0 (cols 0,1), x1 (2,3) of tile 2jp ; y0, y1 of tile 2jp + 1
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            v0[2 * e] *= gelu_bwd<bf16_t>(__uint_as_float(xs[e] << 16));
            v0[2 * e + 1] *= gelu_bwd<bf16_t>(__uint_as_float(xs[e] & 0xffff0000u));
            v1[2 * e] *= gelu_bwd<bf16_t>(__uint_as_float(xs[2 + e] << 16));
            v1[2 * e + 1] *= gelu_bwd<bf16_t>(__uint_as_float(xs[2 + e] & 0xffff0000u));
          }
        }
        emit(rsO, v0, v1);
        if constexpr (EPI == EPI_STORE_F32X) {
          // the fp32 copy carries the bf16-ROUNDED values (what a cast of the bf16 logits would hold): the two outputs agree bit for bit
          const int rf = row_w + 16 * i, cf = col_f + 32 * jp;
          auto rnd = [](floatx4 a) { floatx4 o; const uint32_t p0 = pack2bf(a[0], a[1]), p1 = pack2bf(a[2], a[3]);
                                     o[0] = __uint_as_float(p0 << 16); o[1] = __uint_as_float(p0 & 0xffff0000u);
                                     o[2] = __uint_as_float(p1 << 16); o[3] = __uint_as_float(p1 & 0xffff0000u); return o; };
          const uint32_t of0 = (rf < M && cf < N) ? (uint32_t)(((int64_t)rf * ep.ldo_f32 + cf) * 4) : 0xFFFFFF00u;
          const uint32_t of1 = (rf < M && cf + 16 < N) ? (uint32_t)(((int64_t)rf * ep.ldo_f32 + cf + 16) * 4) : 0xFFFFFF00u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rnd(v0)), rsF, of0, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rnd(v1)), rsF, of1, 0, 0);
        }
      }
    }
    if (!more) break;
    ++it;
    have_prev = true;
  }
}

template <typename C, int EPI>
__global__ void __launch_bounds__(512, 2)
gemm_nt8p_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int M, int N, int K, int tiles_n,
                 int ntiles, EpiParams<bf16_t> ep, int rows_per_xcd) {
  nt8p_body<C, EPI>(A, lda, B, ldb, M, N, K, tiles_n, ntiles, ep, rows_per_xcd);
}

//                BM   BN  WM WN hold_B SQ NS LEAD
typedef Nt8Cfg<256, 192, 4, 2, false, 2, 2, 4> PCfg256x192;

bool nt8p_supported(int M, int N, int K, const EpiParams<bf16_t>& ep, int64_t lda, int64_t ldb) {
  if (ep.col_scale != nullptr) return false;
  if (ep.out_f32 != nullptr && (ep.mode != EPI_STORE || ep.accumulate || (N % 4) != 0 || (ep.ldo_f32 % 4) != 0 || ep.ldo_f32 < N ||
                                (int64_t)M * ep.ldo_f32 * 4 >= 0xFFFFFF00ll)) return false;
  const bool mode_ok = (ep.mode == EPI_STORE && !ep.accumulate) || ep.mode == EPI_GELU ||
                       (ep.mode == EPI_GELU_BWD && !ep.accumulate && ep.aux != nullptr);
  return mode_ok && ep.alpha == 1.0f && ep.rm_hw_shift < 0 && (K % 64) == 0 && K >= 128 && (N % 8) == 0 && (ep.ldo % 8) == 0 &&
         (ep.aux == nullptr || (ep.ldaux % 8) == 0) && (lda % 8) == 0 && (ldb % 8) == 0 && M >= 1 && N >= 8 &&
         (int64_t)M * lda * 2 < 0x7FFFFF00ll && (int64_t)N * ldb * 2 < 0x7FFFFF00ll && (int64_t)M * ep.ldo * 2 < 0x7FFFFF00ll &&
         (ep.aux == nullptr || (int64_t)M * ep.ldaux * 2 < 0x7FFFFF00ll);
}

static int g_nt8p_order = 1;                       // 1: every XCD owns a band of tile rows (A panels L2-resident); 0: chunked ids
void set_nt8p_order(int o) { g_nt8p_order = o; }

template <typename C, int EPI>
static int launch_nt8p(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K,
                       const EpiParams<bf16_t>& ep, int max_wg) {
  const int tiles_m = (M + C::BM - 1) / C::BM, tiles_n = (N + C::BN - 1) / C::BN, ntiles = tiles_m * tiles_n;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt8p_kernel<C, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS); attr_set = true; }
  const int grid = ntiles < max_wg ? ntiles : max_wg;
  const int rows_per_xcd = (g_nt8p_order == 1 && (tiles_m % 8) == 0 && (grid % 8) == 0 && grid >= 8) ? tiles_m / 8 : 0;
  ProfScope ps(st, PK_GEMM_NT, 2.0 * M * N * K);
  if (ep.m_dev != nullptr) prof_set_exec(ep.m_dev, 2.0 * N * K, C::BM, M);
  RL_LAUNCH((gemm_nt8p_kernel<C, EPI>), dim3(grid), dim3(512), C::LDS, st, A, lda, B, ldb, M, N, K, tiles_n, ntiles, ep, rows_per_xcd);
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}

static int g_nt8p_wgs = 256;                       // one workgroup per CU
void set_nt8p_wgs(int n) { g_nt8p_wgs = n > 0 ? n : 256; }

int gemm_nt8p(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, const EpiParams<bf16_t>& ep) {
  if (!nt8p_supported(M, N, K, ep, lda, ldb)) return RL_ERR_ARG;
  switch (ep.mode) {
    case EPI_STORE:
      if (ep.out_f32 != nullptr) return launch_nt8p<PCfg256x192, EPI_STORE_F32X>(st, A, lda, B, ldb, M, N, K, ep, g_nt8p_wgs);
      return launch_nt8p<PCfg256x192, EPI_STORE>(st, A, lda, B, ldb, M, N, K, ep, g_nt8p_wgs);
    case EPI_GELU:
      if (ep.out2 == nullptr) return launch_nt8p<PCfg256x192, EPI_GELU_NOPRE>(st, A, lda, B, ldb, M, N, K, ep, g_nt8p_wgs);
      return launch_nt8p<PCfg256x192, EPI_GELU>(st, A, lda, B, ldb, M, N, K, ep, g_nt8p_wgs);
    case EPI_GELU_BWD: return launch_nt8p<PCfg256x192, EPI_GELU_BWD>(st, A, lda, B, ldb, M, N, K, ep, g_nt8p_wgs);
    default: return RL_ERR_ARG;
  }
}

}  // namespace rl
