// Stream-K persistent NT GEMM for gfx950 (bf16, K % 128 == 0):  C[M,N] = A[M,K] . B[N,K]^T + fused epilogue over 256 x 192 tiles,
// the M dimension optionally a device-side list of live 16-row blocks (EpiParams::live_list, as gemm_nt8_live).
//
// Why: the layer GEMMs of the transformer stacks are one-round launches.  On the two-per-CU
// 128 x 192 shape a K-tile costs 40 KB of L1 fill per 3.1 MFLOP and the fill path (~24 B/clk/CU) is what bounds it; the 256 x 192
// tile needs 30 % fewer bytes per flop (gemm_nt8p: 1.1 PF on the classifier) but 5808 live rows x 768 .. 3072 columns are 92 .. 368
// such tiles for 256 CUs: 0.36 .. 1.44 rounds.  Here the launch is ONE round of 256 workgroups by construction: the iteration space
// (tile, K-tile), tile-major, is cut into 256 equal contiguous ranges (even cut points: a segment holds >= 2 K-tiles), workgroup L
// (XCD-contiguous logical id) walks its range as <= 3 SEGMENTS = (tile, first K-tile, K-tiles):
//   * a segment that starts inside a tile (k > 0) is a PARTIAL: its accumulators go to the workgroup's slot of the exchange buffer as
//     they sit in the registers (24 x 16 B per lane, system-scope write-through stores), then - stores acknowledged, workgroup
//     barrier - the slot's flag takes the launch tag (agent-scope store).  It is the FIRST segment of its range: the tile's finisher
//     needs it last;
//   * a segment that starts a tile and ends before its last K-tile is the tile's FINISHER and the last of its range: it polls the
//     flags of the following workgroups (bounded; EpiParams::sk_timeout reports a wait that gave up), adds their slots in workgroup
//     order (cache-bypassing loads) and runs the epilogue.  The order of the additions is fixed by (live count, N, K, grid): a launch
//     is reproducible bit for bit, but a tile cut here sums its K range in two or three chains - the results differ in the last bits
//     from the one-chain kernels (gemm_nt8, gemm_nt8p) and, under a live list, from the dense launch (other cut points);
//   * whole tiles in between run the gemm_nt8p way: next segment's first fetches before the epilogue math, stores drain under its loop.
// Progress: a finisher only waits for partials that their writers produce FIRST, so whatever subset of the grid is resident, every
// resident workgroup but those whose successor is not yet dispatched completes and frees its CU.
// Main loop, fetch schedule, hazard rules and the register epilogue are gemm_nt8p.hip's (nt8_cfg.h); epilogues: bias (+ accumulate),
// bias + GELU (+ saved pre-activation), GELU' backward, bias + dropout + residual - what the layers call (engine.hip nt_rows).
#include "gemm_dev.h"
#include "nt8_cfg.h"
#include "prof.h"

namespace rl {

typedef __attribute__((ext_vector_type(4))) uint32_t sk_u32x4;
typedef __attribute__((ext_vector_type(4))) int sk_i32x4;

__device__ __forceinline__ sk_u32x4 sk_asm_load_b128(uint32_t voff, sk_i32x4 rsrc) {
  sk_u32x4 v = {0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
  // s_nop 4: the descriptor may have been restored from a spill (v_readlane_b32) right in front of the statement, and the 5 wait states
  // a vector-memory instruction needs behind a VALU write of an SGPR it reads are not padded for asm text (tools/isa_hazard_scan.py)
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(v) : "v"(voff), "s"(rsrc) : "memory");
#endif
  return v;
}
__device__ __forceinline__ void sk_after_wait(sk_u32x4& v) {      // pins the uses of an asm-loaded value behind the preceding s_waitcnt
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
}

// Four list entries by scalar loads.  Behind the kernel's own stores hipcc no longer proves a uniform global load unclobbered and falls
// back to vector loads - tracked on vmcnt next to the LDS-DMA fetches in flight, i.e. waited for with vmcnt(0) at their first use (the
// epilogue's address arithmetic: the next segment's prologue fetches drained in front of the epilogue math).
__device__ __forceinline__ void sk_sload4(const int* p0, const int* p1, const int* p2, const int* p3, int& a, int& b, int& c, int& d) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %5, 0x0\n\ts_load_dword %2, %6, 0x0\n\ts_load_dword %3, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(p0), "s"(p1), "s"(p2), "s"(p3) : "memory");
#else
  a = *p0; b = *p1; c = *p2; d = *p3;
#endif
}

constexpr uint32_t SK_PARK = 0x7FFFFF00u;      // beyond every num_records below: loads return zeros, stores are dropped
constexpr int SK_CP = 0x11;                     // cache policy of the exchange traffic: sc0 sc1 (system scope: write-through / miss always)

template <typename C, int EPI, bool ACC>
__device__ __forceinline__ void nt8s_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int M, int N,
                                          int K, int tiles_n, const EpiParams<bf16_t>& ep) {
  typedef MmaBF16 Mma;
  constexpr int NPW = C::NPW, NPH = C::NPH, NS = C::NS, LEAD = C::LEAD, SQ = C::SQ, HT = C::HT, MT = C::MT, NT = C::NT, HPW = C::HPW;
  static_assert(!C::HOLD_B && C::ISSUE_AT == 0 && C::FW == 8 && (NT % 2) == 0 && C::BM == 256 && C::WN == 2, "stream-K kernel: the persistent 256-row hold-A schedule");
  static_assert(EPI == EPI_STORE || !ACC, "accumulating form: plain store only");
  constexpr bool PRE = ACC || EPI == EPI_GELU_BWD || EPI == EPI_DROP_RESID;      // a second operand in the store layout (old output / saved x / residual)
  constexpr int NSTORE = MT * (NT / 2) * (EPI == EPI_GELU ? 2 : 1);              // 16-byte stores of one tile's epilogue, per lane
  constexpr int NPART = MT * NT;                                                  // 16-byte pieces of a partial tile, per lane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave / C::WN, wn = wave - wm * C::WN;
  const int nk = K >> 6;
  const int G = gridDim.x;                                                        // a multiple of 8
  const int L = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);                  // XCD x (= blockIdx % 8) owns an eighth of the iteration space
  const bool listed = ep.live_list != nullptr;
  const int nblk = listed ? *ep.live_count : (M + 15) >> 4;                       // 16-row blocks of the launch
  const int T = ((nblk + 15) >> 4) * tiles_n * nk;                                // K-tiles of the launch (nk even: T even)
  auto bound = [&](int j) -> int { return j >= G ? T : (int)((((uint32_t)j * (uint32_t)T) / (uint32_t)G) & ~1u); };      // (T x G < 2^31: nt8s_supported)
  int cur = bound(L);
  const int end = bound(L + 1);
  if (cur >= end) return;                                                         // (the whole workgroup leaves before any barrier)

  const int lrow = lane >> 3;
  const int kchunk_b = (((lane & 7) ^ lrow) << 4);
  int lo[NPW];
  uint32_t poB[NPW];
#pragma unroll
  for (int s = 0; s < NPW; ++s) {
    const int p = s * 8 + wave;
    int row, is_b;
    if (s < HPW) { row = p * 8; is_b = 0; }
    else {
      const int pp = p - C::HP, q = pp / C::GP, rem = pp - q * C::GP, slice = rem / (SQ * 2), j = rem - slice * (SQ * 2);
      row = slice * C::SR + q * SQ * 16 + j * 8; is_b = 1;
    }
    lo[s] = (is_b ? C::A_BYTES : 0) + row * 128;
    poB[s] = is_b ? (uint32_t)((int64_t)row * ldb * 2) : 0u;
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(((int64_t)(M - 1) * lda + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(((int64_t)(N - 1) * ldb + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)ep.out, 0, (int)(((int64_t)(M - 1) * ep.ldo + N) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == EPI_GELU ? ep.out2 : ep.out), 0, (int)(((int64_t)(M - 1) * ep.ldo + N) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)ep.sk_part, 0, (int)((int64_t)G * NPART * 512 * 16), 0x00020000);
  auto rsrc_words = [](const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    sk_i32x4 r = {(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
    return r;
  };
  const bool has_bias = EPI != EPI_GELU_BWD && ep.bias != nullptr;
  const sk_i32x4 wBias = rsrc_words(has_bias ? (const void*)ep.bias : (const void*)A, has_bias ? (uint32_t)N * 4u : 0u);
  const void* xsrc = ACC ? (const void*)ep.out : (const void*)ep.aux;
  const int64_t ldx = ACC ? ep.ldo : ep.ldaux;
  const sk_i32x4 wX = rsrc_words(PRE ? xsrc : (const void*)A, PRE ? (uint32_t)(((int64_t)(M - 1) * ldx + N) * 2) : 0u);

  // ---- segment state
  int tile = 0, kb = 0, klen = 0, tm = 0, tn = 0;
  auto decode = [&](int c) {
    tile = c / nk; kb = c - tile * nk;
    const int te = (tile + 1) * nk;
    klen = (end < te ? end : te) - c;
    tm = tile / tiles_n; tn = tile - tm * tiles_n;
  };
  // original 16-row blocks of the launch's blocks j0, j0 + dj, j0 + 2 dj, j0 + 3 dj; -1 beyond the count
  auto blocks_of = [&](int j0, int dj, int (&blk)[4]) {
    if (listed) {
      const int last = nblk - 1;                           // (nblk >= 1 here: an empty launch left above)
      const int* q = ep.live_list;
      sk_sload4(q + min(j0, last), q + min(j0 + dj, last), q + min(j0 + 2 * dj, last), q + min(j0 + 3 * dj, last), blk[0], blk[1], blk[2], blk[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) blk[i] = j0 + i * dj;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) if (j0 + i * dj >= nblk) blk[i] = -1;
  };
  static_assert(HPW == 4 && MT == 4, "four pieces of A per wave and K-tile, four 16-row blocks per wave");
  uint32_t goA[HPW], goB = 0;
  int n0 = 0;
  auto set_tile = [&]() {
    n0 = tn * C::BN;
    goB = (uint32_t)((int64_t)(n0 + lrow) * ldb * 2 + kchunk_b);
    int blk[4];                                           // piece s = rows 8p .. 8p + 7 of the tile, p = 8s + wave: half of block 4s + (wave >> 1)
    blocks_of(tm * 16 + (wave >> 1), 4, blk);
#pragma unroll
    for (int s = 0; s < HPW; ++s)
      goA[s] = blk[s] < 0 ? 0xFFFFFF00u : (uint32_t)((int64_t)(blk[s] * 16 + (wave & 1) * 8 + lrow) * lda * 2 + kchunk_b);
  };
  auto issue = [&](auto s_c, int stage, int ktile) {
    constexpr int s = decltype(s_c)::value;
    constexpr bool is_b = s >= HPW;
    uint32_t voff;
    if constexpr (is_b) voff = goB + poB[s]; else voff = goA[s];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(is_b ? rsB : rsA, (__attribute__((address_space(3))) void*)(smem + stage * C::STAGE + lo[s]), 16,
                                             voff, (kb + ktile) * 128, 0, 0);
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

  // one phase of the K loop (gemm_nt8p.hip; nk there = this segment's klen)
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
    if (t + dt2 < klen) {
      static_for<NPW>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (s >= C::cum(q2) && s < C::cum(q2 + 1)) issue(s_c, (PAR + dt2) % NS, t + dt2);
      });
    }
    if (t + dtw < klen) {
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

  decode(cur);
  set_tile();
  prologue();
  bool have_prev = false;                    // the previous segment's NSTORE epilogue stores sit behind this segment's prologue fetches
  for (;;) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (have_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::VM_PRO + NSTORE) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::VM_PRO) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }     // group 1 runs half a phase behind

    for (int tb = 0; tb < klen; tb += NS) {
      const bool behind = have_prev && tb == 0;
      static_for<NS>([&](auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        if (tb + PAR < klen) static_for<NPH>([&](auto q_c) { phase(par_c, q_c, tb + PAR, behind); });
      });
    }
    if (grp == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }     // barrier census; every LDS read of the segment is complete

    // ---------------- segment boundary
    const int nxt = cur + klen;
    const bool more = nxt < end;
    const bool writer = kb > 0;                            // PARTIAL: the segment starts inside its tile
    const int tile_end = (tile + 1) * nk;
    {
      // FINISHER (a segment that starts its tile and ends before the tile's last K-tile; the last of the range): the rest of the K range
      // sits in the slots of the following workgroups.  (One loop for every segment: it runs zero times for the others.)
      int cover = writer ? tile_end : nxt, j = L;
      while (cover < tile_end) {
        int b1;
        for (;;) {                                         // the next workgroup with a non-empty range: it starts at `cover`
          ++j;
          b1 = bound(j + 1);
          if (b1 > bound(j) || j >= G) break;
        }
        // one wave polls (every wave of every finisher polling kept a few memory channels busy with nothing else: launches of 50 ms,
        // one in two thousand - tools/streamk_probe.py stress), a microsecond between polls; the flags sit 256 bytes apart
        if (wave == 0) {
          int spins = 0;
          for (;;) {
            const int f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ep.sk_flag + j * NT8S_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (f == ep.sk_tag) break;
            if (++spins > (1 << 17)) {                     // (the writer never came: a broken launch - flag it and leave with what there is)
              if (ep.sk_timeout != nullptr && lane == 0) *ep.sk_timeout = 1;
              break;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (drop what this XCD's L2 holds of other XCDs' lines before the next look)
            __builtin_amdgcn_s_sleep(32);
          }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < MT; i += 2) {                  // two batches of twelve 16-byte loads in flight per lane
          sk_u32x4 pp[2][NT];
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < NT; ++jj)
              pp[ii][jj] = __builtin_amdgcn_raw_buffer_load_b128(rsP, (uint32_t)tid * 16u, (j * NPART + (i + ii) * NT + jj) * 8192, SK_CP);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) asm volatile("" : "+v"(pp[ii][jj]));
#endif
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) acc[i + ii][jj] += __builtin_bit_cast(floatx4, pp[ii][jj]);
        }
        cover = b1 < tile_end ? b1 : tile_end;
      }
    }

    // ---------------- the current tile's epilogue context and operands (not for a partial), the next segment's prologue fetches, then
    //                  the epilogue itself (gemm_nt8p.hip) or the partial's way out
    int rowE[MT];                                          // the lane's row in each of the wave's four 16-row blocks, -1: no such row
    {
      int blk[4];
      blocks_of(tm * 16 + wm * MT, 1, blk);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int r = blk[i] * 16 + l15;
        rowE[i] = (blk[i] >= 0 && r < M) ? r : -1;
      }
    }
    const int col_w = n0 + wn * C::RN + 16 * (g & 1) + 8 * (g >> 1);      // store layout
    const int col_f = n0 + wn * C::RN + 4 * g;                              // accumulator layout
    sk_u32x4 bias4[EPI != EPI_GELU_BWD ? NT : 1];
    sk_u32x4 ax[PRE ? MT : 1][PRE ? NT / 2 : 1];
    // (requested for a partial too, whose way out never reads them: a conditional definition makes them phi values, and hipcc resolves a
    //  phi with a register COPY - of a value whose asm load it cannot see is still in flight.  Measured: wrong, run-to-run different bits.)
    if constexpr (EPI != EPI_GELU_BWD) {
      if (has_bias) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int c = col_f + 16 * j;
          bias4[j] = sk_asm_load_b128(c < N ? (uint32_t)c * 4u : SK_PARK, wBias);
        }
      }
    }
    if constexpr (PRE) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jp = 0; jp < NT / 2; ++jp) {
          const int c = col_w + 32 * jp;
          const uint32_t off = (rowE[i] >= 0 && c < N) ? (uint32_t)(((int64_t)rowE[i] * ldx + c) * 2) : SK_PARK;
          ax[i][jp] = sk_asm_load_b128(off, wX);
        }
    }
    if (more) { decode(nxt); set_tile(); prologue(); }
    if (writer) {
      // PARTIAL: the register image of the accumulators into slot L, then - stores acknowledged by memory, every wave's - the flag
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sk_u32x4, acc[i][j]), rsP, (uint32_t)tid * 16u, (L * NPART + i * NT + j) * 8192, SK_CP);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the next segment's first K-tiles are in the LDS with it)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (tid == 0) __hip_atomic_store(ep.sk_flag + L * NT8S_FLAG_STRIDE, ep.sk_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::pro_count()) : "memory");     // the epilogue operands are older than the prologue fetches
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (EPI != EPI_GELU_BWD) {
        if (has_bias) {
#pragma unroll
          for (int j = 0; j < NT; ++j) sk_after_wait(bias4[j]);
        }
      }
      if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int jp = 0; jp < NT / 2; ++jp) sk_after_wait(ax[i][jp]);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int jp = 0; jp < NT / 2; ++jp) {
          floatx4 v0 = acc[i][2 * jp], v1 = acc[i][2 * jp + 1];
          if constexpr (EPI != EPI_GELU_BWD) {
            if (has_bias) { v0 += __builtin_bit_cast(floatx4, bias4[2 * jp]); v1 += __builtin_bit_cast(floatx4, bias4[2 * jp + 1]); }
          }
          const int c = col_w + 32 * jp;
          const uint32_t off = (rowE[i] >= 0 && c < N) ? (uint32_t)(((int64_t)rowE[i] * ep.ldo + c) * 2) : SK_PARK;
          auto emit = [&](const __amdgpu_buffer_rsrc_t& rs, floatx4 a, floatx4 b) {
            const uint32_t x0 = pack2bf(a[0], a[1]), x1 = pack2bf(a[2], a[3]), y0 = pack2bf(b[0], b[1]), y1 = pack2bf(b[2], b[3]);
            const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
            const sk_u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
            __builtin_amdgcn_raw_buffer_store_b128(o, rs, off, 0, 0);
          };
          float x0f[4] = {0.f, 0.f, 0.f, 0.f}, x1f[4] = {0.f, 0.f, 0.f, 0.f};      // the second operand, back in the accumulator layout
          if constexpr (PRE) {
            const auto s0 = __builtin_amdgcn_permlane16_swap(ax[i][jp][0], ax[i][jp][2], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(ax[i][jp][1], ax[i][jp][3], false, false);
            const uint32_t xs[4] = {s0[0], s1[0], s0[1], s1[1]};   // columns (0,1), (2,3) of tile 2jp ; of tile 2jp + 1
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              x0f[2 * e] = __uint_as_float(xs[e] << 16); x0f[2 * e + 1] = __uint_as_float(xs[e] & 0xffff0000u);
              x1f[2 * e] = __uint_as_float(xs[2 + e] << 16); x1f[2 * e + 1] = __uint_as_float(xs[2 + e] & 0xffff0000u);
            }
          }
          if constexpr (EPI == EPI_GELU) {
            emit(rsO2, v0, v1);                                    // pre-activation
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = gelu_fwd<bf16_t>(v0[e]); v1[e] = gelu_fwd<bf16_t>(v1[e]); }
          }
          if constexpr (EPI == EPI_GELU_BWD) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] *= gelu_bwd<bf16_t>(x0f[e]); v1[e] *= gelu_bwd<bf16_t>(x1f[e]); }
          }
          if constexpr (EPI == EPI_DROP_RESID) {
            const uint32_t idx = (uint32_t)(rowE[i] < 0 ? 0 : rowE[i]) * (uint32_t)N + (uint32_t)(col_f + 32 * jp);
            v0 *= drop_mult4(ep.drop_seed, ep.drop_thresh, ep.drop_scale, idx);
            v1 *= drop_mult4(ep.drop_seed, ep.drop_thresh, ep.drop_scale, idx + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] += x0f[e]; v1[e] += x1f[e]; }
          }
          if constexpr (ACC) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] += x0f[e]; v1[e] += x1f[e]; }
          }
          emit(rsO, v0, v1);
        }
      }
    }
    if (!more) break;
    cur = nxt;
    have_prev = !writer;
  }
}

template <typename C, int EPI, bool ACC>
__global__ void __launch_bounds__(512, 2)
gemm_nt8s_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int M, int N, int K, int tiles_n,
                 EpiParams<bf16_t> ep) {
  nt8s_body<C, EPI, ACC>(A, lda, B, ldb, M, N, K, tiles_n, ep);
}

//                BM   BN  WM WN hold_B SQ NS LEAD
typedef Nt8Cfg<256, 192, 4, 2, false, 2, 2, 4> SCfg256x192;

bool nt8s_supported(int M, int N, int K, const EpiParams<bf16_t>& ep, int64_t lda, int64_t ldb) {
  const bool mode_ok = (ep.mode == EPI_STORE) || (ep.mode == EPI_GELU && ep.out2 != nullptr && !ep.accumulate) ||
                       (ep.mode == EPI_GELU_BWD && !ep.accumulate && ep.aux != nullptr) ||
                       (ep.mode == EPI_DROP_RESID && !ep.accumulate && ep.aux != nullptr);
  const bool list_ok = (ep.live_list == nullptr) == (ep.live_count == nullptr) && (ep.live_list == nullptr || (M % 16) == 0);
  return mode_ok && list_ok && ep.out != nullptr && ep.alpha == 1.0f && ep.rm_hw_shift < 0 && ep.m_dev == nullptr && ep.slab == nullptr &&
         ep.ln_y == nullptr && ep.gru_table == nullptr && ep.sk_part != nullptr && ep.sk_flag != nullptr && ep.sk_tag != 0 &&
         (K % 128) == 0 && K >= 128 && (N % 8) == 0 && (ep.ldo % 8) == 0 && (ep.aux == nullptr || (ep.ldaux % 8) == 0) && (lda % 8) == 0 &&
         (ldb % 8) == 0 && lda >= K && ldb >= K && M >= 1 && N >= 8 &&
         (int64_t)M * lda * 2 < 0x7FFFFF00ll && (int64_t)N * ldb * 2 < 0x7FFFFF00ll && (int64_t)M * ep.ldo * 2 < 0x7FFFFF00ll &&
         (ep.aux == nullptr || (int64_t)M * ep.ldaux * 2 < 0x7FFFFF00ll) &&
         (int64_t)((M + 255) / 256) * ((N + 191) / 192) * (K / 64) * NT8S_GRID < (1ll << 31);
}

template <typename C, int EPI, bool ACC>
static int launch_nt8s(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, const EpiParams<bf16_t>& ep) {
  const int tiles_n = (N + C::BN - 1) / C::BN;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt8s_kernel<C, EPI, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS); attr_set = true; }
  ProfScope ps(st, PK_GEMM_NT, 2.0 * M * N * K);
  if (ep.live_count != nullptr) prof_set_exec(ep.live_count, 2.0 * N * K * 16.0, C::BM / 16, M / 16);      // (counted in 16-row blocks, whole 256-row tiles)
  RL_LAUNCH((gemm_nt8s_kernel<C, EPI, ACC>), dim3(NT8S_GRID), dim3(512), C::LDS, st, A, lda, B, ldb, M, N, K, tiles_n, ep);
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}

int gemm_nt8s(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, const EpiParams<bf16_t>& ep) {
  if (!nt8s_supported(M, N, K, ep, lda, ldb)) return RL_ERR_ARG;
  typedef SCfg256x192 C;
  switch (ep.mode) {
    case EPI_STORE:
      return ep.accumulate ? launch_nt8s<C, EPI_STORE, true>(st, A, lda, B, ldb, M, N, K, ep) : launch_nt8s<C, EPI_STORE, false>(st, A, lda, B, ldb, M, N, K, ep);
    case EPI_GELU: return launch_nt8s<C, EPI_GELU, false>(st, A, lda, B, ldb, M, N, K, ep);
    case EPI_GELU_BWD: return launch_nt8s<C, EPI_GELU_BWD, false>(st, A, lda, B, ldb, M, N, K, ep);
    case EPI_DROP_RESID: return launch_nt8s<C, EPI_DROP_RESID, false>(st, A, lda, B, ldb, M, N, K, ep);
    default: return RL_ERR_ARG;
  }
}

}  // namespace rl
