// MEASURED VARIANT, not the production path.  Round 2-3: for single problems (realise_set_tn_variant(8), probe build) with a reduction
// split + slab fold it came out 7-16 % slower than the 4-wave kernel.  Round 4: its GROUPED form (gemm_tn8_group: the four weight
// gradients of a transformer layer, 216 tiles of 256 x 128 = one per CU, no split; realise_set_engine(7, 1)) on cold operands:
// 210 us against 200 for the 4-wave grouped launch (tools/tn_live_probe.py, profiles/round4_tn_live_probe.log) - the K-tile costs
// ~1.6 us in both, whatever the number of tiles in flight: not a fetch-latency problem.  Its live-block list form (a per-lane
// register re-loaded inside the issue path) makes the compiler drain vmcnt before every fetch: 555 us.  Kept for the record.
//
// Ping-pong 8-wave TN GEMM for gfx950 (bf16, dense operands): the weight gradients of the Linear layers,
//   C[I,J] (+)= sum_p A[p,i] * B[p,j]      A = dY [P, lda], B = X [P, ldb], both row-major with the reduction index p slowest.
//
// Same machine as gemm_nt8.hip - one 512-thread workgroup per CU, two wave groups half a phase apart, raw-buffer LDS-DMA fetches
// with SGPR descriptors, a 3-stage ring with counted vmcnt - with the two things a p-major operand changes:
//   * a K-tile is 64 reduction rows of the natural layout ([p][256 i] and [p][128 j]); the MFMA fragments (8 consecutive p for one
//     column) are transposed on the LDS read with ds_read_b64_tr_b16 (tn_frag_bf16), under the byte XOR tn_swz of the row;
//   * every 1-KiB piece (2 or 4 whole rows) is needed by every wave in the tile's FIRST phase, so the fetch schedule is the plain
//     three-stage one: tile t+2's six pieces per wave are issued across phases (t, 1) and (t+1, 0), waited for in (t+1, 1).
// The reduction is split over workgroups to fill the chip (grid = tiles x splits ~ 256): each split writes a dense fp32 slab and
// the fold kernel of gemm.hip adds them in a fixed order (bitwise reproducible); one split accumulates in place.  The bias
// gradient (column sums of A) rides along as one ones-vector MFMA per A fragment in the j-tile-0 workgroups.
#include "gemm_dev.h"
#include "prof.h"

namespace rl {

template <int N> struct ICt { static constexpr int value = N; };
template <int N, int... Is> struct SeqT : SeqT<N - 1, N - 1, Is...> {};
template <int... Is> struct SeqT<0, Is...> {
  template <typename F> static __device__ __forceinline__ void run(F&& f) { (f(ICt<Is>{}), ...); }
};
template <int N, typename F> __device__ __forceinline__ void static_for_t(F&& f) { SeqT<N>::run(f); }

template <int = 0> struct Tn8T {      // a template only so that the constexpr schedule functions are usable inside the class body
  static constexpr int BI = 256, BJ = 128, WI = 4, WJ = 2, BP = 64, NS = 3, SQ = 2;
  static constexpr int MT = 4, NT = 4;                         // per-wave 64 x 64: 4 x 4 MFMA tiles
  static constexpr int NPH = MT / SQ;                          // phases per K-tile (held: B fragments; streamed: A in groups of SQ)
  static constexpr int RPA = BI * 2, RPB = BJ * 2;             // row pitches in bytes
  static constexpr int A_BYTES = BP * RPA, B_BYTES = BP * RPB, STAGE = A_BYTES + B_BYTES;
  static constexpr int NPA = A_BYTES / 1024, NP = STAGE / 1024, NPW = NP / 8;     // 32 + 16 pieces, 6 per wave
  // Pieces of tile t are issued BETWEEN THE MFMAs of phases 2t-4 and 2t-3, waited for in the memory segment of phase 2t-1 and
  // read from phase 2t: two whole phases (~1000 clk) between the last issue and the wait.  (Issued from the memory segments the
  // earliest WAR-safe slots are 2t-3 / 2t-2, one phase before the wait: measured 1.13 us per K-tile = fetch time + MFMA time.)
  static constexpr int LEAD = 4;
  static constexpr int cum(int q) { return (q * NPW + NPH - 1) / NPH; }
  static_assert(NPH == 2 && NPW == 6 && NS * STAGE <= 160 * 1024, "geometry");
  // WAR: the stage of tile t was last read in phase 2(t-3)+1 = 2t-5 (both groups done by the barrier that ends it) and is first
  // re-filled from the MFMA segment of phase 2t-4, one full phase later.
  static constexpr int vm(int q) {                             // wait in the memory segment of phase q: issues of phase q-1 are the youngest
    const int dt1 = (q + 1) / NPH, slot = q + LEAD - 1, dt2 = slot / NPH, q2 = slot % NPH;
    return dt2 * NPW + cum(q2 + 1) - 1 - (dt1 * NPW + NPW - 1);
  }
  static constexpr int PRO_TILES = 2;      // ceil(LEAD / NPH)
  static constexpr bool in_prologue(int dt, int s) { int qi = 0; while (cum(qi + 1) <= s) ++qi; return dt * NPH + qi - LEAD < 0; }
  static constexpr int pro_count() { int n = 0; for (int dt = 0; dt < PRO_TILES; ++dt) for (int s = 0; s < NPW; ++s) if (in_prologue(dt, s)) ++n; return n; }
  static constexpr int VM_PRO = pro_count() - 1 - (NPW - 1);
  static_assert(vm(0) >= 0 && vm(1) >= 0 && VM_PRO >= 0, "schedule");
  static constexpr int LDS = NS * STAGE;
};
typedef Tn8T<> Tn8;

// `logical` = split * ntiles + tile of this workgroup.  ep.tile_list / ep.n_tiles with ep.list_rows == 16 (unsplit launches): the
// reduction runs over the LIVE 16-row blocks of the rows only, four of them (any four) per K-tile.  Every 1-KiB piece lies inside
// one 16-row quarter of the tile (A piece s of a wave: quarter s; its two B pieces: quarters wave / 4 and 2 + wave / 4), so a listed
// tile differs from a dense one in the SCALAR offset of the fetch alone: block index x 16 rows instead of tile index x 64 rows.
__device__ __forceinline__ void tn8_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int P, int I, int J,
                                         int tiles_j, int ntiles, int nsplit, int pchunk, int how, TnEpi ep, int logical) {
  typedef Tn8 C;
  typedef MmaBF16 Mma;
  constexpr int NPW = C::NPW, NPH = C::NPH, NS = C::NS, SQ = C::SQ, MT = C::MT, NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wi = wave & 3, wj = wave >> 2;                 // waves w and w+4 (the two groups on one SIMD) own the two j halves
  const int split = logical / ntiles, tile = logical - split * ntiles;
  const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
  const int i0 = ti * C::BI, j0 = tj * C::BJ;
  const int p_begin = split * pchunk;
  const int p_end = min(P, p_begin + pchunk);
  const bool listed = ep.tile_list != nullptr && ep.list_rows == 16 && nsplit == 1;
  const int n_list = listed ? min(*ep.n_tiles, P / 16) : 0;
  const int nk = listed ? (n_list + 3) >> 2 : (p_end > p_begin ? (p_end - p_begin + C::BP - 1) / C::BP : 0);

  // ---- pieces of this wave: s-th piece is global piece s*8 + wave; A pieces hold 2 rows x 512 B, B pieces 4 rows x 256 B
  int lo[NPW], prow[NPW];
  uint32_t go[NPW];
  bool colok[NPW];
#pragma unroll
  for (int s = 0; s < NPW; ++s) {
    const int p = s * 8 + wave;
    const bool is_b = p >= C::NPA;
    const int k = is_b ? p - C::NPA : p;
    const int rp = is_b ? C::RPB : C::RPA;
    const int off = k * 1024 + lane * 16;                  // byte offset inside the operand's tile image
    const int row = off / rp, cb = off - row * rp;
    const int scb = cb ^ (is_b ? tn_swz<bf16_t, C::RPB>(row) : tn_swz<bf16_t, C::RPA>(row));       // source column (bytes) of this LDS slot
    const int col = (is_b ? j0 : i0) + scb / 2;
    lo[s] = (is_b ? C::A_BYTES : 0) + k * 1024;
    prow[s] = row;
    colok[s] = col < (is_b ? J : I);
    go[s] = (uint32_t)((int64_t)(listed ? (row & 15) : row) * (is_b ? ldb : lda) * 2 + (int64_t)col * 2);
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)0xFFFFFE00u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)0xFFFFFE00u, 0x00020000);
  const uint32_t strideA = (uint32_t)(C::BP * lda * 2), strideB = (uint32_t)(C::BP * ldb * 2);
  const uint32_t baseA = (uint32_t)((int64_t)p_begin * lda * 2), baseB = (uint32_t)((int64_t)p_begin * ldb * 2);
  // listed mode: the list is copied to LDS once (TN_LIST_LDS bytes behind the ring) and read from there by a broadcast ds_read per
  // fetch.  Round 4 kept 64 entries in a register re-loaded INSIDE the issue path: a tracked global load in a conditional block, after
  // which hipcc could not count the fetch queue any more and drained it (vmcnt(0)) before every fetch - 555 us per launch listed, and
  // waits at the merges of the unlisted path too (gemm.hip tn_tile_body has the same story)
  int* lds_list = (int*)(smem + C::LDS);
  if (listed) {
    for (int e = tid; e < n_list; e += 512) lds_list[e] = ep.tile_list[e];
    __syncthreads();
  }
  auto issue = [&](auto s_c, int stage, int ktile) {
    constexpr int s = decltype(s_c)::value;
    constexpr bool is_b = (s * 8 >= C::NPA);                // pieces 0..31 are A, 32..47 B: s = 0..3 -> A, 4..5 -> B for every wave
    static_assert(C::NPA % 8 == 0, "operand boundary falls between two piece rounds");
    if (listed) {
      const int quarter = is_b ? (s - 4) * 2 + (wave >> 2) : s;
      const int e = (ktile << 2) + quarter;
      const int blk = e < n_list ? __builtin_amdgcn_readfirstlane(lds_list[e]) : -1;
      const uint32_t voff = (colok[s] && blk >= 0) ? go[s] : 0xFFFFFF00u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(is_b ? rsB : rsA, (__attribute__((address_space(3))) void*)(smem + stage * C::STAGE + lo[s]), 16,
                                               voff, (uint32_t)max(blk, 0) * ((is_b ? strideB : strideA) >> 2), 0, 0);
      return;
    }
    const bool live = colok[s] && (p_begin + ktile * C::BP + prow[s] < p_end);
    const uint32_t voff = live ? go[s] : 0xFFFFFF00u;       // rows past the split / columns past the matrix read zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(is_b ? rsB : rsA, (__attribute__((address_space(3))) void*)(smem + stage * C::STAGE + lo[s]), 16,
                                             voff, (is_b ? baseB : baseA) + (uint32_t)ktile * (is_b ? strideB : strideA), 0, 0);
  };

  floatx4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = (ep.colsum != nullptr) && (tj == 0) && (wj == 0);
  floatx4 csum[MT];
#pragma unroll
  for (int f = 0; f < MT; ++f) csum[f] = floatx4{0.f, 0.f, 0.f, 0.f};
  bf16x8_t ones;
  {
    typedef __attribute__((ext_vector_type(8))) short short8_t;
    const short8_t o = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    ones = __builtin_bit_cast(bf16x8_t, o);
  }
  bf16x8_t hf[NT][2], sf[SQ][2];

  // ---- prologue
  static_for_t<C::PRO_TILES>([&](auto dt_c) {
    constexpr int dt = decltype(dt_c)::value;
    if (dt < nk) {
      static_for_t<NPW>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (C::in_prologue(dt, s)) issue(s_c, dt % NS, dt);
      });
    }
  });
  if (nk >= C::PRO_TILES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::VM_PRO) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

  auto phase = [&](auto par_c, auto q_c, int t) {
    constexpr int PAR = decltype(par_c)::value, q = decltype(q_c)::value;
    constexpr int dt2 = (q + C::LEAD) / NPH, q2 = (q + C::LEAD) % NPH;
    const char* At = smem + PAR * C::STAGE;
    const char* Bt = At + C::A_BYTES;
    // ---------------- memory segment
    if constexpr (q == 0) {
#pragma unroll
      for (int h = 0; h < NT; ++h)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) hf[h][ks] = tn_frag_bf16<true, C::RPB>(Bt, ks, wj * 64 + h * 16, l15, g);
    }
#pragma unroll
    for (int i = 0; i < SQ; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) sf[i][ks] = tn_frag_bf16<true, C::RPA>(At, ks, wi * 64 + (q * SQ + i) * 16, l15, g);
    constexpr int dtw = (q + C::LEAD - 1) / NPH;            // tile of the youngest issue slot before this wait
    if (t + dtw < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::vm(q)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- MFMA segment, this phase's three fetches spread between the MFMAs
    const bool do_issue = t + dt2 < nk;
    __builtin_amdgcn_s_setprio(1);
    static_for_t<2 * SQ>([&](auto m_c) {
      constexpr int m = decltype(m_c)::value, ks = m / SQ, i = m % SQ;
#pragma unroll
      for (int h = 0; h < NT; ++h) acc[q * SQ + i][h] = Mma::mma(hf[h][ks], sf[i][ks], acc[q * SQ + i][h]);
      if (do_colsum) csum[q * SQ + i] = Mma::mma(ones, sf[i][ks], csum[q * SQ + i]);      // every row of the result = sum_p A[p, i]
      if constexpr (m < C::cum(q2 + 1) - C::cum(q2)) {
        __builtin_amdgcn_sched_barrier(0);
        if (do_issue) issue(ICt<C::cum(q2) + m>{}, (PAR + dt2) % NS, t + dt2);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int tb = 0; tb < nk; tb += NS) {
    static_for_t<NS>([&](auto par_c) {
      constexpr int PAR = decltype(par_c)::value;
      if (tb + PAR < nk) static_for_t<NPH>([&](auto q_c) { phase(par_c, q_c, tb + PAR); });
    });
  }
  if (grp == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
      tn_epilogue4(ep, how, split, I, J, i0 + wi * 64 + i * 16 + l15, j0 + wj * 64 + j * 16 + 4 * g, acc[i][j]);
  if (do_colsum && g == 0) {
#pragma unroll
    for (int f = 0; f < MT; ++f) {
      const int i = i0 + wi * 64 + f * 16 + l15;
      if (i < I) atomicAdd(ep.colsum + i, csum[f][0] * tn_alpha(ep));
    }
  }
}

__global__ void __launch_bounds__(512)
gemm_tn8_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int P, int I, int J, int tiles_j,
                int ntiles, int nsplit, int pchunk, int how, TnEpi ep) {
  tn8_body(A, lda, B, ldb, P, I, J, tiles_j, ntiles, nsplit, pchunk, how, ep, xcd_remap(blockIdx.x, ntiles * nsplit));
}

// Grouped form (the four weight gradients of a transformer layer in one launch, one 256 x 128 tile per workgroup over the whole
// reduction - 216 tiles, one per CU): the 8-wave three-stage body keeps two K-tiles (96 KB) in flight per CU where the two co-resident
// 4-wave workgroups of gemm_tn_group_kernel keep one each (64 KB), and a fetched byte feeds a third more flops (85 vs 64 per byte).
__global__ void __launch_bounds__(512)
gemm_tn8_group_kernel(TnGroup<bf16_t> grp, int P, int pchunk) {
  const int logical = xcd_remap(blockIdx.x, grp.total_tiles);
  int k = 0;
#pragma unroll
  for (int i = 1; i < TN_GROUP_MAX; ++i) if (i < grp.n && logical >= grp.p[i].tile_begin) k = i;
  const TnGroupProblem<bf16_t>& pr = grp.p[k];
  TnEpi ep;
  ep.out = pr.out; ep.ldo = pr.ldo; ep.colsum = pr.colsum; ep.alpha = grp.alpha; ep.overwrite = grp.overwrite;
  ep.tile_list = grp.tile_list; ep.n_tiles = grp.n_tiles; ep.list_rows = grp.list_rows;
  tn8_body(pr.A, pr.lda, pr.B, pr.ldb, P, pr.I, pr.J, pr.tiles_j, pr.ntiles, 1, pchunk, TN_OUT_DIRECT, ep, logical - pr.tile_begin);
}

// RL_ERR_ARG when a problem does not fit the 8-wave kernel (the caller then takes the 4-wave grouped launch)
int gemm_tn8_group(hipStream_t st, int n, const TnGroupProblem<bf16_t>* probs, int P, float alpha, int overwrite, const int* tile_list,
                   const int* n_tiles, int list_rows) {
  typedef Tn8 C;
  if (n < 1 || n > TN_GROUP_MAX || P < 1024 || (P % C::BP) != 0) return RL_ERR_ARG;
  if (tile_list != nullptr && (n_tiles == nullptr || list_rows != 16)) return RL_ERR_ARG;      // (whole-tile lists: the 4-wave kernel)
  TnGroup<bf16_t> grp;
  grp.n = n; grp.alpha = alpha; grp.overwrite = overwrite;
  if (tile_list != nullptr) { grp.tile_list = tile_list; grp.n_tiles = n_tiles; grp.list_rows = 16; }
  int total = 0;
  double flops = 0.0;
  for (int k = 0; k < n; ++k) {
    TnGroupProblem<bf16_t> pr = probs[k];
    TnEpi te; te.ldo = pr.ldo;
    if (!tn8_supported(pr.lda, pr.ldb, P, pr.I, pr.J, te)) return RL_ERR_ARG;
    pr.tiles_j = (pr.J + C::BJ - 1) / C::BJ;
    pr.ntiles = ((pr.I + C::BI - 1) / C::BI) * pr.tiles_j;
    pr.tile_begin = total;
    total += pr.ntiles;
    flops += 2.0 * P * pr.I * pr.J;
    grp.p[k] = pr;
  }
  grp.total_tiles = total;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_tn8_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS + TN_LIST_LDS_MAX); attr_set = true; }
  const int list_lds = grp.tile_list != nullptr ? tn_list_lds_bytes(P / 16) : 0;
  if (list_lds < 0) return RL_ERR_ARG;
  ProfScope ps(st, PK_GEMM_TN, flops);
  if (grp.tile_list != nullptr) prof_set_exec(grp.n_tiles, flops / P * 16, 4, P / 16);
  RL_LAUNCH(gemm_tn8_group_kernel, dim3(total), dim3(512), C::LDS + list_lds, st, grp, P, P);
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}

bool tn8_supported(int64_t lda, int64_t ldb, int P, int I, int J, const TnEpi& ep) {
  // (a live-block list needs the grouped launch's LDS list area: ep.tile_list must be null here)
  return ep.mode == TN_PLAIN && ep.tile_list == nullptr && P >= 1024 && I >= 256 && J >= 128 && (J % 4) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && (I % 8) == 0 &&
         (J % 8) == 0 && (int64_t)P * lda * 2 < 0xFFFFFE00ll && (int64_t)P * ldb * 2 < 0xFFFFFE00ll && (ep.ldo % 4) == 0;
}

int gemm_tn8(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int P, int I, int J, const TnEpi& ep_, int force_split) {
  typedef Tn8 C;
  TnEpi ep = ep_;
  const int tiles_i = (I + C::BI - 1) / C::BI, tiles_j = (J + C::BJ - 1) / C::BJ, ntiles = tiles_i * tiles_j;
  // splits: fill the 256 CUs once (rounds of 256 one-per-CU workgroups), at least 8 K-tiles per workgroup, slabs must fit
  int64_t cap = P / (8 * C::BP);
  if (ep.slab != nullptr && ep.slab_elems / ((int64_t)I * J) < cap) cap = ep.slab_elems / ((int64_t)I * J);
  if (ep.slab == nullptr || cap < 1) cap = 1;
  int nsplit = 1, pchunk = ((P + C::BP - 1) / C::BP) * C::BP;
  {   // cost in microseconds: rounds x (K-tiles x 0.7 + fixed 4) + slab write / fold read at ~5.5 TB/s + the fold launch
    double best = 1e30;
    for (int ns = 1; ns <= (int)cap; ++ns) {
      if (force_split > 0 && ns != (force_split < (int)cap ? force_split : (int)cap)) continue;
      const int pc = (((P + ns - 1) / ns + C::BP - 1) / C::BP) * C::BP;
      const int ne = (P + pc - 1) / pc;
      const double rounds = (double)(((int64_t)ntiles * ne + 255) / 256);
      const double c = rounds * ((double)pc / C::BP * 0.7 + 4.0) + (ne > 1 ? ne * (double)I * J * 8.0 / 5.5e6 + 4.0 : 0.0);
      if (c < best) { best = c; nsplit = ne; pchunk = pc; }
    }
  }
  const int how = nsplit == 1 ? TN_OUT_DIRECT : TN_OUT_SLAB;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_tn8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS); attr_set = true; }
  ProfScope ps(st, PK_GEMM_TN, 2.0 * P * I * J);
  RL_LAUNCH(gemm_tn8_kernel, dim3(ntiles * nsplit), dim3(512), C::LDS, st, A, lda, B, ldb, P, I, J, tiles_j, ntiles, nsplit, pchunk, how, ep);
  if (how == TN_OUT_SLAB) tn_fold_launch(st, ep, nsplit, I, J);
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}

}  // namespace rl
