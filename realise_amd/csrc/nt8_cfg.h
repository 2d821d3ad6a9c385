// Compile-time fetch schedule of the 8-wave ping-pong NT kernels (gemm_nt8.hip, gemm_nt8p.hip): tile geometry, the order and issue
// slot of every 1-KiB LDS-DMA piece, the counted vmcnt of every phase, and the RAW / WAR phase rules as static_asserts.
#pragma once
#include "gemm_dev.h"

namespace rl {

template <int N> struct IC { static constexpr int value = N; };
template <int N, int... Is> struct SeqGen : SeqGen<N - 1, N - 1, Is...> {};
template <int... Is> struct SeqGen<0, Is...> {
  template <typename F> static __device__ __forceinline__ void run(F&& f) { (f(IC<Is>{}), ...); }
};
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { SeqGen<N>::run(f); }

template <int BM_, int BN_, int WM_, int WN_, bool HOLD_B_, int SQ_, int NS_, int LEAD_, int ISSUE_AT_ = 0, int FW_ = 8, int WGS_ = 1>
struct Nt8Cfg {
  static constexpr int WGS = WGS_;               // workgroups meant to share a CU (2: <= 80 KB of LDS and <= 128 VGPRs each)
  static constexpr int FW = FW_;                 // waves that issue the fetches: all 8 (they also multiply), or 4 dedicated loader waves
  // where a phase issues its fetches: 0 end of the memory segment (after the fragment reads), 1 between the MFMAs, 2 head of the
  // memory segment (the texture-address unit serialises the 4 waves' 1-KiB requests, ~29 clk each: issuing them FIRST lets that
  // queueing run under the fragment reads instead of after them)
  static constexpr int ISSUE_AT = ISSUE_AT_;
  static constexpr bool ISSUE_C = ISSUE_AT_ == 1;
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, SQ = SQ_, NS = NS_, LEAD = LEAD_;
  static constexpr bool HOLD_B = HOLD_B_;
  static constexpr int RM = BM / WM, RN = BN / WN, MT = RM / 16, NT = RN / 16;        // per-wave tile, in rows / 16x16 tiles
  static constexpr int HT = HOLD_B ? NT : MT, ST = HOLD_B ? MT : NT, NPH = ST / SQ;   // held / streamed tiles, phases per K-tile
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int NP = (BM + BN) / 8, NPW = NP / FW;                              // 1-KiB pieces per K-tile, per fetching wave
  static constexpr int HP = (HOLD_B ? BN : BM) / 8, HPW = HP / FW;                     // pieces of the held operand
  static constexpr int SW = HOLD_B ? WM : WN, SR = HOLD_B ? RM : RN;                   // wave slices of the streamed operand
  static constexpr int GP = SW * SQ * 2, GPW = GP / FW;                                // pieces of one streamed group
  static_assert(WM * WN == 8, "8 waves");
  static_assert(RM % 16 == 0 && RN % 16 == 0 && ST % SQ == 0, "wave tile");
  static_assert(NP % FW == 0 && HP % FW == 0 && GP % FW == 0, "pieces must split evenly over the fetching waves");
  static_assert(HPW + NPH * GPW == NPW, "piece census");
  static constexpr int LDS_CAP = 160 * 1024 / WGS;
  static_assert(NS * STAGE <= LDS_CAP, "LDS");
  // local issue slot of a wave's s-th piece of a tile, the phase that first reads it, the last piece phase q needs
  static constexpr int cum(int q) { return (q * NPW + NPH - 1) / NPH; }
  static constexpr int qissue(int s) { int q = 0; while (cum(q + 1) <= s) ++q; return q; }
  static constexpr int need_q(int s) { return s < HPW ? 0 : (s - HPW) / GPW; }
  static constexpr int nmax(int q) { return HPW + (q + 1) * GPW - 1; }
  static constexpr bool valid() {
    for (int s = 0; s < NPW; ++s) {
      // RAW: issued no later than the memory segment of the phase before its reader (MFMA-segment issue: one phase earlier)
      if (qissue(s) - LEAD > need_q(s) - 1 - (ISSUE_C ? 1 : 0)) return false;
      // WAR: >= 2 phases after the last reader of the region (MFMA-segment issue sits half a phase later: >= 1 phase)
      if (qissue(s) - LEAD < need_q(s) - NS * NPH + 2 - (ISSUE_C ? 1 : 0)) return false;
    }
    return true;
  }
  static_assert(valid(), "fetch schedule violates the RAW / WAR phase rules");
  // steady-state wait of phase q (after its own issues): everything phase q+1 reads has landed
  static constexpr int vm(int q) {
    const int q1 = (q + 1) % NPH, dt1 = (q + 1) / NPH;
    const int slot = q + LEAD - (ISSUE_C ? 1 : 0);          // last issue slot completed when phase q waits
    const int dt2 = slot / NPH, q2 = slot % NPH;
    return dt2 * NPW + cum(q2 + 1) - 1 - (dt1 * NPW + nmax(q1));
  }
  // persistent kernel: does the wait of absolute phase phi (counted from the tile's first phase) cover prologue fetches only, i.e.
  // was the last piece phase phi + 1 needs issued before the tile's first phase?  (Then the previous tile's epilogue stores,
  // issued right behind the prologue, are YOUNGER than everything awaited and may stay in flight.)
  static constexpr bool wait_is_prologue_only(int phi) {
    const int p1 = phi + 1, dt1 = p1 / NPH, q1 = p1 % NPH;
    return dt1 * NPH + qissue(nmax(q1)) - LEAD < 0;
  }
  // prologue: every (tile, piece) whose issue phase is negative
  static constexpr int PRO_TILES = (LEAD + NPH - 1) / NPH;
  static constexpr bool in_prologue(int dt, int s) { return dt * NPH + qissue(s) - LEAD < 0; }
  static constexpr int pro_count() {
    int n = 0;
    for (int dt = 0; dt < PRO_TILES; ++dt) for (int s = 0; s < NPW; ++s) if (in_prologue(dt, s)) ++n;
    return n;
  }
  static constexpr int VM_PRO = pro_count() - 1 - nmax(0);
  static_assert(VM_PRO >= 0, "prologue");
  // epilogue: per-wave fp32 transpose tile of ER rows x (RN + 4) floats inside the ring
  static constexpr int RS = RN + 4;
  static constexpr int er_fit() { int er = RM; while (er > 16 && 8 * er * RS * 4 > LDS_CAP) er >>= 1; return er; }
  static constexpr int ER = er_fit();
  static_assert((ER * RN / 8) % 64 == 0, "epilogue items per wave");
  static constexpr int LDS = (NS * STAGE > 8 * ER * RS * 4) ? NS * STAGE : 8 * ER * RS * 4;
};

}  // namespace rl
