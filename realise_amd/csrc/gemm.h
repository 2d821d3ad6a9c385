// GEMM launchers shared by the engine and the C-ABI (gemm.hip implements them).
#pragma once
#include "common.h"

namespace rl {

// ---- epilogue of the NT GEMM (runtime-selected; branches are block-uniform) ------------------
enum EpiMode {
  EPI_STORE = 0,       // out = alpha*acc (+bias) (+= if accumulate)
  EPI_GELU = 1,        // out2 = pre = acc+bias ; out = gelu(pre)                (K5)
  EPI_DROP_RESID = 2,  // out = dropout(acc+bias) + aux                            (K4 minus LN)
  EPI_GELU_BWD = 4,    // out = acc * gelu'(aux)                                   (K14)
  EPI_AFFINE = 8,      // out = [relu](acc * col_scale[col] + bias[col] (+ aux))  (K9, evaluation: BatchNorm folded into the convolution)
};

template <typename T> struct EpiParams {
  int mode = EPI_STORE;
  T* out = nullptr;
  int64_t ldo = 0;
  T* out2 = nullptr;
  const float* bias = nullptr;
  const T* aux = nullptr;
  int64_t ldaux = 0;
  float alpha = 1.0f;
  int accumulate = 0;
  uint32_t drop_seed = 0, drop_thresh = 0;
  float drop_scale = 1.0f;
  // optional output-row map (stride-2 data gradients computed per input-pixel parity class): GEMM row r of the class is pixel
  // (n, 2*yy + py, 2*xx + px) of an [N][2^(hw-w)][2^w] map, r = (n, yy, xx) over the half-resolution grid; rm_hw_shift < 0: identity
  int rm_hw_shift = -1, rm_w_shift = 0, rm_par = 0;
  int wide = 0;                 // set by the launcher: rows are 8-element aligned -> LDS-staged epilogue, 16 B per lane
  int probe = 0;                // diagnostics (tools/nt_probe.cpp): 1 all tiles fetch tile 0, 2 no fetches, 3 no MFMA
  // optional device-side count of the rows that matter (the 8-wave kernel skips tiles that start at or beyond it; rows of a partly
  // live tile beyond the count are still computed and stored: for outputs whose dead rows nobody reads - the compacted classifier
  // data gradient, engine.hip stage_head)
  const int* m_dev = nullptr;
  // m_exact != 0 (with m_dev): the A rows at or beyond the count read as zeros (their fetches park out of range), so a partly live
  // tile stores bias-only rows / leaves accumulated rows unchanged - what a launch bounded by a device-side row count needs when its
  // output rows beyond the count ARE read later (the GRU's recurrent data gradient, accumulated into dh)
  int m_exact = 0;
  // split-K form of the 8-wave kernel (gemm_nt8_splitk): the launch covers ksplit K-ranges of every tile and each workgroup stores its
  // fp32 partial tile to slab[split][row][col] (row pitch N, plane pitch slab_stride floats) instead of running an epilogue; the
  // caller folds the planes in a fixed order
  float* slab = nullptr;
  int64_t slab_stride = 0;
  int ksplit = 1;
  // Live-row form of the 8-wave 128 x 192 kernel (gemm_nt8_live): the M dimension is a device-side LIST of 16-row blocks (block ids
  // ascending, *live_count of them: row_liveness's 16-row list of a padded batch).  Tile t works on the rows of blocks
  // live_list[8t .. 8t + 7]: A rows are fetched from, and output rows (aux / out / out2, the dropout hash index) addressed at, their
  // ORIGINAL positions; rows of unlisted blocks are neither read nor written, tiles at or beyond the count leave at once.  A listed
  // row's result is bit-identical to the dense launch's (same kernel, same K order).
  const int* live_list = nullptr;
  const int* live_count = nullptr;
  // rows per list entry: 16 (blocks, above) or 1 (round 6, row-granular packing: live_list holds ROW ids, ascending, *live_count of
  // them; tile t works on rows live_list[128t .. 128t + 127] - no tile row is spent on the padding rows that fill a sentence's last
  // 16-row block: 0.648 instead of 0.709 of the rows of a SIGHAN-shaped batch, and the 2304-wide qkv launch fits ONE round of the 512
  // two-per-CU slots.  Per listed row the result is the same bits as the dense launch's.)
  int live_unit = 16;
  // live-row form: column groups of the XCD split (1, 2, 4 or 8; tiles_n % xcd_gc == 0).  XCD x owns the tile rows of row group
  // x / xcd_gc (of 8 / xcd_gc balanced groups over the LIVE tile rows) and the tile columns of column group x % xcd_gc: its weight
  // slice (N / xcd_gc rows of B) stays in its 4 MiB L2 while the A rows stream through, read by xcd_gc XCDs instead of the whole
  // weight panel being streamed through all eight (gemm_nt8_live picks it from the shape; realise_set_nt8p(3, v) overrides)
  int xcd_gc = 1;
  // EPI_AFFINE (round 6; src/char_cnn.py:15-32 in evaluation mode, where BatchNorm2d is the per-channel affine map scale = gamma /
  // sqrt(running_var + eps), shift = beta - running_mean * scale): the convolution's epilogue applies it to the fp32 accumulators,
  // adds the (already normalised) shortcut `aux` when given and clamps at zero when `relu` - no BatchNorm launch, no pass over the raw
  // convolution output.  col_scale is valid with EPI_AFFINE only; `bias` is the shift.  4-wave kernels and conv_c64_nt.
  const float* col_scale = nullptr;
  int relu = 0;
  // round 6: optional SECOND output of a plain store, the same values widened to fp32 ([M][ldo_f32] floats) - the evaluation logits the
  // reference returns in fp32 (src/models.py:859) leave the classifier kernel itself instead of a cast pass over [B*S, V] (persistent
  // 256 x 192 kernel only: gemm_nt8p; every other launch form refuses it)
  float* out_f32 = nullptr;
  int64_t ldo_f32 = 0;
  int l2_prefetch = 0;          // set by the launcher (realise_set_nt8p key 8, probe): surplus workgroups of a narrow row-list launch prefetch their XCD's operand lines into the L2
  int cu_pair = 0;              // set by the launcher (realise_set_nt8p key 7): tile order that puts two column tiles of a tile row on one CU (common.h cu_pair_local)
  int bias_first = 0;           // set by the launcher (realise_set_nt8p key 4): alpha / bias go into the accumulators before the epilogue's LDS transposes
  // K4 (BertSelfOutput / BertOutput, modeling_bert.py:273-277, 339-343): EPI_DROP_RESID followed by the LayerNorm of the row in the SAME
  // launch (8-wave 128 x 192 kernel, M % 128 == 0, N % 192 == 0, N / 192 <= 8).  A row spans N / 192 column tiles = workgroups: each
  // leaves (sum, M2) of its 192 columns in ln_part[row][tile] as two self-validating 64-bit words {value, tag = ln_target} and polls
  // the other tiles' slots until they carry this launch's tag (the caller passes a tag that differs from the previous launch's on the
  // same buffer, never 0); then every workgroup combines the partials (Chan), normalises its own columns from LDS and writes
  // xhat -> out, y -> ln_y, rstd -> ln_rstd.
  const float* ln_gamma = nullptr;
  const float* ln_beta = nullptr;
  float ln_eps = 0.f;
  T* ln_y = nullptr;
  float* ln_rstd = nullptr;
  float* ln_part = nullptr;       // [M][N / 192][2] 64-bit words (16 B per row and tile), zero-filled once
  int* ln_flag = nullptr;         // (unused)
  int ln_target = 0;              // launch tag
  int* ln_timeout = nullptr;      // nullable: set to 1 if a wait gave up (a workgroup of the band never arrived)
  // K6 (models.py:818-826, one time step of nn.GRU): the recurrent projection gh = h_prev . W_hh^T + b_hh with the gate math in its
  // epilogue (8-wave 128 x 192 kernel, N = 3H).  The B rows of a column tile are gathered gate-interleaved - [r | z | n] of 32 hidden
  // units per wave - so that a wave's staged tile holds all three gates of its units: r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z +
  // gh_z), n = tanh(gi_n + r gh_n), h = (1 - z) n + z h_prev, gi from the [33][3H] input table.  Stores h, (r, z, n) and gh_n (what
  // the backward step reads), and the sequence's output row when it ends at this step.
  const float* gru_table = nullptr;     // != nullptr selects the epilogue; out = h_new [M][H], bias = b_hh [3H]
  const int64_t* gru_pho_idx = nullptr;
  const int* gru_perm = nullptr;
  const int* gru_lens = nullptr;
  const T* gru_hprev = nullptr;
  T* gru_rzn = nullptr;                 // [M][3H]
  T* gru_gh = nullptr;                  // [M][3H]: only the n third is written
  T* gru_out = nullptr;                 // [N tokens][H], original order
  int gru_Tp = 0, gru_t = 0;
  // Stream-K form (gemm_nt8s): the exchange buffer of the launch (NT8S_PART_BYTES, private to the stream while the launch runs), one
  // flag per workgroup (NT8S_GRID x NT8S_FLAG_STRIDE ints, zero-filled once), the launch tag (never 0, different from the previous launch's on the same
  // buffers) and an optional word that is set to 1 if a finisher gave up waiting for a partial
  float* sk_part = nullptr;
  int* sk_flag = nullptr;
  int sk_tag = 0;
  int* sk_timeout = nullptr;
};
constexpr int NT8S_GRID = 256;                                    // one workgroup per CU
constexpr int NT8S_FLAG_STRIDE = 64;                              // ints between two workgroups' flags (256 B: different memory channels)
constexpr int64_t NT8S_PART_BYTES = (int64_t)NT8S_GRID * 24 * 512 * 16;      // 256 x 192 fp32 per workgroup

// ---- operand loaders ---------------------------------------------------------------------------
// A loader maps (row, k) of a logical K-contiguous operand to the global address of a 16-byte chunk
// (or to a zero page when out of range), for direct-to-LDS staging.  Ctx = per-row state, KPos =
// per-column state that is advanced incrementally from K-tile to K-tile (no integer divisions in
// the main loop).
template <typename T> struct DenseLoader {
  const T* base;
  int64_t ld;
  int rows, K;
  const int* rows_dev = nullptr;   // optional device-side row bound (pinyin GRU: #sequences still alive at this step)
  struct Ctx { const T* p; };
  struct KPos { int k; };
  __device__ __forceinline__ void clamp_rows() { if (rows_dev != nullptr) rows = min(rows, *rows_dev); }
  __device__ __forceinline__ Ctx prepare(int row) const {
    Ctx c; c.p = (row < rows) ? base + (int64_t)row * ld : nullptr; return c;
  }
  __device__ __forceinline__ KPos kpos(int k) const { KPos q; q.k = k; return q; }
  __device__ __forceinline__ void advance(KPos& q, int dk) const { q.k += dk; }
  __device__ __forceinline__ const void* addr(const Ctx& c, const KPos& q, const void* zero) const {
    return (c.p != nullptr && q.k < K) ? (const void*)(c.p + q.k) : zero;
  }
};

// Implicit im2col over an NHWC tensor (K7/K8).  Row space = pixels (n, y, x) of an Hr x Wr map,
// K = (kh, kw, c).  mode 0: forward conv gather, source pixel (y*stride + kh - pad, ...);
// mode 1: data-gradient gather, source pixel ((y + pad - kh)/stride, ...) when divisible.
// img_index (optional) redirects image n to table row img_index[n] (glyph lookup by token id).
template <typename T> struct ConvLoader {
  const T* src;
  const int64_t* img_index;
  int rows, Hr, Wr, Hs, Ws, C, KH, KW, stride, pad, mode, K;
  int hw_shift, w_shift;      // log2(Hr*Wr), log2(Wr) when both are powers of two, else -1 (set by finalize())
  const int* rows_dev = nullptr;   // optional device-side row bound (glyph dedup: #distinct ids * pixels per image)
  int64_t index_rows = 0;          // with img_index: number of images in `src` (0 = unknown: the caller vouches for < 4 GiB)
  // the gathered source must span less than 4 GiB: the fetches carry 32-bit buffer offsets
  bool span_ok() const {
    const int64_t per = (int64_t)Hs * Ws * C * (int64_t)sizeof(T);
    int64_t images = index_rows;
    if (img_index == nullptr) {        // rows = images x pixels of the Hr x Wr map (a quarter of them per parity class)
      const int64_t hw = (int64_t)Hr * Wr;
      images = hw > 0 ? ((int64_t)rows * (par >= 0 ? 4 : 1) + hw - 1) / hw : 0;
    }
    return images * per < 0xFFFFFE00ll;
  }
  // Parity-class form of a stride-2 data gradient (mode 1, 32-bit addressing only): par = 2*py + px in 0..3 restricts the rows to the
  // input pixels (2*yy + py, 2*xx + px) - rows = N * Hr/2 * Wr/2 - and K to the taps that can reach such a pixel,
  // kh = kh0 + 2i < KH with kh0 = (py + pad) & 1 (3x3 pad 1: one tap for an even coordinate, two for an odd one), so no fetched tap
  // is a stride miss: 1 + 2 + 2 + 4 = 9 tap-GEMMs over a quarter of the rows each instead of 9 over all rows (4x less work).
  // The B operand must hold the taps of a class contiguously (conv_s2_class()).  Hr, Wr powers of two.
  int par = -1;
  int tap_sel = -1;                                // class reduced to ONE tap kh * KW + kw (a 2x2 map under a 1x1 output: pixel (y, x) is
                                                   // reached by tap (y + pad, x + pad) only); the B operand then points at that tap's slot
  int kh0 = 0, kw0 = 0, nkw = 1, kstep = 1;        // set by finalize()
  struct Ctx { const T* img; int y, x; };
  struct KPos { int kh, kw, ch, k; };
  __device__ __forceinline__ void clamp_rows() { if (rows_dev != nullptr) rows = min(rows, par >= 0 ? (*rows_dev >> 2) : *rows_dev); }
  void finalize() {
    K = KH * KW * C;
    kh0 = kw0 = 0; nkw = KW; kstep = 1;
    if (par >= 0) {
      kh0 = ((par >> 1) + pad) & 1; kw0 = ((par & 1) + pad) & 1; kstep = 2;
      int nkh = kh0 < KH ? (KH - kh0 + 1) / 2 : 0;
      nkw = kw0 < KW ? (KW - kw0 + 1) / 2 : 0;
      if (tap_sel >= 0) { kh0 = tap_sel / KW; kw0 = tap_sel - kh0 * KW; nkh = 1; nkw = 1; }
      K = nkh * nkw * C;
    }
    hw_shift = w_shift = -1;
    const int hw = Hr * Wr;
    if (hw > 0 && (hw & (hw - 1)) == 0 && (Wr & (Wr - 1)) == 0) {
      hw_shift = 0; while ((1 << hw_shift) < hw) ++hw_shift;
      w_shift = 0; while ((1 << w_shift) < Wr) ++w_shift;
    }
  }
  __device__ __forceinline__ Ctx prepare(int row) const {
    Ctx c; c.img = nullptr; c.y = 0; c.x = 0;
    if (row < rows) {
      int n, rem;
      if (hw_shift >= 0) {
        n = row >> hw_shift; rem = row & ((1 << hw_shift) - 1);
        c.y = rem >> w_shift; c.x = rem & ((1 << w_shift) - 1);
      } else {
        const int hw = Hr * Wr;
        n = row / hw; rem = row - n * hw;
        c.y = rem / Wr; c.x = rem - c.y * Wr;
      }
      const int64_t ns = img_index ? img_index[n] : (int64_t)n;
      c.img = src + ns * Hs * Ws * C;
    }
    return c;
  }
  __device__ __forceinline__ KPos kpos(int k) const {
    KPos q; q.k = k;
    const int tap = k / C;
    q.ch = k - tap * C; q.kh = tap / KW; q.kw = tap - q.kh * KW;
    return q;
  }
  __device__ __forceinline__ void advance(KPos& q, int dk) const {
    q.k += dk; q.ch += dk;
    while (q.ch >= C) { q.ch -= C; if (++q.kw == KW) { q.kw = 0; ++q.kh; } }
  }
  // ---- 32-bit addressing for raw-buffer LDS-DMA fetches (one VGPR offset per fetch, the descriptor in SGPRs).  A row keeps the
  // byte offset of its ANCHOR pixel (the tap-(0,0) source pixel, possibly inside the zero padding; arithmetic is mod 2^32) and the
  // anchor's coordinates; a lane keeps its current tap (kh, kw, channel) and the tap's byte offset relative to the anchor.  Per
  // fetch: two adds, two unsigned compares, one select - instead of a 64-bit address recomputed from (n, y, x, kh, kw) each time.
  // A fetch that must read zeros (padding, row / K tails, stride-2 parity misses) gets an offset beyond num_records: the buffer
  // range check returns zeros.
  static constexpr uint32_t OOB = 0xFFFFFF00u;          // with num_records = RECORDS below
  static constexpr uint32_t RECORDS = 0xFFFFFE00u;
  struct Row32 { uint32_t off; int y0, x0; };
  struct Tap32 { int kh, kw, ch, k; uint32_t toff; };
  __device__ __forceinline__ Row32 prepare32(int row) const { return prepare32_t<true>(row); }
  // IDX = false: the caller knows img_index == nullptr (ConvLoaderDirect).  Not a micro-optimisation: the weight-gradient kernel calls
  // this per fetched piece INSIDE its K loop, and a conditional global load there - never executed, img_index being null in every engine
  // call - still made hipcc wait with vmcnt(0) at the merge behind it: every gathered fetch of every K-tile was issued alone, after the
  // previous one had landed (found in the ISA in round 5; the conv weight gradients ran at 0.13 of the MFMA peak).
  template <bool IDX> __device__ __forceinline__ Row32 prepare32_t(int row) const {
    Row32 r; r.off = 0u; r.y0 = -(1 << 24); r.x0 = -(1 << 24);          // out of range: every tap fails the bounds test
    if (row < rows) {
      int n, rem, y, x;
      if (par >= 0) {
        n = row >> (hw_shift - 2); rem = row & ((1 << (hw_shift - 2)) - 1);
        y = ((rem >> (w_shift - 1)) << 1) + (par >> 1); x = ((rem & ((1 << (w_shift - 1)) - 1)) << 1) + (par & 1);
      } else if (hw_shift >= 0) {
        n = row >> hw_shift; rem = row & ((1 << hw_shift) - 1);
        y = rem >> w_shift; x = rem & ((1 << w_shift) - 1);
      } else {
        const int hw = Hr * Wr;
        n = row / hw; rem = row - n * hw;
        y = rem / Wr; x = rem - y * Wr;
      }
      int64_t ns = (int64_t)n;
      if constexpr (IDX) { if (img_index) ns = img_index[n]; }
      const int64_t img = ns * Hs * Ws;
      if (mode == 0) { r.y0 = y * stride - pad; r.x0 = x * stride - pad; }
      else { r.y0 = y + pad; r.x0 = x + pad; }
      if (mode == 0 || stride == 1) r.off = (uint32_t)((img + (int64_t)r.y0 * Ws + r.x0) * C * (int64_t)sizeof(T));
      else r.off = (uint32_t)(img * C * (int64_t)sizeof(T));
    }
    return r;
  }
  __device__ __forceinline__ void tap_offset(Tap32& q) const {
    if (mode == 0) q.toff = (uint32_t)(((q.kh * Ws + q.kw) * C + q.ch) * (int)sizeof(T));
    else if (stride == 1) q.toff = (uint32_t)((q.ch - (q.kh * Ws + q.kw) * C) * (int)sizeof(T));
    else q.toff = (uint32_t)(q.ch * (int)sizeof(T));
  }
  __device__ __forceinline__ Tap32 tap32(int k) const {
    Tap32 q; q.k = k;
    const int tap = k / C;
    const int i = tap / nkw;
    q.ch = k - tap * C; q.kh = kh0 + i * kstep; q.kw = kw0 + (tap - i * nkw) * kstep;
    tap_offset(q);
    return q;
  }
  __device__ __forceinline__ void advance32(Tap32& q, int dk) const {
    q.k += dk; q.ch += dk;
    while (q.ch >= C) { q.ch -= C; q.kw += kstep; if (q.kw >= KW) { q.kw = kw0; q.kh += kstep; } }
    tap_offset(q);
  }
  __device__ __forceinline__ uint32_t voff32(const Row32& r, const Tap32& q) const {
    if (q.k >= K) return OOB;
    if (mode == 0) {
      const bool ok = (unsigned)(r.y0 + q.kh) < (unsigned)Hs && (unsigned)(r.x0 + q.kw) < (unsigned)Ws;
      return ok ? r.off + q.toff : OOB;
    }
    const int ty = r.y0 - q.kh, tx = r.x0 - q.kw;
    if (stride == 1) {
      const bool ok = (unsigned)ty < (unsigned)Hs && (unsigned)tx < (unsigned)Ws;
      return ok ? r.off + q.toff : OOB;
    }
    if (stride == 2) {
      const int sy = ty >> 1, sx = tx >> 1;
      const bool ok = ((ty | tx) >= 0) && (((ty | tx) & 1) == 0) && sy < Hs && sx < Ws;
      return ok ? r.off + (uint32_t)((sy * Ws + sx) * C * (int)sizeof(T)) + q.toff : OOB;
    }
    const int sy = ty / stride, sx = tx / stride;
    const bool ok = ty >= 0 && tx >= 0 && sy * stride == ty && sx * stride == tx && sy < Hs && sx < Ws;
    return ok ? r.off + (uint32_t)((sy * Ws + sx) * C * (int)sizeof(T)) + q.toff : OOB;
  }
  __device__ __forceinline__ const void* addr(const Ctx& c, const KPos& q, const void* zero) const {
    if (c.img == nullptr || q.k >= K) return zero;
    int sy, sx;
    if (mode == 0) {
      sy = c.y * stride + q.kh - pad; sx = c.x * stride + q.kw - pad;
    } else {
      const int ty = c.y + pad - q.kh, tx = c.x + pad - q.kw;
      if (ty < 0 || tx < 0) return zero;
      if (stride == 2) {
        if ((ty | tx) & 1) return zero;
        sy = ty >> 1; sx = tx >> 1;
      } else {
        sy = ty / stride; sx = tx / stride;
        if (sy * stride != ty || sx * stride != tx) return zero;
      }
    }
    if (sy < 0 || sy >= Hs || sx < 0 || sx >= Ws) return zero;
    return (const void*)(c.img + ((int64_t)(sy * Ws + sx) * C + q.ch));
  }
};

// the same geometry for callers that gather straight from `src` (img_index == nullptr): no index load anywhere in the device code
template <typename T> struct ConvLoaderDirect : ConvLoader<T> {
  ConvLoaderDirect() = default;
  explicit ConvLoaderDirect(const ConvLoader<T>& b) : ConvLoader<T>(b) {}
  __device__ __forceinline__ typename ConvLoader<T>::Row32 prepare32(int row) const { return this->template prepare32_t<false>(row); }
};

// Stride-2 data gradient by parity classes: class c = 2*py + px of a KH x KW / pad convolution uses the taps (kh0 + 2i, kw0 + 2j);
// the dgrad weight copy stores the taps class by class (class 0's taps, then class 1's, ...), each class i-major.  Returns the
// number of taps of class c, its first slot in that order, and (optionally) the original tap index of every slot.
inline int conv_s2_class(int KH, int KW, int pad, int c, int* first_slot, int* order /* nullable, KH*KW entries */) {
  int slot = 0, count = 0;
  for (int cc = 0; cc < 4; ++cc) {
    const int kh0 = ((cc >> 1) + pad) & 1, kw0 = ((cc & 1) + pad) & 1;
    if (cc == c) { *first_slot = slot; }
    int n = 0;
    for (int kh = kh0; kh < KH; kh += 2)
      for (int kw = kw0; kw < KW; kw += 2) { if (order) order[slot + n] = kh * KW + kw; ++n; }
    if (cc == c) count = n;
    slot += n;
  }
  return count;
}

// slot of original tap (kh, kw) in the class-ordered copy
inline int conv_s2_slot(int KH, int KW, int pad, int kh, int kw) {
  int order[64];
  int first = 0;
  (void)conv_s2_class(KH, KW, pad, 0, &first, order);
  for (int s = 0; s < KH * KW; ++s) if (order[s] == kh * KW + kw) return s;
  return -1;
}

// ---- TN (weight-gradient) epilogue ---------------------------------------------------------------
enum TnMode { TN_PLAIN = 0, TN_CONVW = 1 };
// LDS a TN kernel keeps its live-block list in, sized per launch from the number of list entries (a fixed 4 KB refused
// batches beyond 16384 token rows): 4 KB (1024 entries = 16384 token rows in 16-row blocks, the bench shape: two 64 KB workgroups per
// CU as before) up to TN_LIST_LDS_MAX = 16 KB (4096 entries = 65536 token rows; 2 x (64 + 16) KB still fit the 160 KB of a CU).
// Longer lists are refused (RL_ERR_ARG); the engine does not build liveness tables for such batches (dense reductions).
constexpr int TN_LIST_LDS = 4096;
constexpr int TN_LIST_LDS_MAX = 16384;
constexpr int TN_LIST_MAX_ENTRIES = TN_LIST_LDS_MAX / 4;
inline int tn_list_lds_bytes(int64_t entries) {
  if (entries > TN_LIST_MAX_ENTRIES) return -1;
  const int64_t b = (entries * 4 + 1023) & ~(int64_t)1023;
  return (int)(b < TN_LIST_LDS ? TN_LIST_LDS : b);
}
struct TnEpi {
  int mode = TN_PLAIN;
  float* out = nullptr;     // fp32, accumulated (out += result)
  int64_t ldo = 0;
  float alpha = 1.0f;
  const float* alpha_dev = nullptr; // optional device scalar multiplied on top of alpha (the incoming d loss of the classifier's gradients: never read on the host)
  int Cin = 0, Cpad = 0, KHW = 0;   // TN_CONVW: j = tap*Cpad + ci -> out[(i*Cin + ci)*KHW + tap + tap0]
  int tap0 = 0;                     // first tap the J columns cover (a 3x3 convolution on a 1x1 map only has its centre tap: J = Cpad, tap0 = 4)
  // split-reduction scratch: when the reduction is split over several workgroups each writes a dense fp32
  // [I][J] partial slab here and a second kernel folds the slabs into `out` (no atomics).  nullptr -> atomics.
  float* slab = nullptr;
  int64_t slab_elems = 0;
  // optional fused column sums of the A operand: colsum[i] += alpha * sum_p A[p,i]  (the bias gradient that goes
  // with a Linear weight gradient), computed by one extra ones-vector MFMA per A fragment in the j-tile-0 workgroups
  float* colsum = nullptr;
  int overwrite = 0;        // TN_OUT_DIRECT only: out = result instead of out += result (the caller knows `out` holds nothing yet:
                            // no read of the old value, no zero-fill before the pass)
  int probe = 0;            // diagnostics (tools/nt_probe.cpp): 2 no fetches, 3 no MFMA, 4 no fold pass
  // Optional list of the LIVE reduction tiles (unsplit dense reductions only): tile_list[t] = index of the t-th BP-row block of the
  // reduction rows that holds anything but exact zeros in A (BP = 64 bf16 / 32 fp32 rows, P % BP == 0), *n_tiles = how many.  The
  // rows of the other blocks are skipped - the gradient rows of padding tokens are exact zeros (engine.hip row_liveness).
  const int* tile_list = nullptr;
  const int* n_tiles = nullptr;
  // rows per list entry: BP (whole reduction tiles), or 16 (bf16, grouped 128 x 128 kernel): entries name live 16-row blocks and a
  // reduction tile is any four of them (wave w of the workgroup fetches the w-th) - the last tile is padded with zero blocks
  int list_rows = 0;
};

// Grouped TN (weight-gradient) launch: problems sharing the reduction length P, one 128x128 tile per workgroup, no reduction split.
constexpr int TN_GROUP_MAX = 4;
template <typename T> struct TnGroupProblem {
  const T* A = nullptr; int64_t lda = 0;        // dY [P, I]
  const T* B = nullptr; int64_t ldb = 0;        // X  [P, J]
  int I = 0, J = 0;
  float* out = nullptr; int64_t ldo = 0;        // fp32 [I, ldo], accumulated; ldo % 4 == 0
  float* colsum = nullptr;                      // nullable: += column sums of A (bias gradient)
  int tiles_j = 0, ntiles = 0, tile_begin = 0;  // filled by gemm_tn_group
  int jmajor = 0;                               // filled by gemm_tn_group: tiles walked j-panel by j-panel (the B operand is the larger one)
};
template <typename T> struct TnGroup {
  TnGroupProblem<T> p[TN_GROUP_MAX];
  int n = 0, total_tiles = 0, probe = 0, overwrite = 0;
  float alpha = 1.0f;
  const int* tile_list = nullptr;      // live reduction tiles (see TnEpi)
  const int* n_tiles = nullptr;
  int list_rows = 0;
};
template <typename T>
int gemm_tn_group(hipStream_t st, int n, const TnGroupProblem<T>* probs, int P, float alpha = 1.0f, int overwrite = 0,
                  const int* tile_list = nullptr, const int* n_tiles = nullptr, int list_rows = 0);

// K4: GEMM + bias + dropout + residual + LayerNorm in one launch (gemm_nt8.hip, see EpiParams::ln_*); RL_ERR_ARG = shape not supported
int gemm_nt8_ln(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, const EpiParams<bf16_t>& ep);
// K6: one GRU time step = recurrent GEMM + gate math in one launch (EpiParams::gru_*); H = N / 3 must be a multiple of 64
int gemm_nt8_gru(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, const EpiParams<bf16_t>& ep);
// Split-K NT GEMM (bf16, K % 64 == 0, nsplit <= K / 64): slab[s][m][n] (fp32, row pitch N, plane pitch slab_stride) = A[m, Ks] . B[n, Ks]^T
// over the s-th K-range; rows at or beyond *m_dev (nullable) are not computed.  For reductions long enough that the output tiles
// alone do not fill the chip: the classifier's data gradient (K = 21184, N = 768, ~4.9 k live rows = 156 tiles of 128 x 192).
int gemm_nt8_live(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, const EpiParams<bf16_t>& ep);
int gemm_nt8_splitk(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, int nsplit,
                    float* slab, int64_t slab_stride, const int* m_dev);

// C[M,N] = A[M,K] . B[N,K]^T   (both operands K-contiguous)
template <typename T>
int gemm_nt(hipStream_t st, const T* A, int64_t lda, const T* B, int64_t ldb, int M, int N, int K,
            const EpiParams<T>& ep, const int* rows_dev = nullptr);
template <typename T>
int gemm_nt_conv(hipStream_t st, const ConvLoader<T>& la, const T* B, int64_t ldb, int M, int N, int K,
                 const EpiParams<T>& ep);
// C[I,J] += sum_p A[p,i] * B[p,j]   (A: [P, >=I] row-major; B dense or gathered)
template <typename T>
int gemm_tn(hipStream_t st, const T* A, int64_t lda, const T* B, int64_t ldb, int P, int I, int J, const TnEpi& ep,
            const int* rows_dev = nullptr);
template <typename T>
int gemm_tn_conv(hipStream_t st, const T* A, int64_t lda, const ConvLoader<T>& lb, int P, int I, int J, const TnEpi& ep);

// Ping-pong 8-wave kernel (gemm_nt8.hip): bf16, dense operands, K % 64 == 0.  tile: 0 heuristic, 1 256x256, 2 256x192,
// 3 256x128, 4 128x192.
bool nt8_supported(int M, int N, int K, const EpiParams<bf16_t>& ep, int64_t lda, int64_t ldb);
// persistent 256 x 192 kernel with the LDS-free epilogue (gemm_nt8p.hip): EPI_STORE / EPI_GELU / EPI_GELU_BWD, K % 64 == 0
bool nt8p_supported(int M, int N, int K, const EpiParams<bf16_t>& ep, int64_t lda, int64_t ldb);
int gemm_nt8p(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, const EpiParams<bf16_t>& ep);
// stream-K persistent 256 x 192 kernel (gemm_nt8s.hip): one round of NT8S_GRID workgroups over equal K-tile ranges, partial tiles folded
// in-kernel in a fixed order; optional live-block list; EPI_STORE (+ accumulate) / EPI_GELU / EPI_GELU_BWD / EPI_DROP_RESID, K % 128 == 0
bool nt8s_supported(int M, int N, int K, const EpiParams<bf16_t>& ep, int64_t lda, int64_t ldb);
int gemm_nt8s(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K, const EpiParams<bf16_t>& ep);
void set_nt8p_wgs(int n);
void set_nt8_epi_pre(int on);      // 1 (default): alpha / bias into the accumulators before the 8-wave kernels' epilogue transposes (one bias fetch per wave, no per-item waits)
void set_nt8_l2_prefetch(int v);  // realise_set_nt8p key 8 (probe): see gemm_nt8.hip nt8_l2_prefetch
void set_nt8_cu_pair(int on);    // realise_set_nt8p key 7 (probe): two-per-CU 8-wave kernels pair the column tiles of a tile row on one CU
void set_tn_jmajor(int on);       // realise_set_nt8p key 6: grouped weight gradients walk a problem's tiles along its LARGER operand's panels (1; default 0: measured level)
void set_nt8_live_big(int v);     // measurement knob (realise_set_nt8p key 5): wide row-list launches on 256 x 256 one-per-CU tiles
void set_nt8_live_gc(int gc);      // live-row GEMMs: 0 (default) column groups of the XCD split from the shape, 1 / 2 / 4 / 8 forced
void set_nt8p_order(int o);
void set_nt8_single_round(int on);   // outputs of at most one 128 x 192 tile per CU: 1 the three-stage one-per-CU shape, 0 (default) the two-per-CU shape
int gemm_nt8(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int M, int N, int K,
             const EpiParams<bf16_t>& ep, int tile);
void set_nt8_probe(int mode);
void set_nt8_group_m(int g);       // tile order of the 8-wave NT kernels: 0/1 row-major, g > 1: g tile rows per column step (L2 blocking)             // 2 no fetches, 3 no MFMA / fragment reads (results wrong)

// Ping-pong 8-wave weight-gradient kernel (gemm_tn8.hip): bf16, dense operands, 256 x 128 output tiles, split reduction + fold
bool tn8_supported(int64_t lda, int64_t ldb, int P, int I, int J, const TnEpi& ep);
// grouped 8-wave form (gemm_tn8.hip): 256 x 128 tiles, one per CU, unsplit, optional list of live 16-row blocks; RL_ERR_ARG = not applicable
int gemm_tn8_group(hipStream_t st, int n, const TnGroupProblem<bf16_t>* probs, int P, float alpha, int overwrite, const int* tile_list,
                   const int* n_tiles, int list_rows);
void set_tn_group8(int on);      // transformer-layer weight gradients through gemm_tn8_group (default 0: measured slower)
int gemm_tn8(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int P, int I, int J, const TnEpi& ep, int force_split);
void tn_fold_launch(hipStream_t st, const TnEpi& ep, int nsplit, int I, int J);     // out += alpha * sum of the split slabs, fixed order
void set_tn_variant(int v);
void set_conv_c64(int on);          // 1 (default): 64-channel 3x3 s1 conv forward / dgrad / wgrad on 16x16 maps via the LDS-resident kernels
int conv_c64_nt(hipStream_t st, const bf16_t* X, const bf16_t* Wt, bf16_t* out, int rows, const int* rows_dev, int flip,
                const float* col_scale = nullptr, const float* col_shift = nullptr, const bf16_t* aux = nullptr, int relu = 0);         // conv_c64_nt.hip
int conv_wgrad_c64(hipStream_t st, const bf16_t* dY, const bf16_t* X, int rows, const int* rows_dev, const TnEpi& te);   // conv_wgrad_c64.hip
void set_tn_group_ring(int on);     // grouped TN: 0 two full stages (default), 1 four stages of half-height K-tiles (measured 11 % slower)               // 0 production, 9 force the 4-wave TN kernel

void set_tn_transpose_read(int use_tr);
void set_nt_wide_epilogue(int on);        // A/B knob: LDS-staged 16-B-per-lane epilogue (default on)
void set_tn_probe(int mode);
void set_tn_split(int n);                 // force the reduction split of the TN kernel (0 = heuristic)
void set_nt_variant(int v);               // experimental NT tile shapes, 0 = production heuristic
void set_nt_probe(int mode);              // bottleneck probe of the NT kernel, 0 = off (results are wrong when on)
void set_nt_allow_n96(int on);            // allow the 128x96 NT tile (chip-balance heuristic), default on

}  // namespace rl
