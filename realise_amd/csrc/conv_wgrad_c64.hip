// Weight gradient of the 64 -> 64 channel 3x3 / stride 1 / pad 1 convolution on 16x16 maps (glyph ResNet block 1, residual_function.3,
// src/char_cnn.py:19 under loss.backward()) with an LDS-RESIDENT INPUT TILE (bf16):
//
//   dW[co][tap][ci] = sum over pixels p of  dY[p][co] * X[p + (kh-1, kw-1)][ci]
//
// The generic TN kernel (gemm.hip) treats this as C[64, 576] = dY^T . im2col(X): three 64x256 column tiles, each streaming dY again and
// fetching every input pixel once per tap - 1.8 GB of LDS fills and ~3x the operand bytes from HBM per call, 439 us on the 957k rows of
// the dedup'd B=64 batch (profiles/round2_conv_tn_probe.log).  Here one workgroup owns ALL 576 output columns:
//   * a reduction tile is 64 consecutive pixels = 4 image rows; the A operand is their dY rows (8 KB), the B operand the 6 input rows
//     around them (4 + a halo row above and below, 12 KB; rows outside the image are zero-filled by out-of-range buffer offsets);
//   * the nine taps read the SAME 12 KB through row-shifted transposed LDS reads (ds_read_b64_tr_b16 takes one row address per lane,
//     so a tap is an address offset of (kh-1)*16 + (kw-1) pixel rows); the x-wrap of a horizontal shift - pixel x = 0 under kw = 0,
//     x = 15 under kw = 2 - is one register element of fixed position per lane class, zeroed with one select;
//   * each input pixel and each dY row is fetched from HBM once: 20 KB of LDS fill per 64 pixels instead of 120 KB.
// Wave w multiplies the 16 input channels [16w, 16w+16) of all nine taps against the four 16-channel blocks of dY: 36 accumulator
// tiles (144 VGPRs).  Three LDS stages with counted vmcnt, two workgroups per CU.  The reduction is split over workgroups by pixel
// range; each writes its [64][576] partial to a slab and tn_fold_launch (gemm.hip) adds them in a fixed order into the reference's
// [Co][Ci][3][3] layout.
#include "gemm_dev.h"
#include "prof.h"

namespace rl {

namespace {
constexpr int CW_C = 64, CW_HW = 16, CW_PIX = 256;             // channels, map edge, pixels per image
constexpr int CW_A_BYTES = 64 * 128, CW_B_ROWS = 96, CW_B_BYTES = CW_B_ROWS * 128, CW_STAGE = CW_A_BYTES + CW_B_BYTES;
constexpr int CW_NST = 3;
constexpr uint32_t CW_OOB = 0xFFFFFF00u, CW_RECORDS = 0xFFFFFE00u;

// B fragment of tap (kh, KW_): 8 consecutive pixels p (k-group g of K-step ks) of input channel c0 + l15, read from the halo tile at
// row p + 16 + shift, shift = (kh - 1) * 16 + (KW_ - 1).
template <int KW_>
__device__ __forceinline__ bf16x8_t cw_bfrag(const char* Bt, int ks, int c0, int l15, int g, int shift) {
  typedef short4_t __attribute__((address_space(3))) * lds_s4;
  short4_t h[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int q = ks * 32 + 8 * g + 4 * hh + (l15 >> 2) + 16 + shift;
    const int colb = ((c0 + 4 * (l15 & 3)) * 2) ^ tn_swz<bf16_t, 128>(q);
    h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(Bt + q * 128 + colb));
  }
  typedef __attribute__((ext_vector_type(8))) short short8_t;
  short8_t r = {h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
  // element e of half hh is pixel x = 8 * (g & 1) + 4 * hh + e of its image row: x - 1 < 0 only for (g even, hh 0, e 0), x + 1 > 15 only
  // for (g odd, hh 1, e 3) - the neighbouring row's pixel sits there in the flat tile
  if (KW_ == 0) r[0] = (g & 1) ? r[0] : (short)0;
  if (KW_ == 2) r[7] = (g & 1) ? (short)0 : r[7];
  return __builtin_bit_cast(bf16x8_t, r);
}
}  // namespace

__global__ void __launch_bounds__(256, 2)
conv_wgrad_c64_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X, int rows_max, const int* __restrict__ rows_dev, int nsplit,
                      float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rows = rows_dev != nullptr ? min(rows_max, *rows_dev) : rows_max;
  const int ntiles = rows >> 6;                                  // rows is a multiple of 256 (whole images)
  const int split = blockIdx.x;
  const int chunk = (ntiles + nsplit - 1) / nsplit;
  const int t0 = min(ntiles, split * chunk), t1 = min(ntiles, t0 + chunk);
  const int nt = t1 - t0;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)CW_RECORDS, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)CW_RECORDS, 0x00020000);
  // per-lane source offsets inside a tile (LDS image: row r at r * 128, 16-byte chunk c of row r holds source chunk c ^ swizzle(r))
  const int lrow = lane >> 3, lchunk = (lane & 7) * 16;
  uint32_t aoff[2], boff[3];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 8 + lrow;                    // dY row of the tile, 0..63
    aoff[j] = (uint32_t)(r * 128 + (lchunk ^ tn_swz<bf16_t, 128>(r)));
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int q = (wave * 3 + j) * 8 + lrow;                    // halo-tile row 0..95 <-> image pixel quarter*64 - 16 + q
    boff[j] = (uint32_t)(q * 128 + (lchunk ^ tn_swz<bf16_t, 128>(q)));
  }
  auto issue = [&](int t, int stage) {
    char* base = smem + stage * CW_STAGE;
    const int quarter = t & 3;
    const uint32_t abase = (uint32_t)t * (uint32_t)CW_A_BYTES;
    const uint32_t bbase = (uint32_t)t * (uint32_t)CW_A_BYTES - 16u * 128u;           // pixel t*64 - 16 (never used when it would be negative)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + (wave * 2 + j) * 1024), 16, abase + aoff[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int piece = wave * 3 + j;                           // 8 halo rows; an image row is two pieces
      const bool valid = !(quarter == 0 && piece < 2) && !(quarter == 3 && piece >= 10);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(base + CW_A_BYTES + piece * 1024), 16,
                                               valid ? bbase + boff[j] : CW_OOB, 0, 0, 0);
    }
  };

  floatx4 acc[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  constexpr int NL = 5;
  int issued = 0;
  auto issue_next = [&]() {
    if (issued < nt) { issue(t0 + issued, issued % CW_NST); ++issued; }
  };
  issue_next();
  issue_next();
  const int c0 = wave * 16;
  for (int t = 0; t < nt; ++t) {
    const int younger = issued - 1 - t;
    if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_next();
    const char* At = smem + (t % CW_NST) * CW_STAGE;
    const char* Bt = At + CW_A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) a[f] = tn_frag_bf16<true, 128>(At, ks, f * 16, l15, g);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int sh = (kh - 1) * 16;
        bf16x8_t b0 = cw_bfrag<0>(Bt, ks, c0, l15, g, sh - 1);
        bf16x8_t b1 = cw_bfrag<1>(Bt, ks, c0, l15, g, sh);
        bf16x8_t b2 = cw_bfrag<2>(Bt, ks, c0, l15, g, sh + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i][kh * 3 + 0] = MmaBF16::mma(b0, a[i], acc[i][kh * 3 + 0]);
          acc[i][kh * 3 + 1] = MmaBF16::mma(b1, a[i], acc[i][kh * 3 + 1]);
          acc[i][kh * 3 + 2] = MmaBF16::mma(b2, a[i], acc[i][kh * 3 + 2]);
        }
      }
    }
  }
  // partial [64][576] of this split: row = co = i * 16 + l15, column = tap * 64 + c0 + 4 * g .. + 3
  float* out = slab + (int64_t)split * 64 * 576;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      *(floatx4*)(out + (i * 16 + l15) * 576 + tap * 64 + c0 + 4 * g) = acc[i][tap];
}

// dW (reference layout [64][64][3][3], fp32) += conv weight gradient; dY [rows][64], X [rows][64] NHWC over 16x16 maps (rows = images * 256,
// optionally bounded on the device).  te: TN_CONVW epilogue (Cin = Cpad = 64, KHW = 9) with its slab.
int conv_wgrad_c64(hipStream_t st, const bf16_t* dY, const bf16_t* X, int rows, const int* rows_dev, const TnEpi& te) {
  if (rows <= 0) return RL_OK;
  if ((rows % CW_PIX) || te.mode != TN_CONVW || te.Cin != 64 || te.Cpad != 64 || te.KHW != 9 || te.slab == nullptr) return RL_ERR_ARG;
  if ((int64_t)rows * 128 >= (int64_t)CW_RECORDS) return RL_ERR_ARG;                 // 32-bit buffer offsets
  int nsplit = (int)(te.slab_elems / (64 * 576));
  if (nsplit > 512) nsplit = 512;
  const int ntiles = rows / 64;
  if (nsplit > ntiles) nsplit = ntiles;
  if (nsplit < 1) return RL_ERR_ARG;
  const int lds = CW_NST * CW_STAGE + 256;        // + slack: the masked x-wrap elements of the last stage's last rows are read one row past the tile
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)conv_wgrad_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
  {
    ProfScope ps(st, PK_CONV_TN, 2.0 * rows * 64 * 576);
    RL_LAUNCH(conv_wgrad_c64_kernel, dim3(nsplit), dim3(256), lds, st, dY, X, rows, rows_dev, nsplit, te.slab);
    tn_fold_launch(st, te, nsplit, 64, 576);
  }
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}

}  // namespace rl
