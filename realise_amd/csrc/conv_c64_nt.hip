// Forward and data gradient of the 64 -> 64 channel 3x3 / stride 1 / pad 1 convolution on 16x16 maps (glyph ResNet block 1,
// residual_function.3, src/char_cnn.py:19) with an LDS-RESIDENT IMAGE AND WEIGHT MATRIX (bf16):
//
//   out[p][n] = sum over taps (kh, kw) and k of  X[p + s * ((kh-1) * 16 + (kw-1))][k] * W[n][tap][k]
//
// s = +1, W = the [co][tap][ci] operand copy: the convolution;  s = -1, W = the [ci][tap][co] copy: its input gradient.
// The generic implicit-GEMM kernel (gemm.hip) fetches every input pixel once per tap through the LDS-DMA path (9 x 128 B per output
// pixel) and re-reads the 73 KB weight matrix for every 128-pixel tile.  Here a workgroup keeps the weights in LDS for its whole
// life (64 rows x 1152 B, pitch 1168 B: the 16 rows of a B fragment land on 16 distinct 4-bank groups) and walks over images: one
// 256-pixel image (32 KB, double buffered) is fetched once, the nine taps are row-shifted ds_read_b128 fragment reads of it.  An
// image row is exactly one 16-row MFMA block (lane = x), so the x-wrap of a horizontal tap is lane 0 / lane 15 of the A fragment,
// and the rows above / below the image are 17 zero pixels kept on both sides of the LDS image.  HBM traffic: every input pixel and
// every output pixel once - a streaming kernel (algorithmic bytes 2 * P * 64 * 2 B).
#include "gemm_dev.h"
#include "prof.h"

namespace rl {

namespace {
constexpr int CN_WPITCH = 1168, CN_W_BYTES = 64 * CN_WPITCH;            // 74752
constexpr int CN_HALO = 17, CN_IMG_ROWS = CN_HALO + 256 + CN_HALO;      // 290 pixel rows of 128 B
constexpr int CN_STAGE = CN_IMG_ROWS * 128;                             // 37120
constexpr int CN_LDS = CN_W_BYTES + 2 * CN_STAGE;                       // 148992
constexpr uint32_t CN_RECORDS = 0xFFFFFE00u;
}  // namespace

template <int FLIP>
__global__ void __launch_bounds__(512, 1)
conv_c64_nt_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wt, bf16_t* __restrict__ out, int rows_max,
                   const int* __restrict__ rows_dev, const float* __restrict__ col_scale, const float* __restrict__ col_shift,
                   const bf16_t* __restrict__ aux, int relu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wl = smem;
  char* img0 = smem + CN_W_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rows = rows_dev != nullptr ? min(rows_max, *rows_dev) : rows_max;
  const int nimg = rows >> 8;

  // weights -> LDS (once), halo pixels -> 0 (once; the image fetches never touch them)
  for (int idx = tid; idx < 64 * 72; idx += 512) {
    const int r = idx / 72, ch = idx - r * 72;
    *(uint4*)(Wl + r * CN_WPITCH + ch * 16) = *(const uint4*)((const char*)Wt + (int64_t)r * 1152 + ch * 16);
  }
  for (int idx = tid; idx < 2 * 2 * CN_HALO * 8; idx += 512) {          // 2 stages x 2 halos x 17 rows x 8 chunks
    const int st = idx / (2 * CN_HALO * 8), rem = idx - st * (2 * CN_HALO * 8);
    const int hr = rem >> 3, ch = rem & 7;
    const int row = hr < CN_HALO ? hr : CN_HALO + 256 + (hr - CN_HALO);
    *(uint4*)(img0 + st * CN_STAGE + row * 128 + ch * 16) = uint4{0u, 0u, 0u, 0u};
  }

  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)CN_RECORDS, 0x00020000);
  // image fetch: 32 pieces of 8 pixels, 4 per wave; LDS row r = 17 + pixel holds source chunk c at chunk position c ^ (r & 7)
  uint32_t foff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int px = (wave * 4 + j) * 8 + (lane >> 3);
    const int r = CN_HALO + px;
    foff[j] = (uint32_t)(px * 128 + ((((lane & 7) ^ (r & 7))) << 4));
  }
  auto issue = [&](int img, int stage) {
    char* base = img0 + stage * CN_STAGE + CN_HALO * 128;
    const uint32_t ib = (uint32_t)img * 32768u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(base + (wave * 4 + j) * 1024), 16, ib + foff[j], 0, 0, 0);
  };

  // EPI_AFFINE (round 6: evaluation-mode BatchNorm folded into the convolution): out = [relu](acc * scale[c] + shift[c] (+ aux)); a
  // lane owns channels j * 16 + 4 g .. + 3 of its pixels for the whole walk
  floatx4 sc4[4], sh4[4];
  const bool affine = col_scale != nullptr;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sc4[j] = affine ? *(const floatx4*)(col_scale + j * 16 + 4 * g) : floatx4{1.f, 1.f, 1.f, 1.f};
    sh4[j] = affine ? *(const floatx4*)(col_shift + j * 16 + 4 * g) : floatx4{0.f, 0.f, 0.f, 0.f};
  }
  int img = blockIdx.x, stage = 0;
  if (img < nimg) issue(img, 0);
  for (; img < nimg; img += gridDim.x, stage ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // this image landed for every wave; the other stage is free (and W / halos are written)
    if (img + (int)gridDim.x < nimg) issue(img + gridDim.x, stage ^ 1);
    const char* It = img0 + stage * CN_STAGE;
    floatx4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      constexpr int SGN = FLIP ? -1 : 1;
      const int kh = tap / 3, kw = tap - kh * 3;
      const int dx = SGN * (kw - 1);
      const int sh = SGN * ((kh - 1) * 16 + (kw - 1));
      const bool dead = (dx < 0 && l15 == 0) || (dx > 0 && l15 == 15);      // the neighbour in x lies outside the image row
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t a[2], b[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = CN_HALO + wave * 32 + i * 16 + l15 + sh;
          a[i] = *(const bf16x8_t*)(It + r * 128 + (((ks * 4 + g) ^ (r & 7)) << 4));
          if (dx != 0 && dead) a[i] = bf16x8_t{};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8_t*)(Wl + (j * 16 + l15) * CN_WPITCH + (tap * 8 + ks * 4 + g) * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = MmaBF16::mma(b[j], a[i], acc[i][j]);
      }
    }
    bf16_t* o = out + ((int64_t)img * 256 + wave * 32) * 64;
    if (affine) {
      const bf16_t* ax = aux != nullptr ? aux + ((int64_t)img * 256 + wave * 32) * 64 : nullptr;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          floatx4 v = acc[i][j] * sc4[j] + sh4[j];
          if (ax != nullptr) v += load4<bf16_t>(ax + (i * 16 + l15) * 64 + j * 16 + 4 * g);
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          store4<bf16_t>(o + (i * 16 + l15) * 64 + j * 16 + 4 * g, v);
        }
      continue;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) store4<bf16_t>(o + (i * 16 + l15) * 64 + j * 16 + 4 * g, acc[i][j]);
  }
}

// out [rows][64] = conv (flip 0, Wt = [co][tap][ci]) or its input gradient (flip 1, Wt = [ci][tap][co]) over [rows][64] NHWC 16x16 maps
int conv_c64_nt(hipStream_t st, const bf16_t* X, const bf16_t* Wt, bf16_t* out, int rows, const int* rows_dev, int flip,
                const float* col_scale, const float* col_shift, const bf16_t* aux, int relu) {
  if (rows <= 0) return RL_OK;
  if ((rows % 256) || (int64_t)rows * 128 >= (int64_t)CN_RECORDS) return RL_ERR_ARG;
  int grid = rows / 256;
  if (grid > 256) grid = 256;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_c64_nt_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, CN_LDS);
    (void)hipFuncSetAttribute((const void*)conv_c64_nt_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CN_LDS);
    attr = true;
  }
  ProfScope ps(st, PK_CONV_NT, 2.0 * rows * 64 * 576);
  if (col_scale != nullptr && col_shift == nullptr) return RL_ERR_ARG;
  if (flip) RL_LAUNCH(conv_c64_nt_kernel<1>, dim3(grid), dim3(512), CN_LDS, st, X, Wt, out, rows, rows_dev, col_scale, col_shift, aux, relu);
  else RL_LAUNCH(conv_c64_nt_kernel<0>, dim3(grid), dim3(512), CN_LDS, st, X, Wt, out, rows, rows_dev, col_scale, col_shift, aux, relu);
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}

}  // namespace rl
