// MFMA GEMM kernels for gfx950 (CDNA4):
//   gemm_nt : C[M,N]  = A[M,K] . B[N,K]^T, fused epilogues   (Linear fwd, dgrad on W^T shadows,
//             classifier, GRU recurrent step, implicit-im2col conv fwd / dgrad)      K2 K4 K5 K6 K8 K12
//   gemm_tn : C[I,J] += sum_p A[p,i] B[p,j], fp32 atomics, split over p             K14 weight grads
// One code path for both numerics modes: T = bf16 (v_mfma_f32_16x16x32_bf16, speed mode) and
// T = float (v_mfma_f32_16x16x4_f32, exact-fp32 parity mode).
//
// Tile: 128x128 per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4 MFMA tiles),
// K-tile = one 128-byte LDS row per operand row (64 bf16 / 32 f32), double-buffered in LDS,
// register-staged (global -> VGPR issued before the MFMA block, VGPR -> LDS after it, one
// barrier per K-tile).  LDS rows are XOR-swizzled at 16-byte granularity so both the
// ds_write_b128 staging stores and the ds_read_b128 operand reads are bank-conflict free.
// Operands are passed to the MFMA swapped (B-fragment first) so each lane ends up with 4
// CONSECUTIVE OUTPUT COLUMNS of one row: epilogue loads/stores are 8-16 B per lane.
#include "gemm.h"

namespace rl {

template <typename T> struct Geo;
template <> struct Geo<bf16_t> { static constexpr int BK = 64, VEC = 8, KSTEPS = 2; };
template <> struct Geo<float> { static constexpr int BK = 32, VEC = 4, KSTEPS = 8; };

static constexpr int BM = 128, BN = 128;
static constexpr int TILE_BYTES = 128 * 128;

template <typename T>
__device__ __forceinline__ void epilogue4(const EpiParams<T>& ep, int M, int N, int row, int col, floatx4 v) {
  if (row >= M || col >= N) return;
  if (ep.alpha != 1.0f) v *= ep.alpha;
  if (ep.bias != nullptr) v += *(const floatx4*)(ep.bias + col);
  switch (ep.mode) {
    case EPI_STORE: {
      T* o = ep.out + (int64_t)row * ep.ldo + col;
      if (ep.accumulate) v += load4<T>(o);
      store4<T>(o, v);
    } break;
    case EPI_GELU: {
      if (ep.out2 != nullptr) store4<T>(ep.out2 + (int64_t)row * ep.ldo + col, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
      store4<T>(ep.out + (int64_t)row * ep.ldo + col, v);
    } break;
    case EPI_DROP_RESID: {
      const uint32_t idx = (uint32_t)row * (uint32_t)N + (uint32_t)col;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= drop_mult(ep.drop_seed, ep.drop_thresh, ep.drop_scale, idx + j);
      v += load4<T>(ep.aux + (int64_t)row * ep.ldaux + col);
      store4<T>(ep.out + (int64_t)row * ep.ldo + col, v);
    } break;
    case EPI_QKV: {
      const int H = ep.nh * 64;
      const int which = col / H, rem = col - which * H;
      const int head = rem >> 6, d = rem & 63;
      const int b = row / ep.S, s = row - b * ep.S;
      T* o = ep.out + (int64_t)which * ep.qkv_plane + ((int64_t)(b * ep.nh + head) * ep.S + s) * 64 + d;
      store4<T>(o, v);
    } break;
    case EPI_GELU_BWD: {
      const floatx4 x = load4<T>(ep.aux + (int64_t)row * ep.ldaux + col);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_grad(x[j]);
      T* o = ep.out + (int64_t)row * ep.ldo + col;
      if (ep.accumulate) v += load4<T>(o);
      store4<T>(o, v);
    } break;
    default: break;
  }
}

template <typename T, typename ALoader>
__global__ void __launch_bounds__(256, 2)
gemm_nt_kernel(ALoader la, DenseLoader<T> lb, int M, int N, int K, int tiles_n, int ntiles, EpiParams<T> ep) {
  typedef typename MmaOf<T>::type Mma;
  typedef Geo<T> G;
  typedef KTile<T, G::BK> LT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int sc = tid & 7, sr = tid >> 3;      // staging: 8 x 16-byte chunks per row, 32 rows per pass

  typename ALoader::Ctx actx[4];
  typename DenseLoader<T>::Ctx bctx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    actx[i] = la.prepare(m0 + sr + 32 * i);
    bctx[i] = lb.prepare(n0 + sr + 32 * i);
  }
  uint4 ar[4], br[4];
  const int nk = (K + G::BK - 1) / G::BK;

  floatx4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  {
    const int k = sc * G::VEC;
#pragma unroll
    for (int i = 0; i < 4; ++i) { ar[i] = la.load(actx[i], k); br[i] = lb.load(bctx[i], k); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(uint4*)(smem + LT::off(sr + 32 * i, sc)) = ar[i];
      *(uint4*)(smem + TILE_BYTES + LT::off(sr + 32 * i, sc)) = br[i];
    }
  }
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) {
      const int k = (kt + 1) * G::BK + sc * G::VEC;
#pragma unroll
      for (int i = 0; i < 4; ++i) { ar[i] = la.load(actx[i], k); br[i] = lb.load(bctx[i], k); }
    }
    const char* As = smem + cur * 2 * TILE_BYTES;
    const char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks) {
      typename Mma::Frag a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = ktile_frag<T, G::BK>(As, wm * 64 + i * 16 + l15, ks, g);
        b[i] = ktile_frag<T, G::BK>(Bs, wn * 64 + i * 16 + l15, ks, g);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mma::mma(b[j], a[i], acc[i][j]);
    }
    if (more) {
      char* An = smem + (cur ^ 1) * 2 * TILE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *(uint4*)(An + LT::off(sr + 32 * i, sc)) = ar[i];
        *(uint4*)(An + TILE_BYTES + LT::off(sr + 32 * i, sc)) = br[i];
      }
    }
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      epilogue4<T>(ep, M, N, m0 + wm * 64 + i * 16 + l15, n0 + wn * 64 + j * 16 + 4 * g, acc[i][j]);
}

template <typename T, typename ALoader>
static int launch_nt(hipStream_t st, const ALoader& la, const T* B, int64_t ldb, int M, int N, int K,
                     const EpiParams<T>& ep) {
  if (M <= 0 || N <= 0 || K <= 0) return RL_OK;
  if ((N & 3) || (K % Geo<T>::VEC) || (ldb % Geo<T>::VEC)) return RL_ERR_ARG;
  DenseLoader<T> lb{B, ldb, N, K};
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  const size_t lds = 4 * TILE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, ALoader>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<T, ALoader>), dim3(ntiles), dim3(256), lds, st, la, lb, M, N, K, tiles_n, ntiles, ep);
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}

template <typename T>
int gemm_nt(hipStream_t st, const T* A, int64_t lda, const T* B, int64_t ldb, int M, int N, int K,
            const EpiParams<T>& ep) {
  if (lda % Geo<T>::VEC) return RL_ERR_ARG;
  DenseLoader<T> la{A, lda, M, K};
  return launch_nt<T, DenseLoader<T>>(st, la, B, ldb, M, N, K, ep);
}
template <typename T>
int gemm_nt_conv(hipStream_t st, const ConvLoader<T>& la, const T* B, int64_t ldb, int M, int N, int K,
                 const EpiParams<T>& ep) {
  if (la.C % Geo<T>::VEC) return RL_ERR_ARG;
  return launch_nt<T, ConvLoader<T>>(st, la, B, ldb, M, N, K, ep);
}
template int gemm_nt<bf16_t>(hipStream_t, const bf16_t*, int64_t, const bf16_t*, int64_t, int, int, int, const EpiParams<bf16_t>&);
template int gemm_nt<float>(hipStream_t, const float*, int64_t, const float*, int64_t, int, int, int, const EpiParams<float>&);
template int gemm_nt_conv<bf16_t>(hipStream_t, const ConvLoader<bf16_t>&, const bf16_t*, int64_t, int, int, int, const EpiParams<bf16_t>&);
template int gemm_nt_conv<float>(hipStream_t, const ConvLoader<float>&, const float*, int64_t, int, int, int, const EpiParams<float>&);

// =================================================================================================
// TN: C[I,J] += sum_p A[p,i] * B[p,j].  Operand tiles are staged in their natural layout
// ([p][feature], feature contiguous) and transposed on the LDS read: ds_read_b64_tr_b16 for bf16
// (TR = true), 16-bit gathers otherwise; a single ds_read_b32 per operand for the fp32 atom.
// =================================================================================================
template <typename T> struct TnGeo;
template <> struct TnGeo<bf16_t> { static constexpr int BP = 32, ROWB = 256, CPR = 16, RPP = 16, PASSES = 2, VEC = 8, KSTEPS = 1; };
template <> struct TnGeo<float> { static constexpr int BP = 32, ROWB = 512, CPR = 32, RPP = 8, PASSES = 4, VEC = 4, KSTEPS = 8; };

template <typename T> __device__ __forceinline__ int tn_swz(int p);
template <> __device__ __forceinline__ int tn_swz<bf16_t>(int p) { return ((p >> 3) & 3) << 5; }
template <> __device__ __forceinline__ int tn_swz<float>(int p) { return (p & 1) << 6; }

template <bool TR>
__device__ __forceinline__ bf16x8_t tn_frag_bf16(const char* tile, int col0, int l15, int g) {
  if constexpr (TR) {
    // 16-lane group g reads the [4 p][16 col] block rows 8g+4h .. +3; lane q of the group points at
    // row (q >> 2), columns 4*(q & 3) .. +3 and receives column q, 4 consecutive p.
    typedef short4_t __attribute__((address_space(3))) * lds_s4;
    short4_t h[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int p = 8 * g + 4 * hh + (l15 >> 2);
      const int colb = ((col0 + 4 * (l15 & 3)) * 2) ^ tn_swz<bf16_t>(p);
      h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(tile + p * 256 + colb));
    }
    typedef __attribute__((ext_vector_type(8))) short short8_t;
    short8_t r = {h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
    return __builtin_bit_cast(bf16x8_t, r);
  } else {
    typedef __attribute__((ext_vector_type(8))) unsigned short ushort8_t;
    ushort8_t r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = 8 * g + e;
      r[e] = *(const unsigned short*)(tile + p * 256 + (((col0 + l15) * 2) ^ tn_swz<bf16_t>(p)));
    }
    return __builtin_bit_cast(bf16x8_t, r);
  }
}

__device__ __forceinline__ void tn_epilogue4(const TnEpi& ep, int I, int J, int i, int j, floatx4 v) {
  if (i >= I) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int jj = j + r;
    if (jj >= J) continue;
    const float x = v[r] * ep.alpha;
    if (ep.mode == TN_PLAIN) {
      atomicAdd(ep.out + (int64_t)i * ep.ldo + jj, x);
    } else {
      const int tap = jj / ep.Cpad, ci = jj - tap * ep.Cpad;
      if (ci < ep.Cin) atomicAdd(ep.out + ((int64_t)i * ep.Cin + ci) * ep.KHW + tap, x);
    }
  }
}

template <typename T, typename BLoader, bool TR>
__global__ void __launch_bounds__(256, 2)
gemm_tn_kernel(const T* __restrict__ A, int64_t lda, BLoader lb, int P, int I, int J, int tiles_j, int ntiles,
               int pchunk, TnEpi ep) {
  typedef typename MmaOf<T>::type Mma;
  typedef TnGeo<T> G;
  constexpr int TILEB = G::BP * G::ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (A tile, B tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int wi = wave >> 1, wj = wave & 1;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
  const int i0 = ti * 128, j0 = tj * 128;
  const int p_begin = blockIdx.y * pchunk;
  const int p_end = min(P, p_begin + pchunk);
  if (p_begin >= p_end) return;
  const int sc = tid % G::CPR, sr = tid / G::CPR;
  const int acol = i0 + sc * G::VEC, bcol = j0 + sc * G::VEC;
  const bool a_ok = acol < I;
  uint4 ar[G::PASSES], br[G::PASSES];

  floatx4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  auto gload = [&](int pt) {
#pragma unroll
    for (int s = 0; s < G::PASSES; ++s) {
      const int p = pt + sr + s * G::RPP;
      const bool ok = p < p_end;
      ar[s] = (ok && a_ok) ? *(const uint4*)(A + (int64_t)p * lda + acol) : make_uint4(0, 0, 0, 0);
      if (ok && bcol < J) {
        typename BLoader::Ctx c = lb.prepare(p);
        br[s] = lb.load(c, bcol);
      } else {
        br[s] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto sstore = [&](int stage) {
    char* At = smem + stage * 2 * TILEB;
    char* Bt = At + TILEB;
#pragma unroll
    for (int s = 0; s < G::PASSES; ++s) {
      const int pl = sr + s * G::RPP;
      const int off = pl * G::ROWB + ((sc * 16) ^ tn_swz<T>(pl));
      *(uint4*)(At + off) = ar[s];
      *(uint4*)(Bt + off) = br[s];
    }
  };

  gload(p_begin);
  sstore(0);
  __syncthreads();
  int cur = 0;
  for (int pt = p_begin; pt < p_end; pt += G::BP) {
    const bool more = pt + G::BP < p_end;
    if (more) gload(pt + G::BP);
    const char* At = smem + cur * 2 * TILEB;
    const char* Bt = At + TILEB;
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks) {
      typename Mma::Frag a[4], b[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int ca = wi * 64 + f * 16, cb = wj * 64 + f * 16;
        if constexpr (sizeof(T) == 2) {
          a[f] = tn_frag_bf16<TR>(At, ca, l15, g);
          b[f] = tn_frag_bf16<TR>(Bt, cb, l15, g);
        } else {
          const int p = ks * 4 + g;
          a[f] = *(const float*)(At + p * G::ROWB + (((ca + l15) * 4) ^ tn_swz<float>(p)));
          b[f] = *(const float*)(Bt + p * G::ROWB + (((cb + l15) * 4) ^ tn_swz<float>(p)));
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mma::mma(b[j], a[i], acc[i][j]);
    }
    if (more) sstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      tn_epilogue4(ep, I, J, i0 + wi * 64 + i * 16 + l15, j0 + wj * 64 + j * 16 + 4 * g, acc[i][j]);
}

static int g_tn_tr = 1;   // ds_read_b64_tr_b16 verified on MI355X (tests/test_kernels_gpu.py::test_gemm_tn)
void set_tn_transpose_read(int use_tr) { g_tn_tr = use_tr; }

template <typename T, typename BLoader>
static int launch_tn(hipStream_t st, const T* A, int64_t lda, const BLoader& lb, int P, int I, int J, const TnEpi& ep) {
  if (P <= 0 || I <= 0 || J <= 0) return RL_OK;
  typedef TnGeo<T> G;
  if ((lda % G::VEC) || (I % G::VEC) || (J % G::VEC)) return RL_ERR_ARG;
  const int tiles_i = (I + 127) / 128, tiles_j = (J + 127) / 128, ntiles = tiles_i * tiles_j;
  // split the reduction so the grid fills the chip (~4 workgroups per CU), chunks multiple of BP
  int nsplit = (1024 + ntiles - 1) / ntiles;
  const int max_split = (P + 4 * G::BP - 1) / (4 * G::BP);
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1) nsplit = 1;
  int pchunk = (P + nsplit - 1) / nsplit;
  pchunk = ((pchunk + G::BP - 1) / G::BP) * G::BP;
  nsplit = (P + pchunk - 1) / pchunk;
  const size_t lds = 4 * (size_t)G::BP * G::ROWB;
  dim3 grid(ntiles, nsplit);
  if (sizeof(T) == 2 && g_tn_tr) {
    hipLaunchKernelGGL((gemm_tn_kernel<T, BLoader, true>), grid, dim3(256), lds, st, A, lda, lb, P, I, J, tiles_j, ntiles, pchunk, ep);
  } else {
    hipLaunchKernelGGL((gemm_tn_kernel<T, BLoader, false>), grid, dim3(256), lds, st, A, lda, lb, P, I, J, tiles_j, ntiles, pchunk, ep);
  }
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}

template <typename T>
int gemm_tn(hipStream_t st, const T* A, int64_t lda, const T* B, int64_t ldb, int P, int I, int J, const TnEpi& ep) {
  if (ldb % TnGeo<T>::VEC) return RL_ERR_ARG;
  DenseLoader<T> lb{B, ldb, P, J};
  return launch_tn<T, DenseLoader<T>>(st, A, lda, lb, P, I, J, ep);
}
template <typename T>
int gemm_tn_conv(hipStream_t st, const T* A, int64_t lda, const ConvLoader<T>& lb, int P, int I, int J, const TnEpi& ep) {
  if (lb.C % TnGeo<T>::VEC) return RL_ERR_ARG;
  return launch_tn<T, ConvLoader<T>>(st, A, lda, lb, P, I, J, ep);
}
template int gemm_tn<bf16_t>(hipStream_t, const bf16_t*, int64_t, const bf16_t*, int64_t, int, int, int, const TnEpi&);
template int gemm_tn<float>(hipStream_t, const float*, int64_t, const float*, int64_t, int, int, int, const TnEpi&);
template int gemm_tn_conv<bf16_t>(hipStream_t, const bf16_t*, int64_t, const ConvLoader<bf16_t>&, int, int, int, const TnEpi&);
template int gemm_tn_conv<float>(hipStream_t, const float*, int64_t, const ConvLoader<float>&, int, int, int, const TnEpi&);

}  // namespace rl
