// HBM-bound kernels, part 2: pinyin-GRU gate math (K6), BatchNorm finalize/apply (K9),
// operand shadows (compute-dtype copies of the fp32 master weights), optimizer (K15/K16).
#include <algorithm>
#include "ops.h"

namespace rl {

#define RL_LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH)

// ---------------------------------------------------------------------------------------------
// GRU (models.py:661-669, 818-826).  The input projection only ever sees 33 distinct embedding
// rows, so W_ih x + b_ih is a [33][3H] table computed once per forward (fp32, from the masters).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void gru_step_fwd_kernel(GruStepArgs<T> a) {    // grid (H/4/64, n_alive)
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int i = blockIdx.y;
  if (c >= a.H) return;
  if (a.n_alive_dev != nullptr && i >= *a.n_alive_dev) return;
  const int H = a.H;
  const int tok = a.perm[i];
  const int64_t v = a.pho_idx[(int64_t)tok * a.Tp + a.t];
  const float* gi = a.table + v * 3 * H + c;
  const floatx4 ir = *(const floatx4*)gi, iz = *(const floatx4*)(gi + H), in = *(const floatx4*)(gi + 2 * H);
  floatx4 hr, hz, hn, hp;
  if (a.gh != nullptr) {
    const T* gh = a.gh + (int64_t)i * 3 * H + c;
    hr = load4<T>(gh); hz = load4<T>(gh + H); hn = load4<T>(gh + 2 * H);
    hp = load4<T>(a.h_prev + (int64_t)i * H + c);
  } else {
    hr = *(const floatx4*)(a.b_hh + c); hz = *(const floatx4*)(a.b_hh + H + c); hn = *(const floatx4*)(a.b_hh + 2 * H + c);
    hp = floatx4{0.f, 0.f, 0.f, 0.f};
  }
  floatx4 r, z, n, h;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float rj, zj, nj, hj;
    gru_unit<T>(ir[j], iz[j], in[j], hr[j], hz[j], hn[j], hp[j], rj, zj, nj, hj);
    r[j] = rj; z[j] = zj; n[j] = nj; h[j] = hj;
  }
  if (a.rzn != nullptr) {
    T* s = a.rzn + (int64_t)i * 3 * H + c;
    store4<T>(s, r); store4<T>(s + H, z); store4<T>(s + 2 * H, n);
  }
  store4<T>(a.h_new + (int64_t)i * H + c, h);
  if (a.lens[i] == a.t + 1) store4<T>(a.out + (int64_t)tok * H + c, h);
}
template <typename T> int gru_step_fwd(hipStream_t st, const GruStepArgs<T>& a) {
  if (a.n_alive <= 0) return RL_OK;
  if (a.H & 3) return RL_ERR_ARG;
  hipLaunchKernelGGL((gru_step_fwd_kernel<T>), dim3((a.H / 4 + 63) / 64, a.n_alive), dim3(64), 0, st, a);
  return RL_LAUNCH_CHECK();
}
// ---- device-side pinyin batch ----------------------------------------------------------------------------------------
// One 1024-thread workgroup: each thread owns a contiguous run of tokens (so ranks inside a length class follow the token
// order = a STABLE sort, like the host's argsort(-lens, kind="stable")); one block-wide exclusive scan per length class.
__global__ void __launch_bounds__(1024) pho_prepare_kernel(const int64_t* __restrict__ src, int T_, const int64_t* __restrict__ table,
                                                            const int32_t* __restrict__ vlens, int V, int Tw, int64_t* __restrict__ pho_idx,
                                                            int32_t* __restrict__ perm, int32_t* __restrict__ lens_sorted,
                                                            int32_t* __restrict__ n_alive) {
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = (T_ + 1023) / 1024;
  const int t0 = min(T_, tid * per), t1 = min(T_, t0 + per);
  auto len_of = [&](int t) {
    const int64_t v = src[t];
    const int l = (v >= 0 && v < V) ? vlens[v] : 1;
    return min(max(l, 1), Tw);
  };
  // (the per-vocabulary rows are gathered by pho_gather_kernel, one thread per symbol: this workgroup only sorts)
  constexpr int CACHE = 16;                               // lengths of the thread's run, read once (T_ <= 16384; longer runs re-read)
  int myl[CACHE];
#pragma unroll
  for (int k = 0; k < CACHE; ++k) myl[k] = (per <= CACHE && t0 + k < t1) ? len_of(t0 + k) : 0;
  auto len_at = [&](int t) {
    if (per > CACHE) return len_of(t);
    int l = 0;
#pragma unroll
    for (int k = 0; k < CACHE; ++k) if (k == t - t0) l = myl[k];
    return l;
  };
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int L = Tw; L >= 1; --L) {                         // longest first
    int c = 0;
    for (int t = t0; t < t1; ++t) c += len_at(t) == L;
    int incl = c;                                         // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 16; ++w) { const int x = wave_tot[w]; if (w < wv) before += x; total += x; }
    const int base = base_s;
    int pos = base + before + incl - c;
    for (int t = t0; t < t1; ++t)
      if (len_at(t) == L) { perm[pos] = t; lens_sorted[pos] = L; ++pos; }
    __syncthreads();
    if (tid == 0) {
      base_s = base + total;
      n_alive[L - 1] = base + total;                      // #{len > L - 1} = #{len >= L}
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) pho_gather_kernel(const int64_t* __restrict__ src, int T_, const int64_t* __restrict__ table, int V, int Tw,
                                                         int64_t* __restrict__ pho_idx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)T_ * Tw) return;
  const int t = (int)(i / Tw), k = (int)(i - (int64_t)t * Tw);
  const int64_t v = src[t];
  pho_idx[i] = (v >= 0 && v < V) ? table[v * Tw + k] : (k == 0 ? 32 : 0);
}
// Token-id range check (nn.Embedding raises on an id outside its table, modeling_bert.py:183-186; models.py:818,831): the engine
// reads CLEAN copies - ids clamped into [0, V) so that nothing downstream can index past a table - and a sticky flag (device- or
// host-mapped int) records that a bad id was seen; the module raises IndexError when it reads the flag.
__global__ void sanitize_ids_kernel(const int64_t* __restrict__ ids, int64_t n, int V, int64_t* __restrict__ out, int* flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = ids[i];
  const bool bad = v < 0 || v >= V;
  out[i] = bad ? 0 : v;
  if (bad && flag != nullptr) *flag = 1;
}
int sanitize_ids(hipStream_t st, const int64_t* ids, int64_t n, int V, int64_t* out, int* flag) {
  if (n <= 0) return RL_OK;
  hipLaunchKernelGGL(sanitize_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ids, n, V, out, flag);
  return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_LAUNCH;
}
int pho_prepare(hipStream_t st, const int64_t* src, int T_, const int64_t* table, const int32_t* vlens, int V, int Tw,
                int64_t* pho_idx, int32_t* perm, int32_t* lens_sorted, int32_t* n_alive) {
  if (T_ <= 0) return RL_OK;
  if (Tw < 1 || Tw > 16 || V < 1) return RL_ERR_ARG;
  hipLaunchKernelGGL(pho_gather_kernel, dim3((unsigned)(((int64_t)T_ * Tw + 255) / 256)), dim3(256), 0, st, src, T_, table, V, Tw, pho_idx);
  hipLaunchKernelGGL(pho_prepare_kernel, dim3(1), dim3(1024), 0, st, src, T_, table, vlens, V, Tw, pho_idx, perm, lens_sorted, n_alive);
  return RL_LAUNCH_CHECK();
}
template int gru_step_fwd<bf16_t>(hipStream_t, const GruStepArgs<bf16_t>&);
template int gru_step_fwd<float>(hipStream_t, const GruStepArgs<float>&);

// one BPTT step: consumes dL/dh_t (dh, or dout where the sequence ended at t), emits dGi, dGh,
// the one-hot of the input letter (for the table gradient GEMM) and dh * z into dh.
template <typename T>
__global__ void gru_step_bwd_kernel(GruStepArgs<T> a) {    // grid (H/4/64, n_alive)
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int i = blockIdx.y;
  if (c >= a.H) return;
  if (a.n_alive_dev != nullptr && i >= *a.n_alive_dev) return;
  const int H = a.H;
  const int tok = a.perm[i];
  const bool ends_here = a.lens[i] == a.t + 1;
  const floatx4 dh = ends_here ? load4<T>(a.dout + (int64_t)tok * H + c) : load4<T>(a.dh + (int64_t)i * H + c);
  const T* s = a.rzn + (int64_t)i * 3 * H + c;
  const floatx4 r = load4<T>(s), z = load4<T>(s + H), n = load4<T>(s + 2 * H);
  floatx4 hn, hp;
  if (a.gh != nullptr) {
    hn = load4<T>(a.gh + (int64_t)i * 3 * H + 2 * H + c);
    hp = load4<T>(a.h_prev + (int64_t)i * H + c);
  } else {
    hn = *(const floatx4*)(a.b_hh + 2 * H + c);
    hp = floatx4{0.f, 0.f, 0.f, 0.f};
  }
  floatx4 dar, daz, dan, dhn, dhp;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float dn = dh[j] * (1.0f - z[j]);
    const float dz = dh[j] * (hp[j] - n[j]);
    dan[j] = dn * (1.0f - n[j] * n[j]);
    dar[j] = dan[j] * hn[j] * r[j] * (1.0f - r[j]);
    daz[j] = dz * z[j] * (1.0f - z[j]);
    dhn[j] = dan[j] * r[j];
    dhp[j] = dh[j] * z[j];
  }
  T* gi = a.dgi + (int64_t)i * 3 * H + c;
  store4<T>(gi, dar); store4<T>(gi + H, daz); store4<T>(gi + 2 * H, dan);
  T* gh = a.dgh + (int64_t)i * 3 * H + c;
  store4<T>(gh, dar); store4<T>(gh + H, daz); store4<T>(gh + 2 * H, dhn);
  store4<T>(a.dh + (int64_t)i * H + c, dhp);
  if (c < 64) {
    const int v = (int)a.pho_idx[(int64_t)tok * a.Tp + a.t];
    floatx4 oh;
#pragma unroll
    for (int j = 0; j < 4; ++j) oh[j] = (c + j) == v ? 1.0f : 0.0f;
    store4<T>(a.onehot + (int64_t)i * 64 + c, oh);
  }
}
template <typename T> int gru_step_bwd(hipStream_t st, const GruStepArgs<T>& a) {
  if (a.n_alive <= 0) return RL_OK;
  if ((a.H & 3) || a.H < 64) return RL_ERR_ARG;
  hipLaunchKernelGGL((gru_step_bwd_kernel<T>), dim3((a.H / 4 + 63) / 64, a.n_alive), dim3(64), 0, st, a);
  return RL_LAUNCH_CHECK();
}
template int gru_step_bwd<bf16_t>(hipStream_t, const GruStepArgs<bf16_t>&);
template int gru_step_bwd<float>(hipStream_t, const GruStepArgs<float>&);

// The pinyin alphabet has 33 symbols, so the GRU's input projection W_ih x + b_ih is a [V <= 64][3H] table (models.py:818-826).
// Forward, speed mode: one wave per output unit j keeps W_ih[j, :] in registers (H / 64 values per lane), and accumulates the V dot
// products with the embedding rows (L1-resident, 100 KB) - W_ih is read once, 7 MB; the general MFMA GEMM ran the 33-row problem as
// 18 workgroups of 128 x 128 tiles (56 us).
template <int NK>
__global__ void __launch_bounds__(256) gru_table_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ w_ih, const float* __restrict__ b_ih,
                                                            int V, int H, int J, float* __restrict__ table) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + wave;
  if (j >= J) return;
  float w[NK];
#pragma unroll
  for (int i = 0; i < NK; ++i) { const int k = i * 64 + lane; w[i] = k < H ? w_ih[(int64_t)j * H + k] : 0.f; }
  const float bias = b_ih[j];
  // three symbols per trip: their 3 * NK embedding loads are in flight together (one symbol per trip paid an L2 latency per symbol: 35 us)
  for (int v0 = 0; v0 < V; v0 += 3) {
    float e[3][NK];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int i = 0; i < NK; ++i) { const int k = i * 64 + lane; e[u][i] = (k < H && v0 + u < V) ? emb[(int64_t)(v0 + u) * H + k] : 0.f; }
    float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int i = 0; i < NK; ++i) s[u] = fmaf(e[u][i], w[i], s[u]);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const float t = wave_sum(s[u]);
      if (lane == 0 && v0 + u < V) table[(int64_t)(v0 + u) * J + j] = t + bias;
    }
  }
}
int gru_table_fwd(hipStream_t st, const float* emb, const float* w_ih, const float* b_ih, int V, int H, float* table) {
  if (H > 1024 || V < 1) return RL_ERR_ARG;
  const int J = 3 * H, NK = (H + 63) / 64;
  const dim3 grid((J + 3) / 4), block(256);
  if (NK <= 4) hipLaunchKernelGGL((gru_table_fwd_kernel<4>), grid, block, 0, st, emb, w_ih, b_ih, V, H, J, table);
  else if (NK <= 8) hipLaunchKernelGGL((gru_table_fwd_kernel<8>), grid, block, 0, st, emb, w_ih, b_ih, V, H, J, table);
  else if (NK <= 12) hipLaunchKernelGGL((gru_table_fwd_kernel<12>), grid, block, 0, st, emb, w_ih, b_ih, V, H, J, table);
  else hipLaunchKernelGGL((gru_table_fwd_kernel<16>), grid, block, 0, st, emb, w_ih, b_ih, V, H, J, table);
  return RL_LAUNCH_CHECK();
}

// dTable [V][3H] -> d b_ih, d W_ih [3H][H], d Emb [V][H].  Every element of d W_ih has ONE owner thread (plain add, no atomics: 1.8 M
// atomics were the whole cost of this kernel); d Emb: a thread owns column k for ALL V rows over a chunk of 64 units, so W_ih is
// read once (7 MB; it was read once per symbol, 33 x) and the chunks meet in V atomics per thread.
__global__ void __launch_bounds__(256)
gru_table_bwd_w_kernel(const float* __restrict__ dt, int ldt, const float* __restrict__ emb, int V, int H,
                       float* d_w_ih, float* d_b_ih) {                  // block per output unit j, thread per k
  __shared__ float dcol[64];
  const int j = blockIdx.x;
  if (threadIdx.x < 64) dcol[threadIdx.x] = (int)threadIdx.x < V ? dt[(int64_t)threadIdx.x * ldt + j] : 0.f;
  __syncthreads();
  if (threadIdx.x == 0) {
    float bsum = 0.f;
    for (int v = 0; v < V; ++v) bsum += dcol[v];
    d_b_ih[j] += bsum;
  }
  for (int k = threadIdx.x; k < H; k += 256) {
    float s = 0.f;
    for (int v = 0; v < V; ++v) s = fmaf(dcol[v], emb[(int64_t)v * H + k], s);
    d_w_ih[(int64_t)j * H + k] += s;
  }
}
constexpr int GRU_TBE_CHUNK = 64, GRU_TBE_VG = 12;
__global__ void __launch_bounds__(256)
gru_table_bwd_e_kernel(const float* __restrict__ dt, int ldt, const float* __restrict__ w_ih, int V, int H, float* d_emb) {
  __shared__ float dts[GRU_TBE_VG][GRU_TBE_CHUNK];                        // this block's [12 symbols][64 units] slice of dTable
  const int k = blockIdx.x * 256 + threadIdx.x;                           // grid (H / 256, 3H / 64, V / 12)
  const int j0 = blockIdx.y * GRU_TBE_CHUNK, nj = min(GRU_TBE_CHUNK, 3 * H - j0);
  const int v0 = blockIdx.z * GRU_TBE_VG;
  for (int e = threadIdx.x; e < GRU_TBE_VG * GRU_TBE_CHUNK; e += 256) {
    const int v = e / GRU_TBE_CHUNK, jj = e - v * GRU_TBE_CHUNK;
    dts[v][jj] = (jj < nj && v0 + v < V) ? dt[(int64_t)(v0 + v) * ldt + j0 + jj] : 0.f;
  }
  __syncthreads();
  if (k >= H) return;
  float acc[GRU_TBE_VG];
#pragma unroll
  for (int v = 0; v < GRU_TBE_VG; ++v) acc[v] = 0.f;
  for (int jj = 0; jj < nj; ++jj) {
    const float w = w_ih[(int64_t)(j0 + jj) * H + k];
#pragma unroll
    for (int v = 0; v < GRU_TBE_VG; ++v) acc[v] = fmaf(dts[v][jj], w, acc[v]);
  }
#pragma unroll
  for (int v = 0; v < GRU_TBE_VG; ++v)                                    // padding_idx = 0 row gets no gradient
    if (v0 + v >= 1 && v0 + v < V) atomicAdd(d_emb + (int64_t)(v0 + v) * H + k, acc[v]);
}
int gru_table_bwd(hipStream_t st, const float* dtable, int ld_dtable, const float* emb, const float* w_ih, int V, int H,
                  float* d_emb, float* d_w_ih, float* d_b_ih) {
  if (V > 64) return RL_ERR_ARG;               // (gru_table_bwd_w_kernel keeps a symbol column of dTable in 64 LDS floats)
  hipLaunchKernelGGL(gru_table_bwd_w_kernel, dim3(3 * H), dim3(256), 0, st, dtable, ld_dtable, emb, V, H, d_w_ih, d_b_ih);
  hipLaunchKernelGGL(gru_table_bwd_e_kernel, dim3((H + 255) / 256, (3 * H + GRU_TBE_CHUNK - 1) / GRU_TBE_CHUNK, (V + GRU_TBE_VG - 1) / GRU_TBE_VG), dim3(256), 0, st,
                     dtable, ld_dtable, w_ih, V, H, d_emb);
  return RL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// BatchNorm2d over NHWC [P][C] (char_cnn.py:17-28): statistics come from col_sum /
// col_sumsq_centered (ops.hip); these kernels finalise and apply them.
// ---------------------------------------------------------------------------------------------
__global__ void bn_mean_kernel(const float* sum, int C, float inv_p, float* mean) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) mean[c] = sum[c] * inv_p;
}
int bn_finalize_mean(hipStream_t st, const float* sum, int C, int P, float* mean) {
  hipLaunchKernelGGL(bn_mean_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sum, C, 1.0f / (float)P, mean);
  return RL_LAUNCH_CHECK();
}
__global__ void bn_train_kernel(const float* mean, const float* sqsum, int C, int P, const float* gamma, const float* beta,
                                float eps, float momentum, float* rmean, float* rvar, float* rstd, float* scale, float* shift, int64_t* nbt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt != nullptr) *nbt += 1;                               // num_batches_tracked
  if (c >= C) return;
  const float var = sqsum[c] / (float)P;                                 // biased: used to normalise
  const float unbiased = sqsum[c] / (float)(P > 1 ? P - 1 : 1);          // running_var update
  const float rs = 1.0f / sqrtf(var + eps);
  rstd[c] = rs;
  scale[c] = gamma[c] * rs;
  shift[c] = beta[c] - mean[c] * gamma[c] * rs;
  rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean[c];
  rvar[c] = (1.0f - momentum) * rvar[c] + momentum * unbiased;
}
int bn_finalize_train(hipStream_t st, const float* mean, const float* sqsum, int C, int P, const float* gamma, const float* beta,
                      float eps, float momentum, float* running_mean, float* running_var, float* rstd, float* scale, float* shift,
                      int64_t* num_batches_tracked) {
  hipLaunchKernelGGL(bn_train_kernel, dim3((C + 255) / 256), dim3(256), 0, st, mean, sqsum, C, P, gamma, beta, eps, momentum,
                     running_mean, running_var, rstd, scale, shift, num_batches_tracked);
  return RL_LAUNCH_CHECK();
}
__global__ void bn_eval_kernel(int C, const float* gamma, const float* beta, float eps, const float* rmean, const float* rvar,
                               float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rs = 1.0f / sqrtf(rvar[c] + eps);
  scale[c] = gamma[c] * rs;
  shift[c] = beta[c] - rmean[c] * gamma[c] * rs;
}
int bn_finalize_eval(hipStream_t st, int C, const float* gamma, const float* beta, float eps, const float* running_mean,
                     const float* running_var, float* scale, float* shift) {
  hipLaunchKernelGGL(bn_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, st, C, gamma, beta, eps, running_mean, running_var,
                     scale, shift);
  return RL_LAUNCH_CHECK();
}

template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x1, const float* __restrict__ sc1, const float* __restrict__ sh1,
                                const T* __restrict__ x2, const float* __restrict__ sc2, const float* __restrict__ sh2,
                                T* __restrict__ y, int P, int C, int relu, RowBound rb) {
  const int64_t n = (int64_t)rb_rows(rb, P) * C;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    const int c = (int)(i % C);
    floatx4 v = load4<T>(x1 + i) * *(const floatx4*)(sc1 + c) + *(const floatx4*)(sh1 + c);
    if (x2 != nullptr) v += load4<T>(x2 + i) * *(const floatx4*)(sc2 + c) + *(const floatx4*)(sh2 + c);
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    store4<T>(y + i, v);
  }
}
// ---- bf16 fast paths of the BatchNorm maps: 16-byte accesses, C and hw powers of two ------------------------------------------
// A thread owns 8 consecutive channels; the grid stride (workgroups x 2048 elements) is a multiple of C, so its channels never
// change and scale / shift / mean / rstd / gamma live in registers for the whole loop; rows and glyph multiplicities come from
// shifts (the generic kernels pay a 64-bit i % C and i / C per 4 elements); four items are loaded before the first is used.
static inline int pow2_shift2(int C) { int s = 0; while ((1 << s) < C) ++s; return (1 << s) == C ? s : -1; }
struct F8 { floatx4 lo, hi; };
__device__ __forceinline__ F8 ld8f(const float* p) { return F8{*(const floatx4*)p, *(const floatx4*)(p + 4)}; }

template <int NX>
__global__ void __launch_bounds__(256) bn_apply16_kernel(const bf16_t* __restrict__ x1, const float* __restrict__ sc1, const float* __restrict__ sh1,
                                                          const bf16_t* __restrict__ x2, const float* __restrict__ sc2, const float* __restrict__ sh2,
                                                          bf16_t* __restrict__ y, int P, int c_shift, int relu, RowBound rb) {
  const int64_t n8 = ((int64_t)rb_rows(rb, P) << c_shift) >> 3;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)((i << 3) & ((1 << c_shift) - 1));
  const F8 s1 = ld8f(sc1 + c), h1 = ld8f(sh1 + c);
  F8 s2 = s1, h2 = h1;
  if constexpr (NX == 2) { s2 = ld8f(sc2 + c); h2 = ld8f(sh2 + c); }
  auto one = [&](int64_t k, uint4 u1, uint4 u2) {
    floatx4 lo, hi; unpack8(u1, lo, hi);
    lo = lo * s1.lo + h1.lo; hi = hi * s1.hi + h1.hi;
    if constexpr (NX == 2) {
      floatx4 lo2, hi2; unpack8(u2, lo2, hi2);
      lo += lo2 * s2.lo + h2.lo; hi += hi2 * s2.hi + h2.hi;
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { lo[j] = fmaxf(lo[j], 0.f); hi[j] = fmaxf(hi[j], 0.f); }
    }
    store8<bf16_t>(y + (k << 3), lo, hi);
  };
  const uint4* p1 = (const uint4*)x1; const uint4* p2 = (const uint4*)x2;
  for (; i + 3 * stride < n8; i += 4 * stride) {
    const uint4 a0 = p1[i], a1 = p1[i + stride], a2 = p1[i + 2 * stride], a3 = p1[i + 3 * stride];
    uint4 b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    if constexpr (NX == 2) { b0 = p2[i]; b1 = p2[i + stride]; b2 = p2[i + 2 * stride]; b3 = p2[i + 3 * stride]; }
    one(i, a0, b0); one(i + stride, a1, b1); one(i + 2 * stride, a2, b2); one(i + 3 * stride, a3, b3);
  }
  for (; i < n8; i += stride) { const uint4 a0 = p1[i]; uint4 b0 = a0; if constexpr (NX == 2) b0 = p2[i]; one(i, a0, b0); }
}

// dx_b = gamma_b rstd_b (g - w m1 - xhat_b w m2_b) for NB normalisations that share dy and the ReLU mask (bn2 + shortcut BN)
struct BnBwdBranch { const bf16_t* x; const float* mean; const float* rstd; const float* gamma; const float* sums; bf16_t* dx; float* dgamma; float* dbeta; };
template <int NB>
__global__ void __launch_bounds__(256) bn_bwd_apply16_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ relu_src, BnBwdBranch b0,
                                                              BnBwdBranch b1, int P, int c_shift, int hw_shift, float inv_p, RowBound rb) {
  const int C = 1 << c_shift;
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      if (b0.dgamma != nullptr) { b0.dbeta[c] += b0.sums[c]; b0.dgamma[c] += b0.sums[C + c]; }
      if (NB == 2 && b1.dgamma != nullptr) { b1.dbeta[c] += b1.sums[c]; b1.dgamma[c] += b1.sums[C + c]; }
    }
  }
  const int64_t n8 = ((int64_t)rb_rows(rb, P) << c_shift) >> 3;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)((i << 3) & (C - 1));
  const BnBwdBranch br[2] = {b0, b1};
  F8 mean[NB], k0[NB], rs[NB], m2[NB];
  const F8 m1 = ld8f(b0.sums + c);                      // sum g is the same for both branches
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    mean[b] = ld8f(br[b].mean + c); rs[b] = ld8f(br[b].rstd + c); m2[b] = ld8f(br[b].sums + C + c);
    const F8 g = ld8f(br[b].gamma + c);
    k0[b].lo = g.lo * rs[b].lo; k0[b].hi = g.hi * rs[b].hi;
  }
  const int row_shift = c_shift - 3 + hw_shift;         // item index -> glyph index
  const uint4* pg = (const uint4*)dy; const uint4* po = (const uint4*)relu_src;
  auto one = [&](int64_t k, uint4 ug, uint4 uo, const uint4* ux) {
    floatx4 glo, ghi, olo, ohi;
    unpack8(ug, glo, ghi);
    if (relu_src != nullptr) {
      unpack8(uo, olo, ohi);
#pragma unroll
      for (int j = 0; j < 4; ++j) { glo[j] = olo[j] > 0.f ? glo[j] : 0.f; ghi[j] = ohi[j] > 0.f ? ghi[j] : 0.f; }
    }
    const float w = (rb.counts ? rb.counts[k >> row_shift] : 1.0f) * inv_p;
    const floatx4 a_lo = glo - m1.lo * w, a_hi = ghi - m1.hi * w;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      floatx4 xlo, xhi; unpack8(ux[b], xlo, xhi);
      xlo = (xlo - mean[b].lo) * rs[b].lo; xhi = (xhi - mean[b].hi) * rs[b].hi;
      store8<bf16_t>(br[b].dx + (k << 3), k0[b].lo * (a_lo - xlo * (m2[b].lo * w)), k0[b].hi * (a_hi - xhi * (m2[b].hi * w)));
    }
  };
  for (; i + stride < n8; i += 2 * stride) {
    const uint4 g0 = pg[i], g1 = pg[i + stride];
    uint4 o0 = g0, o1 = g1;
    if (relu_src != nullptr) { o0 = po[i]; o1 = po[i + stride]; }
    uint4 x0[NB], x1[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { x0[b] = ((const uint4*)br[b].x)[i]; x1[b] = ((const uint4*)br[b].x)[i + stride]; }
    one(i, g0, o0, x0); one(i + stride, g1, o1, x1);
  }
  for (; i < n8; i += stride) {
    const uint4 g0 = pg[i];
    uint4 o0 = g0;
    if (relu_src != nullptr) o0 = po[i];
    uint4 x0[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) x0[b] = ((const uint4*)br[b].x)[i];
    one(i, g0, o0, x0);
  }
}
static inline int ew16_blocks(int64_t n8) { int64_t b = (n8 + 255) / 256; b = (b + 3) / 4; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }
static inline bool bn16_ok(int P, int C, const RowBound& rb) {
  const int cs = pow2_shift2(C);
  return bn_fast() && cs >= 3 && cs <= 11 && pow2_shift2(rb.hw) >= 0 && (((int64_t)P * C) & 7) == 0;
}
// the two-branch backward map (bn2 + shortcut BN of a BasicBlock): RL_ERR_ARG when the fast path does not apply
int bn_bwd_apply2(hipStream_t st, const bf16_t* dy, const bf16_t* relu_src, const bf16_t* xa, const float* mean_a, const float* rstd_a,
                  const float* gamma_a, bf16_t* dxa, float* dgamma_a, float* dbeta_a, const bf16_t* xb, const float* mean_b, const float* rstd_b,
                  const float* gamma_b, bf16_t* dxb, float* dgamma_b, float* dbeta_b, const float* sums4, int P, int C, RowBound rb, int n_stat) {
  if (!bn16_ok(P, C, rb)) return RL_ERR_ARG;
  const BnBwdBranch b0{xa, mean_a, rstd_a, gamma_a, sums4, dxa, dgamma_a, dbeta_a}, b1{xb, mean_b, rstd_b, gamma_b, sums4 + 2 * C, dxb, dgamma_b, dbeta_b};
  const int64_t n8 = (int64_t)P * C / 8;
  hipLaunchKernelGGL((bn_bwd_apply16_kernel<2>), dim3(ew16_blocks(n8)), dim3(256), 0, st, dy, relu_src, b0, b1, P, pow2_shift2(C), pow2_shift2(rb.hw),
                     1.0f / (float)(n_stat > 0 ? n_stat : P), rb);
  return RL_LAUNCH_CHECK();
}

static inline int ew_blocks(int64_t n4) { int64_t b = (n4 + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }
template <typename T>
int bn_apply(hipStream_t st, const T* x1, const float* sc1, const float* sh1, const T* x2, const float* sc2, const float* sh2,
             T* y, int P, int C, int relu, RowBound rb) {
  if (C & 3) return RL_ERR_ARG;
  const int64_t n = (int64_t)P * C;
  if constexpr (sizeof(T) == 2) {
    if (bn16_ok(P, C, rb)) {
      if (x2 != nullptr) hipLaunchKernelGGL((bn_apply16_kernel<2>), dim3(ew16_blocks(n / 8)), dim3(256), 0, st, x1, sc1, sh1, x2, sc2, sh2, y, P, pow2_shift2(C), relu, rb);
      else hipLaunchKernelGGL((bn_apply16_kernel<1>), dim3(ew16_blocks(n / 8)), dim3(256), 0, st, x1, sc1, sh1, x2, sc2, sh2, y, P, pow2_shift2(C), relu, rb);
      return RL_LAUNCH_CHECK();
    }
  }
  hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(ew_blocks(n / 4)), dim3(256), 0, st, x1, sc1, sh1, x2, sc2, sh2, y, P, C, relu, rb);
  return RL_LAUNCH_CHECK();
}
template int bn_apply<bf16_t>(hipStream_t, const bf16_t*, const float*, const float*, const bf16_t*, const float*, const float*, bf16_t*, int, int, int, RowBound);
template int bn_apply<float>(hipStream_t, const float*, const float*, const float*, const float*, const float*, const float*, float*, int, int, int, RowBound);

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ relu_src, const T* __restrict__ x,
                                    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ sums, int P, int C, float inv_p, T* __restrict__ dx, RowBound rb,
                                    float* dgamma, float* dbeta) {
  if (blockIdx.x == 0 && dgamma != nullptr)            // parameter gradients ride along: dbeta += sum g, dgamma += sum g * xhat
    for (int c = threadIdx.x; c < C; c += blockDim.x) { dbeta[c] += sums[c]; dgamma[c] += sums[C + c]; }
  const int64_t n = (int64_t)rb_rows(rb, P) * C;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    const int c = (int)(i % C);
    const float w = rb_weight(rb, (int)(i / C));
    floatx4 g = load4<T>(dy + i);
    if (relu_src != nullptr) {
      const floatx4 o = load4<T>(relu_src + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
    }
    const floatx4 rs = *(const floatx4*)(rstd + c);
    const floatx4 xh = (load4<T>(x + i) - *(const floatx4*)(mean + c)) * rs;
    const floatx4 m1 = *(const floatx4*)(sums + c) * (inv_p * w), m2 = *(const floatx4*)(sums + C + c) * (inv_p * w);
    store4<T>(dx + i, *(const floatx4*)(gamma + c) * rs * (g - m1 - xh * m2));
  }
}
__global__ void bn_param_grad_kernel(const float* sums, int C, float* dgamma, float* dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { dbeta[c] += sums[c]; dgamma[c] += sums[C + c]; }
}
template <typename T>
int bn_bwd_apply(hipStream_t st, const T* dy, const T* relu_src, const T* x, const float* mean, const float* rstd,
                 const float* gamma, const float* sums, int P, int C, T* dx, float* dgamma, float* dbeta, RowBound rb, int n_stat) {
  if (C & 3) return RL_ERR_ARG;
  const int64_t n = (int64_t)P * C;
  if constexpr (sizeof(T) == 2) {
    if (bn16_ok(P, C, rb)) {
      const BnBwdBranch b0{x, mean, rstd, gamma, sums, dx, dgamma, dbeta};
      hipLaunchKernelGGL((bn_bwd_apply16_kernel<1>), dim3(ew16_blocks(n / 8)), dim3(256), 0, st, dy, relu_src, b0, b0, P, pow2_shift2(C), pow2_shift2(rb.hw),
                         1.0f / (float)(n_stat > 0 ? n_stat : P), rb);
      return RL_LAUNCH_CHECK();
    }
  }
  hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(ew_blocks(n / 4)), dim3(256), 0, st, dy, relu_src, x, mean, rstd, gamma,
                     sums, P, C, 1.0f / (float)(n_stat > 0 ? n_stat : P), dx, rb, dgamma, dbeta);
  return RL_LAUNCH_CHECK();
}
template int bn_bwd_apply<bf16_t>(hipStream_t, const bf16_t*, const bf16_t*, const bf16_t*, const float*, const float*, const float*, const float*, int, int, bf16_t*, float*, float*, RowBound, int);
template int bn_bwd_apply<float>(hipStream_t, const float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, float*, float*, float*, RowBound, int);

template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ src, T* __restrict__ g, int64_t n) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    floatx4 d = load4<T>(dy + i);
    const floatx4 o = load4<T>(src + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = o[j] > 0.f ? d[j] : 0.f;
    store4<T>(g + i, d);
  }
}
template <typename T> int relu_bwd(hipStream_t st, const T* dy, const T* relu_src, T* g, int64_t n) {
  if (n & 3) return RL_ERR_ARG;
  hipLaunchKernelGGL((relu_bwd_kernel<T>), dim3(ew_blocks(n / 4)), dim3(256), 0, st, dy, relu_src, g, n);
  return RL_LAUNCH_CHECK();
}
template int relu_bwd<bf16_t>(hipStream_t, const bf16_t*, const bf16_t*, bf16_t*, int64_t);
template int relu_bwd<float>(hipStream_t, const float*, const float*, float*, int64_t);

// ---------------------------------------------------------------------------------------------
// Operand shadows
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_copy_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = from_f<T>(src[i]);
}
template <typename T> int cast_copy(hipStream_t st, const float* src, T* dst, int64_t n) {
  if (n <= 0) return RL_OK;
  hipLaunchKernelGGL((cast_copy_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, st, src, dst, n);
  return RL_LAUNCH_CHECK();
}
template int cast_copy<bf16_t>(hipStream_t, const float*, bf16_t*, int64_t);
template int cast_copy<float>(hipStream_t, const float*, float*, int64_t);

// T -> fp32 (the reference returns fp32 logits, models.py:859; the bf16 engine widens them on request), 8 elements per lane
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    floatx4 a, b;
    load8<T>(src + i * 8, a, b);
    *(floatx4*)(dst + i * 8) = a;
    *(floatx4*)(dst + i * 8 + 4) = b;
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n8 << 3) + threadIdx.x; i < n; i += blockDim.x) dst[i] = to_f<T>(src[i]);
}
template <typename T> int cast_to_f32(hipStream_t st, const T* src, float* dst, int64_t n) {
  if (n <= 0) return RL_OK;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return RL_ERR_ARG;
  hipLaunchKernelGGL((cast_to_f32_kernel<T>), dim3(ew_blocks(n / 8)), dim3(256), 0, st, src, dst, n);
  return RL_LAUNCH_CHECK();
}
template int cast_to_f32<bf16_t>(hipStream_t, const bf16_t*, float*, int64_t);
template int cast_to_f32<float>(hipStream_t, const float*, float*, int64_t);

template <typename T>
__global__ void __launch_bounds__(256)
cast_transpose_kernel(const float* __restrict__ src, int R, int C, T* __restrict__ dst, T* __restrict__ dstT) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) {
      v = src[(int64_t)r * C + c];
      if (dst != nullptr) dst[(int64_t)r * C + c] = from_f<T>(v);
    }
    tile[ty + 8 * k][tx] = v;
  }
  __syncthreads();
  if (dstT != nullptr) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ty + 8 * k, r = r0 + tx;
      if (r < R && c < C) dstT[(int64_t)c * R + r] = from_f<T>(tile[tx][ty + 8 * k]);
    }
  }
}
template <typename T> int cast_transpose(hipStream_t st, const float* src, int R, int C, T* dst, T* dstT) {
  hipLaunchKernelGGL((cast_transpose_kernel<T>), dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, st, src, R, C, dst, dstT);
  return RL_LAUNCH_CHECK();
}
// 64 x 64 tiles, 16 B fp32 reads / 4-element writes; the transposed copy goes through a padded LDS tile so both outputs
// are written in 128-byte (bf16) row segments.
template <typename T>
__global__ void __launch_bounds__(256) cast_transpose_multi_kernel(const CastDesc* __restrict__ descs, int n) {
  __shared__ float tile[64][65];
  int lo = 0, hi = n - 1;                               // last descriptor with tile_begin <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const CastDesc d = descs[lo];
  const int t = blockIdx.x - d.tile_begin;
  const int r0 = (t / d.tiles_c) * 64, c0 = (t % d.tiles_c) * 64;
  const int q = threadIdx.x & 15, y = threadIdx.x >> 4;     // 16 quads x 16 rows per pass
  T* dst = (T*)d.dst;
  T* dstT = (T*)d.dstT;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + y + 16 * k, c = c0 + 4 * q;
    floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
    if (r < d.R && c < d.C) {                           // C % 4 == 0
      v = *(const floatx4*)(d.src + (int64_t)r * d.C + c);
      if (dst != nullptr) store4<T>(dst + (int64_t)r * d.C + c, v);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[y + 16 * k][4 * q + j] = v[j];
  }
  __syncthreads();
  if (dstT == nullptr) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + y + 16 * k, r = r0 + 4 * q;
    if (c < d.C && r < d.R) {
      if (r + 3 < d.R && (d.R & 3) == 0) {
        floatx4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = tile[4 * q + j][y + 16 * k];
        store4<T>(dstT + (int64_t)c * d.ldT + r, v);
      } else {
        for (int j = 0; j < 4 && r + j < d.R; ++j) dstT[(int64_t)c * d.ldT + r + j] = from_f<T>(tile[4 * q + j][y + 16 * k]);
      }
    }
  }
}
template <typename T> int cast_transpose_multi(hipStream_t st, const CastDesc* descs, int n, int total_tiles) {
  if (n <= 0 || total_tiles <= 0) return RL_OK;
  hipLaunchKernelGGL((cast_transpose_multi_kernel<T>), dim3(total_tiles), dim3(256), 0, st, descs, n);
  return RL_LAUNCH_CHECK();
}
template int cast_transpose_multi<bf16_t>(hipStream_t, const CastDesc*, int, int);
template int cast_transpose_multi<float>(hipStream_t, const CastDesc*, int, int);
template int cast_transpose<bf16_t>(hipStream_t, const float*, int, int, bf16_t*, bf16_t*);
template int cast_transpose<float>(hipStream_t, const float*, int, int, float*, float*);

template <typename T>
__global__ void conv_weight_shadow_kernel(const float* __restrict__ w, int Co, int Ci, int KHW, int Cpad, int CiRows,
                                          T* __restrict__ fwd, T* __restrict__ dgrad, TapOrder order) {
  const int64_t nf = (int64_t)Co * KHW * Cpad, nd = (int64_t)CiRows * KHW * Co;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nf) {
      if (fwd == nullptr) continue;
      const int ci = (int)(i % Cpad);
      const int tap = (int)((i / Cpad) % KHW);
      const int co = (int)(i / ((int64_t)Cpad * KHW));
      fwd[i] = from_f<T>(ci < Ci ? w[((int64_t)co * Ci + ci) * KHW + tap] : 0.f);
    } else {
      if (dgrad == nullptr) continue;
      const int64_t k = i - nf;
      const int co = (int)(k % Co);
      const int tap = (int)((k / Co) % KHW);
      const int ci = (int)(k / ((int64_t)Co * KHW));
      const int src_tap = order.n ? order.t[tap] : tap;          // slot `tap` of the data-gradient copy holds original tap order.t[tap]
      dgrad[k] = from_f<T>(ci < Ci ? w[((int64_t)co * Ci + ci) * KHW + src_tap] : 0.f);
    }
  }
}
template <typename T>
int conv_weight_shadow(hipStream_t st, const float* w, int Co, int Ci, int KHW, int Cpad, int CiRows, T* fwd, T* dgrad, const TapOrder& order) {
  if (order.n != 0 && order.n != KHW) return RL_ERR_ARG;
  const int64_t n = (int64_t)Co * KHW * Cpad + (int64_t)CiRows * KHW * Co;
  hipLaunchKernelGGL((conv_weight_shadow_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, st, w, Co, Ci, KHW, Cpad, CiRows, fwd, dgrad, order);
  return RL_LAUNCH_CHECK();
}
template int conv_weight_shadow<bf16_t>(hipStream_t, const float*, int, int, int, int, int, bf16_t*, bf16_t*, const TapOrder&);
template int conv_weight_shadow<float>(hipStream_t, const float*, int, int, int, int, int, float*, float*, const TapOrder&);

// All conv-weight operand copies of the glyph ResNet in ONE launch (was 20 launches of ~13 us on the glyph stream).  A workgroup
// takes a [32 co][32 ci][KHW] block of one weight tensor: the source rows w[co][ci0 .. ci0 + 31][:] are contiguous runs of 32 * KHW
// floats (coalesced reads), the block goes through LDS, and both copies leave in 64-byte segments - fwd[co][tap][ci0 ..] and
// dgrad[ci][slot][co0 ..].  (The element-per-thread form gathered every source float with a 36-byte stride: 167 us for 14 M weights.)
// Padding entries (ci >= Ci of a Cpad-wide row, rows >= Ci of the dgrad copy) are never written: the shadow arena is zero-filled.
constexpr int CSH_T = 32;
// (KHW and, on full tiles, the tile edges are compile-time constants: the index arithmetic is shifts and multiply-shifts - with
// run-time divisors the kernel was bound by its integer divisions, 123 us)
template <typename T, int KHW, bool FULL>
__device__ __forceinline__ void conv_shadow_tile(const ConvShadowDesc& d, float (*tile)[CSH_T * 9 + 1], int co0, int ci0, int nco_, int nci_) {
  const int nco = FULL ? CSH_T : nco_, nci = FULL ? CSH_T : nci_, run = nci * KHW;
  const int Co = d.Co, Ci = d.Ci, Cpad = d.Cpad;
  T* fwd = (T*)d.fwd;
  T* dgrad = (T*)d.dgrad;
  for (int e = threadIdx.x; e < nco * run; e += 256) {                    // [co][ci][tap] block -> LDS, the same order
    const int c = e / run, r = e - c * run;
    tile[c][r] = d.w[((int64_t)(co0 + c) * Ci + ci0) * KHW + r];
  }
  __syncthreads();
  if (fwd != nullptr)
    for (int e = threadIdx.x; e < nco * KHW * nci; e += 256) {            // fwd[co][tap][ci]: ci fastest
      const int ci = e % nci, rest = e / nci, tap = rest % KHW, c = rest / KHW;
      fwd[((int64_t)(co0 + c) * KHW + tap) * Cpad + ci0 + ci] = from_f<T>(tile[c][ci * KHW + tap]);
    }
  if (dgrad != nullptr)
    for (int e = threadIdx.x; e < nci * KHW * nco; e += 256) {            // dgrad[ci][slot][co]: co fastest
      const int c = e % nco, rest = e / nco, slot = rest % KHW, ci = rest / KHW;
      const int src_tap = d.order.n ? d.order.t[slot] : slot;
      dgrad[((int64_t)(ci0 + ci) * KHW + slot) * Co + co0 + c] = from_f<T>(tile[c][ci * KHW + src_tap]);
    }
}
template <typename T>
__global__ void __launch_bounds__(256) conv_weight_shadow_multi_kernel(ConvShadowDescs ds) {
  __shared__ float tile[CSH_T][CSH_T * 9 + 1];
  int k = 0;
#pragma unroll 1
  for (int i = 1; i < ds.n; ++i) if ((int)blockIdx.x >= ds.d[i].block_begin) k = i;
  const ConvShadowDesc& d = ds.d[k];
  const int tiles_ci = (d.Ci + CSH_T - 1) / CSH_T;
  const int b = (int)blockIdx.x - d.block_begin;
  const int co0 = (b / tiles_ci) * CSH_T, ci0 = (b % tiles_ci) * CSH_T;
  const int nco = min(CSH_T, d.Co - co0), nci = min(CSH_T, d.Ci - ci0);
  const bool full = nco == CSH_T && nci == CSH_T;
  if (d.KHW == 9) { if (full) conv_shadow_tile<T, 9, true>(d, tile, co0, ci0, nco, nci); else conv_shadow_tile<T, 9, false>(d, tile, co0, ci0, nco, nci); }
  else if (d.KHW == 1) { if (full) conv_shadow_tile<T, 1, true>(d, tile, co0, ci0, nco, nci); else conv_shadow_tile<T, 1, false>(d, tile, co0, ci0, nco, nci); }
}
template <typename T> int conv_weight_shadow_multi(hipStream_t st, ConvShadowDescs& ds) {
  if (ds.n < 1 || ds.n > CONV_SHADOW_MAX) return RL_ERR_ARG;
  int blocks = 0;
  for (int k = 0; k < ds.n; ++k) {
    const ConvShadowDesc& d = ds.d[k];
    if ((d.order.n != 0 && d.order.n != d.KHW) || (d.KHW != 1 && d.KHW != 9) || d.Ci > d.Cpad || d.Ci > d.CiRows) return RL_ERR_ARG;
    ds.d[k].block_begin = blocks;
    blocks += ((d.Co + CSH_T - 1) / CSH_T) * ((d.Ci + CSH_T - 1) / CSH_T);
  }
  hipLaunchKernelGGL((conv_weight_shadow_multi_kernel<T>), dim3(blocks), dim3(256), 0, st, ds);
  return RL_LAUNCH_CHECK();
}
template int conv_weight_shadow_multi<bf16_t>(hipStream_t, ConvShadowDescs&);
template int conv_weight_shadow_multi<float>(hipStream_t, ConvShadowDescs&);

template <typename T>
__global__ void glyph_shadow_kernel(const float* __restrict__ tbl, int64_t V, int F, int HW, int Cpad, T* __restrict__ out) {
  const int64_t n = V * HW * Cpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % Cpad);
    const int64_t px = (i / Cpad) % HW;
    const int64_t v = i / ((int64_t)Cpad * HW);
    out[i] = from_f<T>(f < F ? tbl[(v * F + f) * HW + px] : 0.f);
  }
}
template <typename T> int glyph_shadow(hipStream_t st, const float* tbl, int V, int F, int HW, int Cpad, T* out) {
  hipLaunchKernelGGL((glyph_shadow_kernel<T>), dim3(4096), dim3(256), 0, st, tbl, (int64_t)V, F, HW, Cpad, out);
  return RL_LAUNCH_CHECK();
}
template int glyph_shadow<bf16_t>(hipStream_t, const float*, int, int, int, int, bf16_t*);
template int glyph_shadow<float>(hipStream_t, const float*, int, int, int, int, float*);

// ---------------------------------------------------------------------------------------------
// Optimizer: global grad norm + AdamW over flat arenas
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ g, int64_t n, float* out) {
  __shared__ float sm[4];
  float s = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const floatx4 v = *(const floatx4*)(g + i * 4);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = n4 * 4; i < n; ++i) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, sm[0] + sm[1] + sm[2] + sm[3]);
}
int sumsq_accum(hipStream_t st, const float* g, int64_t n, float* out) {
  if (n <= 0) return RL_OK;
  hipLaunchKernelGGL(sumsq_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, st, g, n, out);
  return RL_LAUNCH_CHECK();
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             int64_t n, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                             const float* __restrict__ norm_sq, float max_norm) {
  float clip = 1.0f;
  if (norm_sq != nullptr) {
    const float c = max_norm / (sqrtf(norm_sq[0]) + 1e-6f);
    clip = c < 1.0f ? c : 1.0f;
  }
  const float step_size = lr * sqrtf(bc2) / bc1;     // optimization.py:156-160
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * clip;
    const float mi = m[i] * beta1 + (1.0f - beta1) * gi;
    const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
    float pi = p[i] - step_size * (mi / (sqrtf(vi) + eps));
    if (wd > 0.f) pi -= lr * wd * pi;                 // decoupled decay AFTER the update (:162-167)
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
}
int adamw_flat(hipStream_t st, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
               float eps, float weight_decay, float bias_c1, float bias_c2, const float* norm_sq, float max_norm) {
  if (n <= 0) return RL_OK;
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                     bias_c1, bias_c2, norm_sq, max_norm);
  return RL_LAUNCH_CHECK();
}

// AdamW with per-group hyper-parameters in ONE launch over the arena (the reference's decay / no-decay parameter groups,
// run.py:146-151, with a non-zero weight decay): group_of_block[i >> 6] names the group of elements [64 b, 64 b + 64) (every tensor
// of the arena starts on a 64-element boundary), 255 = not optimised.
__global__ void adamw_grouped_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                     int64_t n, const uint8_t* __restrict__ group_of_block, AdamwGroups gs,
                                     const float* __restrict__ norm_sq, float max_norm, const uint8_t* __restrict__ skip_block) {
  float clip = 1.0f;
  if (norm_sq != nullptr) {
    const float c = max_norm / (sqrtf(norm_sq[0]) + 1e-6f);
    clip = c < 1.0f ? c : 1.0f;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int gid = group_of_block[i >> 6];
    if (gid >= gs.n) continue;
    if (skip_block != nullptr && skip_block[i >> 6]) continue;      // stepped by adamw_cast_multi_kernel (the Linear weights)
    const AdamwGroup h = gs.g[gid];
    const float gi = g[i] * clip;
    const float mi = m[i] * h.beta1 + (1.0f - h.beta1) * gi;
    const float vi = v[i] * h.beta2 + (1.0f - h.beta2) * gi * gi;
    float pi = p[i] - h.step_size * (mi / (sqrtf(vi) + h.eps));
    if (h.weight_decay > 0.f) pi -= h.lr * h.weight_decay * pi;
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
}
int adamw_grouped(hipStream_t st, float* p, const float* g, float* m, float* v, int64_t n, const uint8_t* group_of_block,
                  const AdamwGroups& gs, const float* norm_sq, float max_norm, const uint8_t* skip_block) {
  if (n <= 0) return RL_OK;
  if (gs.n < 1 || gs.n > ADAMW_MAX_GROUPS || group_of_block == nullptr) return RL_ERR_ARG;
  hipLaunchKernelGGL(adamw_grouped_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, p, g, m, v, n, group_of_block, gs, norm_sq, max_norm, skip_block);
  return RL_LAUNCH_CHECK();
}

// the grouped sweep over a chunk list (block b steps chunk b: [off, off + len), len <= 64 Ki floats, 64-element aligned): what the
// tiled launches below do not own
__global__ void __launch_bounds__(256) adamw_chunks_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                           const FillChunk* __restrict__ chunks, const uint8_t* __restrict__ group_of_block, AdamwGroups gs,
                                                           const float* __restrict__ norm_sq, float max_norm) {
  float clip = 1.0f;
  if (norm_sq != nullptr) {
    const float c = max_norm / (sqrtf(norm_sq[0]) + 1e-6f);
    clip = c < 1.0f ? c : 1.0f;
  }
  const FillChunk ch = chunks[blockIdx.x];
  if (threadIdx.x == 0)
    for (int k = ch.len & ~3; k < ch.len; ++k) {          // a tensor whose length is not a multiple of 4 ends the arena
      const int64_t i = ch.off + k;
      const int gid = group_of_block[i >> 6];
      if (gid >= gs.n) continue;
      const AdamwGroup h = gs.g[gid];
      const float gi = g[i] * clip;
      const float mi = m[i] * h.beta1 + (1.0f - h.beta1) * gi;
      const float vi = v[i] * h.beta2 + (1.0f - h.beta2) * gi * gi;
      float pi = p[i] - h.step_size * (mi / (sqrtf(vi) + h.eps));
      if (h.weight_decay > 0.f) pi -= h.lr * h.weight_decay * pi;
      m[i] = mi; v[i] = vi; p[i] = pi;
    }
  for (int k = threadIdx.x * 4; k + 3 < ch.len; k += 1024) {
    const int64_t i = ch.off + k;
    const int gid = group_of_block[i >> 6];
    if (gid >= gs.n) continue;
    const AdamwGroup h = gs.g[gid];
    floatx4 pv = *(const floatx4*)(p + i), mv = *(const floatx4*)(m + i), vv = *(const floatx4*)(v + i);
    const floatx4 gv = *(const floatx4*)(g + i) * clip;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mv[j] = mv[j] * h.beta1 + (1.0f - h.beta1) * gv[j];
      vv[j] = vv[j] * h.beta2 + (1.0f - h.beta2) * gv[j] * gv[j];
      float pi = pv[j] - h.step_size * (mv[j] / (sqrtf(vv[j]) + h.eps));
      if (h.weight_decay > 0.f) pi -= h.lr * h.weight_decay * pi;
      pv[j] = pi;
    }
    *(floatx4*)(m + i) = mv; *(floatx4*)(v + i) = vv; *(floatx4*)(p + i) = pv;
  }
}
int adamw_chunks(hipStream_t st, float* p, const float* g, float* m, float* v, const FillChunk* chunks_dev, int n_chunks,
                 const uint8_t* group_of_block, const AdamwGroups& gs, const float* norm_sq, float max_norm) {
  if (n_chunks <= 0) return RL_OK;
  if (gs.n < 1 || gs.n > ADAMW_MAX_GROUPS || group_of_block == nullptr) return RL_ERR_ARG;
  hipLaunchKernelGGL(adamw_chunks_kernel, dim3(n_chunks), dim3(256), 0, st, p, g, m, v, chunks_dev, group_of_block, gs, norm_sq, max_norm);
  return RL_LAUNCH_CHECK();
}

// AdamW over the Linear weights in the 64 x 64 tiles of cast_transpose_multi: the pass that reads and writes every parameter anyway
// also emits the compute-dtype operand copies W and W^T the next forward / backward multiply with (the transposed copy through a
// padded LDS tile, 128-byte row segments on both sides) - the separate refresh (8 B / parameter of reads + writes, two launches)
// disappears from a FusedAdamW step.  Element (r, c) of descriptor d sits at arena offset (d.src - P0) + r * C + c in p / g / m / v.
template <typename T>
__global__ void __launch_bounds__(256) adamw_cast_multi_kernel(const CastDesc* __restrict__ descs, int n, float* __restrict__ P0,
                                                               const float* __restrict__ G0, float* __restrict__ M0, float* __restrict__ V0,
                                                               const uint8_t* __restrict__ group_of_block, AdamwGroups gs,
                                                               const float* __restrict__ norm_sq, float max_norm, int tile_base) {
  __shared__ float tile[64][65];
  // tile_base (round 6): the launch covers a SLICE of a descriptor group - descs points at the slice's first descriptor, whose
  // tile_begin (counted from the group's first) is tile_base
  const int bid = (int)blockIdx.x + tile_base;
  int lo = 0, hi = n - 1;                               // last descriptor with tile_begin <= bid
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_begin <= bid) lo = mid; else hi = mid - 1;
  }
  const CastDesc d = descs[lo];
  const int t = bid - d.tile_begin;
  const int r0 = (t / d.tiles_c) * 64, c0 = (t % d.tiles_c) * 64;
  const int q = threadIdx.x & 15, y = threadIdx.x >> 4;     // 16 quads x 16 rows per pass
  T* dst = (T*)d.dst;
  T* dstT = (T*)d.dstT;
  const int64_t base = d.src - P0;
  float clip = 1.0f;
  if (norm_sq != nullptr) {
    const float c = max_norm / (sqrtf(norm_sq[0]) + 1e-6f);
    clip = c < 1.0f ? c : 1.0f;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + y + 16 * k, c = c0 + 4 * q;
    floatx4 pv = floatx4{0.f, 0.f, 0.f, 0.f};
    if (r < d.R && c < d.C) {                           // C % 4 == 0
      const int64_t e = base + (int64_t)r * d.C + c;
      pv = *(const floatx4*)(P0 + e);
      const int gid = group_of_block[e >> 6];           // a quad never straddles a 64-element block (base % 64 == 0, C % 4 == 0)
      if (gid < gs.n) {
        const AdamwGroup h = gs.g[gid];
        const floatx4 gv = *(const floatx4*)(G0 + e) * clip;
        floatx4 mv = *(const floatx4*)(M0 + e), vv = *(const floatx4*)(V0 + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          mv[j] = mv[j] * h.beta1 + (1.0f - h.beta1) * gv[j];
          vv[j] = vv[j] * h.beta2 + (1.0f - h.beta2) * gv[j] * gv[j];
          float pi = pv[j] - h.step_size * (mv[j] / (sqrtf(vv[j]) + h.eps));
          if (h.weight_decay > 0.f) pi -= h.lr * h.weight_decay * pi;
          pv[j] = pi;
        }
        *(floatx4*)(M0 + e) = mv; *(floatx4*)(V0 + e) = vv; *(floatx4*)(P0 + e) = pv;
      }
      if (dst != nullptr) store4<T>(dst + (int64_t)r * d.C + c, pv);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[y + 16 * k][4 * q + j] = pv[j];
  }
  __syncthreads();
  if (dstT == nullptr) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + y + 16 * k, r = r0 + 4 * q;
    if (c < d.C && r < d.R) {
      if (r + 3 < d.R && (d.R & 3) == 0) {
        floatx4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = tile[4 * q + j][y + 16 * k];
        store4<T>(dstT + (int64_t)c * d.ldT + r, v);
      } else {
        for (int j = 0; j < 4 && r + j < d.R; ++j) dstT[(int64_t)c * d.ldT + r + j] = from_f<T>(tile[4 * q + j][y + 16 * k]);
      }
    }
  }
}
// The same tile without LDS (round 6): a thread owns a 4 x 4 block - rows r0 + 4 y .. + 3, columns c0 + 4 q .. + 3 - so the transposed
// copy is an in-register transpose (four 8-byte stores of four consecutive W^T columns each), all sixteen 16-byte loads of the thread
// are in flight together, there is no barrier, and - the reason it was written - a workgroup that needs no LDS can share a CU with the
// two 80 KB GEMM workgroups of a running forward (the pipelined sweep, engine.hip adamw_step).  Same arithmetic per element: same bits.
template <typename T>
__global__ void __launch_bounds__(256) adamw_cast_multi_reg_kernel(const CastDesc* __restrict__ descs, int n, float* __restrict__ P0,
                                                                   const float* __restrict__ G0, float* __restrict__ M0, float* __restrict__ V0,
                                                                   const uint8_t* __restrict__ group_of_block, AdamwGroups gs,
                                                                   const float* __restrict__ norm_sq, float max_norm, int tile_base) {
  const int bid = (int)blockIdx.x + tile_base;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_begin <= bid) lo = mid; else hi = mid - 1;
  }
  const CastDesc d = descs[lo];
  const int t = bid - d.tile_begin;
  const int r0 = (t / d.tiles_c) * 64, c0 = (t % d.tiles_c) * 64;
  const int q = threadIdx.x & 15, y = threadIdx.x >> 4;
  T* dst = (T*)d.dst;
  T* dstT = (T*)d.dstT;
  const int64_t base = d.src - P0;
  float clip = 1.0f;
  if (norm_sq != nullptr) {
    const float c = max_norm / (sqrtf(norm_sq[0]) + 1e-6f);
    clip = c < 1.0f ? c : 1.0f;
  }
  const int c = c0 + 4 * q, rb = r0 + 4 * y;
  if (c >= d.C || rb >= d.R) return;
  floatx4 pv[4], gv[4], mv[4], vv[4];
  int gid[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = rb + k;
    pv[k] = floatx4{0.f, 0.f, 0.f, 0.f}; gid[k] = 255;
    if (r < d.R) {
      const int64_t e = base + (int64_t)r * d.C + c;
      pv[k] = *(const floatx4*)(P0 + e);
      gid[k] = group_of_block[e >> 6];
      if (gid[k] < gs.n) { gv[k] = *(const floatx4*)(G0 + e); mv[k] = *(const floatx4*)(M0 + e); vv[k] = *(const floatx4*)(V0 + e); }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = rb + k;
    if (r >= d.R) continue;
    const int64_t e = base + (int64_t)r * d.C + c;
    if (gid[k] < gs.n) {
      const AdamwGroup h = gs.g[gid[k]];
      const floatx4 g4 = gv[k] * clip;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mv[k][j] = mv[k][j] * h.beta1 + (1.0f - h.beta1) * g4[j];
        vv[k][j] = vv[k][j] * h.beta2 + (1.0f - h.beta2) * g4[j] * g4[j];
        float pi = pv[k][j] - h.step_size * (mv[k][j] / (sqrtf(vv[k][j]) + h.eps));
        if (h.weight_decay > 0.f) pi -= h.lr * h.weight_decay * pi;
        pv[k][j] = pi;
      }
      *(floatx4*)(M0 + e) = mv[k]; *(floatx4*)(V0 + e) = vv[k]; *(floatx4*)(P0 + e) = pv[k];
    }
    if (dst != nullptr) store4<T>(dst + (int64_t)r * d.C + c, pv[k]);
  }
  if (dstT == nullptr) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const floatx4 v = floatx4{pv[0][j], pv[1][j], pv[2][j], pv[3][j]};
    if (rb + 3 < d.R && (d.ldT & 3) == 0) store4<T>(dstT + (int64_t)(c + j) * d.ldT + rb, v);
    else for (int k = 0; k < 4 && rb + k < d.R; ++k) dstT[(int64_t)(c + j) * d.ldT + rb + k] = from_f<T>(v[k]);
  }
}
// realise_set_ln key 6: 1 = the register-transpose tile above, 0 (default) = the LDS tile (round 4).  Measured level (14.67 against 14.61-
// 14.83 ms/step, three pairs), and the pipelined sweep on it is level with the plain sweep too: what overlaps competes for HBM.
static int g_adamw_reg = 0;
void set_adamw_reg(int on) { g_adamw_reg = on; }

template <typename T>
int adamw_cast_multi(hipStream_t st, const CastDesc* descs, int n, int total_tiles, float* P0, const float* G0, float* M0, float* V0,
                     const uint8_t* group_of_block, const AdamwGroups& gs, const float* norm_sq, float max_norm, int tile_base) {
  if (n <= 0 || total_tiles <= 0) return RL_OK;
  if (gs.n < 1 || gs.n > ADAMW_MAX_GROUPS || group_of_block == nullptr) return RL_ERR_ARG;
  if (g_adamw_reg) hipLaunchKernelGGL((adamw_cast_multi_reg_kernel<T>), dim3(total_tiles), dim3(256), 0, st, descs, n, P0, G0, M0, V0, group_of_block, gs, norm_sq, max_norm, tile_base);
  else hipLaunchKernelGGL((adamw_cast_multi_kernel<T>), dim3(total_tiles), dim3(256), 0, st, descs, n, P0, G0, M0, V0, group_of_block, gs, norm_sq, max_norm, tile_base);
  return RL_LAUNCH_CHECK();
}
template int adamw_cast_multi<bf16_t>(hipStream_t, const CastDesc*, int, int, float*, const float*, float*, float*, const uint8_t*, const AdamwGroups&, const float*, float, int);
template int adamw_cast_multi<float>(hipStream_t, const CastDesc*, int, int, float*, const float*, float*, float*, const uint8_t*, const AdamwGroups&, const float*, float, int);

// g *= min(1, max_norm / (sqrt(*norm_sq) + 1e-6)) - the in-place half of clip_grad_norm_ (run.py:207) for callers that step with a
// stock optimizer (FusedAdamW applies the coefficient inside its own sweep instead)
__global__ void clip_scale_kernel(float* __restrict__ g, int64_t n, const float* __restrict__ norm_sq, float max_norm) {
  const float c = max_norm / (sqrtf(norm_sq[0]) + 1e-6f);
  if (c >= 1.0f) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) g[i] *= c;
}
int clip_scale(hipStream_t st, float* g, int64_t n, const float* norm_sq, float max_norm) {
  if (n <= 0) return RL_OK;
  hipLaunchKernelGGL(clip_scale_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, g, n, norm_sq, max_norm);
  return RL_LAUNCH_CHECK();
}

// zero several ranges of one arena in ONE launch: block b clears chunk b = [off[b], off[b] + len[b]) (chunks <= 1 Mi floats)
__global__ void __launch_bounds__(256) zero_chunks_kernel(float* __restrict__ base, const FillChunk* __restrict__ chunks) {
  const FillChunk c = chunks[blockIdx.x];
  float* p = base + c.off;
  const int n4 = c.len >> 2;
  for (int i = threadIdx.x; i < n4; i += 256) ((floatx4*)p)[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int i = (n4 << 2) + threadIdx.x; i < c.len; i += 256) p[i] = 0.f;
}
int zero_chunks(hipStream_t st, float* base, const FillChunk* chunks_dev, int n) {
  if (n <= 0) return RL_OK;
  hipLaunchKernelGGL(zero_chunks_kernel, dim3(n), dim3(256), 0, st, base, chunks_dev);
  return RL_LAUNCH_CHECK();
}

__global__ void fill_kernel(float* p, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
int fill_f32(hipStream_t st, float* p, float v, int64_t n) {
  if (n <= 0) return RL_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, p, v, n);
  return RL_LAUNCH_CHECK();
}
__global__ void add_i64_kernel(int64_t* p, int64_t v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += v;
}
int add_i64(hipStream_t st, int64_t* p, int64_t v, int n) {
  hipLaunchKernelGGL(add_i64_kernel, dim3((n + 63) / 64), dim3(64), 0, st, p, v, n);
  return RL_LAUNCH_CHECK();
}

}  // namespace rl
