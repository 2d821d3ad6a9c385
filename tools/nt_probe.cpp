// Bottleneck probe of the NT GEMM kernel on the shapes of the BERT stacks (no torch: starts in seconds).
//   hipcc --offload-arch=gfx950 -O2 tools/nt_probe.cpp -o gpurun_out/nt_probe -ldl && gpurun_out/nt_probe realise_amd/librealise_hip.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/realise_hip.h"

typedef int (*gemm_nt_fn)(void*, int, const void*, int64_t, const void*, int64_t, int, int, int, const realise_epilogue*);
typedef void (*seti_fn)(int);

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "realise_amd/librealise_hip.so";
  void* h = dlopen(path, RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  gemm_nt_fn gemm = (gemm_nt_fn)dlsym(h, "realise_gemm_nt");
  seti_fn probe = (seti_fn)dlsym(h, "realise_set_nt_probe");
  seti_fn n96 = (seti_fn)dlsym(h, "realise_set_nt_allow_n96");
  seti_fn variant = (seti_fn)dlsym(h, "realise_set_nt_variant");
  if (!gemm || !probe) { fprintf(stderr, "missing symbols\n"); return 1; }
  struct Shape { int M, N, K; const char* what; };
  const Shape shapes[] = {{8192, 768, 64, "one K-tile"}, {8192, 768, 128, "two K-tiles"}, {8192, 3072, 64, "one K-tile"}, {8192, 768, 768, "attn-out / dgrad"}, {8192, 2304, 768, "qkv"}, {8192, 3072, 768, "ffn1 / ffn2-dgrad"},
                          {8192, 768, 3072, "ffn2 / ffn1-dgrad"}, {8192, 768, 2304, "qkv-dgrad"}, {8192, 21128, 768, "classifier"}};
  size_t maxA = (size_t)8192 * 3072, maxB = (size_t)21128 * 768, maxC = (size_t)8192 * 21128;
  uint16_t *A, *B, *C;
  hipMalloc(&A, maxA * 2); hipMalloc(&B, maxB * 2); hipMalloc(&C, maxC * 2);
  {   // uniform random bf16 in [-1, 1): zero operands would run at a higher clock
    std::vector<uint16_t> hbuf(maxB > maxA ? maxB : maxA);
    uint32_t s = 12345;
    for (auto& v : hbuf) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) / 8388608.0f) - 1.0f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    hipMemcpy(A, hbuf.data(), maxA * 2, hipMemcpyHostToDevice); hipMemcpy(B, hbuf.data(), maxB * 2, hipMemcpyHostToDevice);
  }
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  seti_fn wide = (seti_fn)dlsym(h, "realise_set_nt_wide_epilogue");
  uint16_t *C2, *AUX; float* bias;
  hipMalloc(&C2, (size_t)8192 * 3072 * 2); hipMalloc(&AUX, (size_t)8192 * 3072 * 2); hipMalloc(&bias, 21128 * 4);
  hipMemcpy(AUX, A, (size_t)8192 * 3072 * 2, hipMemcpyDeviceToDevice); hipMemset(bias, 0, 21128 * 4);
  auto run = [&](const Shape& sh, int epi) {
    realise_epilogue ep; memset(&ep, 0, sizeof(ep));
    ep.mode = epi; ep.out = C; ep.ldo = sh.N; ep.alpha = 1.0f; ep.drop_scale = 1.0f;
    if (epi != 0) ep.bias = bias;
    if (epi == 1) ep.out2 = C2;
    if (epi == 2) { ep.aux = AUX; ep.ldaux = sh.N; ep.drop_seed = 77; ep.drop_thresh = 429496730u; ep.drop_scale = 1.0f / 0.9f; }
    if (epi == 4) { ep.aux = AUX; ep.ldaux = sh.N; ep.bias = nullptr; }
    for (int i = 0; i < 3; ++i) gemm(st, 1, A, sh.K, B, sh.K, sh.M, sh.N, sh.K, &ep);
    const int reps = 20;
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) gemm(st, 1, A, sh.K, B, sh.K, sh.M, sh.N, sh.K, &ep);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / reps;
  };
  const int nvar = variant ? (argc > 2 ? atoi(argv[2]) : 1) : 1;
  for (int v = 0; v < nvar; ++v) {
    if (variant) variant(v);
    for (const Shape& sh : shapes) {
      printf("variant %d  %5d x %5d x %4d %-18s", v, sh.M, sh.N, sh.K, sh.what);
      for (int mode = 0; mode <= 3; ++mode) {
        probe(mode);
        const double us = run(sh, 0);
        printf(" | m%d %6.1f us %5.0f TF", mode, us, 2.0 * sh.M * sh.N * sh.K / us * 1e-6);
      }
      probe(0);
      printf("\n     epilogues (direct -> wide, us):");
      const int epis[4] = {0, 1, 2, 4};
      const char* names[4] = {"store", "gelu+pre", "drop+resid", "gelu_bwd"};
      for (int k = 0; k < 4; ++k) {
        if (sh.N > 3072 && k > 0) break;
        double t[2];
        for (int w = 0; w < 2; ++w) { if (wide) wide(w); t[w] = run(sh, epis[k]); }
        printf("  %s %.1f -> %.1f", names[k], t[0], t[1]);
      }
      if (wide) wide(1);
      printf("\n");
    }
  }
  probe(0);
  if (variant) variant(0);
  {   // sustained run: the four forward GEMMs of a layer, cycling through 19 different weight sets for ~0.3 s
    uint16_t* W; const size_t wl = (size_t)(2304 + 768 + 3072 + 3072) * 768;
    hipMalloc(&W, wl * 19 * 2); hipMemset(W, 0x3c, wl * 19 * 2);
    const Shape ls[4] = {{8192, 2304, 768, "qkv"}, {8192, 768, 768, "attn-out"}, {8192, 3072, 768, "ffn1"}, {8192, 768, 3072, "ffn2"}};
    const size_t woff[4] = {0, (size_t)2304 * 768, (size_t)(2304 + 768) * 768, (size_t)(2304 + 768 + 3072) * 768};
    realise_epilogue ep; memset(&ep, 0, sizeof(ep)); ep.out = C; ep.alpha = 1.0f; ep.drop_scale = 1.0f;
    for (int pass = 0; pass < 2; ++pass) {
      const int layers = pass == 0 ? 19 : 19 * 100;
      hipEventRecord(e0, st);
      for (int l = 0; l < layers; ++l)
        for (int k = 0; k < 4; ++k) { ep.ldo = ls[k].N; gemm(st, 1, A, ls[k].K, W + (size_t)(l % 19) * wl + woff[k], ls[k].K, ls[k].M, ls[k].N, ls[k].K, &ep); }
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("sustained: %d layers x 4 GEMMs, %.1f us per layer (isolated sum of the four: see m0 above), %.0f TF\n", layers, ms * 1000.0 / layers,
             2.0 * 8192 * 768 * (2304 + 768 + 3072 + 3072) * layers / (ms * 1e-3) * 1e-12);
    }
  }
  {   // attention forward, B = 64, 12 heads, S = 128 (q/k/v = column slices of one [T, 2304] matrix)
    typedef int (*attn_fn)(void*, int, const void*, const void*, const void*, int64_t, const float*, void*, int64_t, float*, int, int, int, uint32_t, uint32_t, float);
    attn_fn attn = (attn_fn)dlsym(h, "realise_attention_fwd");
    seti_fn aprobe = (seti_fn)dlsym(h, "realise_set_attn_probe");
    float *maskadd, *lse;
    hipMalloc(&maskadd, 8192 * 4); hipMemset(maskadd, 0, 8192 * 4); hipMalloc(&lse, 64 * 12 * 128 * 4);
    if (attn && aprobe) {
      printf("attention fwd (768 workgroups):");
      for (int mode = 0; mode <= 2; ++mode) {
        aprobe(mode);
        for (int drop = 0; drop < 2; ++drop) {
          if (mode != 0 && drop) continue;
          for (int i = 0; i < 3; ++i) attn(st, 1, A, A + 768, A + 1536, 2304, maskadd, C, 768, lse, 64, 12, 128, 7u, drop ? 429496730u : 0u, drop ? 1.0f / 0.9f : 1.0f);
          hipEventRecord(e0, st);
          for (int i = 0; i < 20; ++i) attn(st, 1, A, A + 768, A + 1536, 2304, maskadd, C, 768, lse, 64, 12, 128, 7u, drop ? 429496730u : 0u, drop ? 1.0f / 0.9f : 1.0f);
          hipEventRecord(e1, st); hipEventSynchronize(e1);
          float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
          printf("  mode %d%s %.1f us", mode, drop ? " +dropout" : "", ms * 1000.0 / 20);
        }
      }
      aprobe(0);
      printf("\n");
      typedef int (*attnb_fn)(void*, int, const void*, const void*, const void*, int64_t, const float*, const void*, const void*, int64_t, const float*,
                              float*, void*, void*, void*, int64_t, int, int, int, uint32_t, uint32_t, float);
      attnb_fn attnb = (attnb_fn)dlsym(h, "realise_attention_bwd");
      float* rowdot; hipMalloc(&rowdot, 64 * 12 * 128 * 4);
      if (attnb) {
        for (int drop = 0; drop < 2; ++drop) {
          for (int i = 0; i < 3; ++i) attnb(st, 1, A, A + 768, A + 1536, 2304, maskadd, C2, AUX, 768, lse, rowdot, C, C + 768, C + 1536, 2304, 64, 12, 128, 7u, drop ? 429496730u : 0u, drop ? 1.0f / 0.9f : 1.0f);
          hipEventRecord(e0, st);
          for (int i = 0; i < 20; ++i) attnb(st, 1, A, A + 768, A + 1536, 2304, maskadd, C2, AUX, 768, lse, rowdot, C, C + 768, C + 1536, 2304, 64, 12, 128, 7u, drop ? 429496730u : 0u, drop ? 1.0f / 0.9f : 1.0f);
          hipEventRecord(e1, st); hipEventSynchronize(e1);
          float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
          printf("attention bwd (dkv + dq)%s: %.1f us\n", drop ? " +dropout" : "", ms * 1000.0 / 20);
        }
      }
    }
  }
  {   // weight-gradient shapes: out[I,J] += sum_p A[p,i] B[p,j], P = 8192 (kernel + fold)
    typedef int (*gemm_tn_fn)(void*, int, const void*, int64_t, const void*, int64_t, int, int, int, float*, int64_t, float*, int64_t, float*);
    gemm_tn_fn tn = (gemm_tn_fn)dlsym(h, "realise_gemm_tn");
    seti_fn tnprobe = (seti_fn)dlsym(h, "realise_set_tn_probe");
    float *out, *slab, *colsum;
    const int64_t slab_elems = 16 * 1024 * 1024;
    hipMalloc(&out, (size_t)3072 * 3072 * 4); hipMalloc(&slab, slab_elems * 4); hipMalloc(&colsum, 21128 * 4);
    hipMemset(out, 0, (size_t)3072 * 3072 * 4); hipMemset(colsum, 0, 21128 * 4);
    const Shape tshapes[] = {{8192, 768, 768, "attn-out wgrad"}, {8192, 2304, 768, "qkv wgrad"}, {8192, 3072, 768, "ffn1 wgrad"},
                             {8192, 768, 3072, "ffn2 wgrad"}};
    for (const Shape& sh : tshapes) {
      printf("TN  P %5d  I %5d  J %5d %-16s", sh.M, sh.N, sh.K, sh.what);
      for (int mode = 0; mode <= (tnprobe ? 4 : 0); ++mode) {
        if (mode == 1) continue;
        if (tnprobe) tnprobe(mode);
        for (int i = 0; i < 3; ++i) tn(st, 1, A, sh.N, AUX, sh.K, sh.M, sh.N, sh.K, out, sh.K, slab, slab_elems, colsum);
        const int reps = 20;
        hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) tn(st, 1, A, sh.N, AUX, sh.K, sh.M, sh.N, sh.K, out, sh.K, slab, slab_elems, colsum);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000.0 / reps;
        printf(" | m%d %6.1f us %5.0f TF", mode, us, 2.0 * sh.M * sh.N * sh.K / us * 1e-6);
      }
      printf("\n");
    }
    if (tnprobe) tnprobe(0);
    seti_fn tnsplit = (seti_fn)dlsym(h, "realise_set_tn_split");
    if (tnsplit) {
      for (const Shape& sh : tshapes) {
        printf("TN split sweep I %5d J %5d:", sh.N, sh.K);
        for (int ns = 1; ns <= 16; ++ns) {
          if (ns > 8 && (ns & 1) && ns != 13 && ns != 15) continue;
          tnsplit(ns);
          for (int i = 0; i < 2; ++i) tn(st, 1, A, sh.N, AUX, sh.K, sh.M, sh.N, sh.K, out, sh.K, slab, slab_elems, colsum);
          const int reps = 10;
          hipEventRecord(e0, st);
          for (int i = 0; i < reps; ++i) tn(st, 1, A, sh.N, AUX, sh.K, sh.M, sh.N, sh.K, out, sh.K, slab, slab_elems, colsum);
          hipEventRecord(e1, st); hipEventSynchronize(e1);
          float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
          printf(" %d:%.1f", ns, ms * 1000.0 / reps);
        }
        printf("\n");
      }
      tnsplit(0);
    }
  }
  return 0;
}
