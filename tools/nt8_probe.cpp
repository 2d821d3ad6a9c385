// Probe of the ping-pong 8-wave NT kernel (gemm_nt8.hip) against the 4-wave production kernel on the BERT-stack shapes:
// bit-exact comparison of the outputs (same k-order of the fp32 accumulation), then timings with / without fetches / MFMAs.
//   hipcc --offload-arch=gfx950 -O2 tools/nt8_probe.cpp -o tools/_bin/nt8_probe -ldl && tools/_bin/nt8_probe realise_amd/librealise_hip.so
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
  const int quick = argc > 2 ? atoi(argv[2]) : 0;
  void* h = dlopen(path, RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  gemm_nt_fn gemm = (gemm_nt_fn)dlsym(h, "realise_gemm_nt");
  seti_fn probe = (seti_fn)dlsym(h, "realise_set_nt_probe");
  seti_fn variant = (seti_fn)dlsym(h, "realise_set_nt_variant");
  seti_fn group_m = (seti_fn)dlsym(h, "realise_set_nt_group_m");
  if (!gemm || !probe || !variant || !group_m) { fprintf(stderr, "missing symbols\n"); return 1; }
  struct Shape { int M, N, K; const char* what; };
  const Shape shapes[] = {{8192, 768, 64, "one K-tile"}, {8192, 3072, 64, "one K-tile"}, {8192, 768, 768, "attn-out / dgrad"}, {8192, 2304, 768, "qkv"},
                          {8192, 3072, 768, "ffn1 / ffn2-dgrad"}, {8192, 768, 3072, "ffn2 / ffn1-dgrad"}, {8192, 768, 2304, "qkv-dgrad"},
                          {8192, 21128, 768, "classifier"}, {8192, 768, 21128, "classifier dgrad"}};
  const Shape odd[] = {{1000, 776, 128, "ragged M, N"}, {300, 2304, 768, "small M"}, {8192, 768, 64, "nk=1"}, {513, 200, 192, "nk=3"}, {256, 256, 320, "nk=5"}};
  size_t maxA = (size_t)8192 * 21128, maxB = (size_t)21128 * 768, maxC = (size_t)8192 * 21128;
  uint16_t *A, *B, *C, *Cref;
  hipMalloc(&A, maxA * 2); hipMalloc(&B, maxB * 2); hipMalloc(&C, maxC * 2); hipMalloc(&Cref, maxC * 2);
  {
    std::vector<uint16_t> hbuf(maxA > maxB ? maxA : maxB);
    uint32_t s = 12345;
    for (auto& v : hbuf) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) / 8388608.0f) - 1.0f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    hipMemcpy(A, hbuf.data(), maxA * 2, hipMemcpyHostToDevice);
    for (auto& v : hbuf) { s = s * 1664525u + 1013904223u; float f = (((s >> 8) / 8388608.0f) - 1.0f) * 0.05f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    hipMemcpy(B, hbuf.data(), maxB * 2, hipMemcpyHostToDevice);
  }
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  uint16_t *C2, *AUX; float* bias;
  hipMalloc(&C2, (size_t)8192 * 3072 * 2); hipMalloc(&AUX, (size_t)8192 * 3072 * 2); hipMalloc(&bias, 21128 * 4);
  hipMemcpy(AUX, A, (size_t)8192 * 3072 * 2, hipMemcpyDeviceToDevice);
  { std::vector<float> hb(21128); for (int i = 0; i < 21128; ++i) hb[i] = 0.001f * (i % 97); hipMemcpy(bias, hb.data(), 21128 * 4, hipMemcpyHostToDevice); }
  auto call = [&](const Shape& sh, int epi, uint16_t* out) {
    realise_epilogue ep; memset(&ep, 0, sizeof(ep));
    ep.mode = epi; ep.out = out; ep.ldo = sh.N; ep.alpha = 1.0f; ep.drop_scale = 1.0f;
    ep.bias = bias;
    if (epi == 1) ep.out2 = C2;
    if (epi == 2) { ep.aux = AUX; ep.ldaux = sh.N; ep.drop_seed = 77; ep.drop_thresh = 429496730u; ep.drop_scale = 1.0f / 0.9f; }
    if (epi == 4) { ep.aux = AUX; ep.ldaux = sh.N; ep.bias = nullptr; }
    return gemm(st, 1, A, sh.K, B, sh.K, sh.M, sh.N, sh.K, &ep);
  };
  auto time_us = [&](const Shape& sh, int epi, int reps) {
    for (int i = 0; i < 3; ++i) call(sh, epi, C);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) call(sh, epi, C);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / reps;
  };
  auto compare = [&](const Shape& sh, int epi, int v) {
    const size_t n = (size_t)sh.M * sh.N;
    hipMemsetAsync(C, 0xEE, n * 2, st); hipMemsetAsync(Cref, 0xEE, n * 2, st);
    variant(9); call(sh, epi, Cref);
    variant(v); const int rc = call(sh, epi, C);
    hipStreamSynchronize(st);
    std::vector<uint16_t> a(n), b(n);
    hipMemcpy(a.data(), Cref, n * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), C, n * 2, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < n; ++i) if (a[i] != b[i]) { if (!bad) first = i; ++bad; }
    printf("  check v%-2d epi %d  %5d x %5d x %5d %-28s rc %d  mismatches %zu", v, epi, sh.M, sh.N, sh.K, sh.what, rc, bad);
    if (bad) printf("  (first at row %zu col %zu: ref %04x got %04x)", first / sh.N, first % sh.N, a[first], b[first]);
    printf("\n");
    return bad;
  };
  if (argc > 3 && !strcmp(argv[3], "gm")) {     // tile-order sweep: production tile choice, group_m rows per column step
    size_t bad = 0;
    for (int g : {8, 5}) { group_m(g); bad += compare(shapes[3], 0, 0); bad += compare(shapes[4], 1, 0); bad += compare(shapes[5], 2, 0); bad += compare(odd[0], 0, 16); bad += compare(odd[3], 0, 16); bad += compare(shapes[7], 0, 0); }
    printf("TOTAL mismatches: %zu\n", bad);
    for (const Shape& sh : shapes) {
      if (sh.K == 64 || sh.K > 4000) continue;
      printf("%5d x %5d x %5d %-22s", sh.M, sh.N, sh.K, sh.what);
      variant(0);
      for (int g : {0, 2, 4, 8, 16, 32}) { group_m(g); printf(" | gm%-2d %7.1f us", g, time_us(sh, 0, sh.N > 3072 ? 5 : 20)); }
      printf("\n"); fflush(stdout);
    }
    group_m(0);
    return bad ? 2 : 0;
  }
  size_t total_bad = 0;
  const bool ws = argc > 3 && !strcmp(argv[3], "ws");
  std::vector<int> vars = {11, 12, 13, 14};
  if (ws) vars = {15, 16};
  for (int v : vars) {
    for (const Shape& sh : odd) total_bad += compare(sh, 0, v);
    total_bad += compare(shapes[2], 0, v);
    total_bad += compare(shapes[4], 1, v);
    total_bad += compare(shapes[5], 2, v);
    total_bad += compare(shapes[4], 4, v);
  }
  total_bad += compare(shapes[7], 0, 10);
  total_bad += compare(shapes[8], 0, 10);
  total_bad += compare(shapes[8], 0, 11);
  total_bad += compare(shapes[8], 0, 12);
  { const Shape t1 = {777, 520, 200, "ragged K tail"}; for (int v : {11, 12, 13, 14}) total_bad += compare(t1, 0, v); }
  for (int rep = 0; rep < 3; ++rep) total_bad += compare(shapes[3], 0, 10);     // race screen: repeated runs of the auto choice
  printf("TOTAL mismatches: %zu\n", total_bad);
  fflush(stdout);

  std::vector<int> tv = {9, 0, 11, 12, 13, 14};
  if (ws) tv = {12, 14, 15, 16};
  for (const Shape& sh : shapes) {
    for (int v : tv) {
      if (quick && v > 10) continue;
      if (sh.N > 3072 && (v % 10 == 3 || v % 10 == 4) && !ws) continue;
      variant(v);
      printf("v%-2d %5d x %5d x %5d %-22s", v, sh.M, sh.N, sh.K, sh.what);
      const int reps = sh.N > 3072 || sh.K > 3072 ? 5 : 20;
      for (int mode : (ws && v >= 40) ? std::vector<int>{0, 2, 3, 4, 5} : std::vector<int>{0, 2, 3}) {
        probe(mode);
        const double us = time_us(sh, 0, reps);
        printf(" | m%d %7.1f us %5.0f TF", mode, us, 2.0 * sh.M * sh.N * sh.K / us * 1e-6);
      }
      probe(0);
      if (sh.N <= 3072) {
        printf(" | epi");
        for (int epi : {1, 2, 4}) printf(" %d:%.1f", epi, time_us(sh, epi, reps));
      }
      printf("\n");
      fflush(stdout);
    }
  }
  probe(0);
  for (int v : ws ? std::vector<int>{0, 15, 16} : std::vector<int>{9, 0}) {   // sustained: the four forward GEMMs of a layer, cycling through 19 weight sets
    variant(v);
    uint16_t* W; const size_t wl = (size_t)(2304 + 768 + 3072 + 3072) * 768;
    hipMalloc(&W, wl * 19 * 2); hipMemcpy(W, B, wl * 2, hipMemcpyDeviceToDevice);
    for (int l = 1; l < 19; ++l) hipMemcpy(W + l * wl, W, wl * 2, hipMemcpyDeviceToDevice);
    const Shape ls[4] = {{8192, 2304, 768, "qkv"}, {8192, 768, 768, "attn-out"}, {8192, 3072, 768, "ffn1"}, {8192, 768, 3072, "ffn2"}};
    const size_t woff[4] = {0, (size_t)2304 * 768, (size_t)(2304 + 768) * 768, (size_t)(2304 + 768 + 3072) * 768};
    realise_epilogue ep; memset(&ep, 0, sizeof(ep)); ep.out = C; ep.alpha = 1.0f; ep.drop_scale = 1.0f;
    const int layers = 19 * 20;
    hipEventRecord(e0, st);
    for (int l = 0; l < layers; ++l)
      for (int k = 0; k < 4; ++k) { ep.ldo = ls[k].N; gemm(st, 1, A, ls[k].K, W + (size_t)(l % 19) * wl + woff[k], ls[k].K, ls[k].M, ls[k].N, ls[k].K, &ep); }
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("sustained v%d: %d layers x 4 GEMMs, %.1f us per layer, %.0f TF\n", v, layers, ms * 1000.0 / layers,
           2.0 * 8192 * 768 * (2304 + 768 + 3072 + 3072) * layers / (ms * 1e-3) * 1e-12);
    hipFree(W);
  }
  variant(0);
  return total_bad ? 2 : 0;
}
