// Probe of the ping-pong 8-wave NT kernel (gemm_nt8.hip) against the 4-wave production kernel on the BERT-stack shapes:
// bit-exact comparison of the outputs (same k-order of the fp32 accumulation), then timings with / without fetches / MFMAs.
//   hipcc --offload-arch=gfx950 -O2 tools/nt8_probe.cpp -o tools/_bin/nt8_probe -ldl && tools/_bin/nt8_probe realise_amd/librealise_hip.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cmath>
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
  typedef void (*setii_fn)(int, int);
  setii_fn nt8p_knob = (setii_fn)dlsym(h, "realise_set_nt8p");
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
  if (argc > 3 && !strcmp(argv[3], "pers")) {
    // persistent kernel (variant 50) against the 4-wave reference kernel (variant 9): bit-exact outputs on the shapes / epilogues it
    // supports, ragged tiles, odd K-tile counts, repeated runs (race screen)
    size_t bad = 0;
    const Shape more[] = {{1000, 776, 128, "ragged M, N"}, {300, 2304, 768, "small M"}, {513, 200, 192, "nk=3"}, {256, 256, 320, "nk=5"},
                          {8000, 3064, 768, "ragged M, N, 2 tiles/CU"}, {4096, 21128, 768, "classifier half"}, {8192, 384, 1536, "few tiles"}};
    for (const Shape& sh : more)
      for (int epi : {0, 1, 4}) {
        if (epi != 0 && (size_t)sh.M * sh.N > (size_t)8192 * 3072) continue;      // the second output / aux buffers hold 8192 x 3072
        bad += compare(sh, epi, 50);
      }
    for (int rep = 0; rep < 3; ++rep) {
      bad += compare(shapes[3], 0, 50); bad += compare(shapes[4], 1, 50); bad += compare(shapes[4], 4, 50); bad += compare(shapes[7], 0, 50);
    }
    printf("TOTAL mismatches: %zu\n", bad);
    return bad ? 2 : 0;
  }
  if (argc > 3 && !strcmp(argv[3], "dual")) {
    // One chain of the four forward GEMMs of a transformer layer over M = 8192 rows on one stream, against the same work as two
    // independent row halves (M = 4096 each) on two streams: does a second chain fill the launch ramps / output-write tails of the
    // first?  19 weight sets and 4 activation sets cycle so that operands come from HBM as in a training step.
    uint16_t* pool; const size_t wl = (size_t)(2304 + 768 + 3072 + 3072) * 768;
    const size_t act = (size_t)8192 * (768 + 2304 + 768 + 3072 + 3072 + 768);
    const int nact = 6;
    hipMalloc(&pool, (wl * 19 + act * nact) * 2);
    for (int l = 0; l < 19; ++l) hipMemcpy(pool + l * wl, B, wl * 2, hipMemcpyDeviceToDevice);
    for (int a = 0; a < nact; ++a) hipMemcpy(pool + 19 * wl + a * act, A, act * 2, hipMemcpyDeviceToDevice);
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipStream_t s1; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipEvent_t ej; hipEventCreate(&ej);
    auto layer = [&](hipStream_t stx, int l, int row0, int rows) {
      const uint16_t* W = pool + (size_t)(l % 19) * wl;
      uint16_t* a = pool + 19 * wl + (size_t)(l % nact) * act;
      uint16_t* x = a + (size_t)row0 * 768;                                   // [8192][768]
      uint16_t* qkv = a + (size_t)8192 * 768 + (size_t)row0 * 2304;
      uint16_t* ctx = a + (size_t)8192 * (768 + 2304) + (size_t)row0 * 768;   // stands in for the attention output
      uint16_t* pre = a + (size_t)8192 * (768 + 2304 + 768) + (size_t)row0 * 3072;
      uint16_t* post = a + (size_t)8192 * (768 + 2304 + 768 + 3072) + (size_t)row0 * 3072;
      uint16_t* y = a + (size_t)8192 * (768 + 2304 + 768 + 3072 + 3072) + (size_t)row0 * 768;
      realise_epilogue ep; memset(&ep, 0, sizeof(ep)); ep.alpha = 1.0f; ep.drop_scale = 1.0f; ep.bias = bias;
      ep.mode = 0; ep.out = qkv; ep.ldo = 2304; gemm(stx, 1, x, 768, W, 768, rows, 2304, 768, &ep);
      ep.mode = 2; ep.out = y; ep.ldo = 768; ep.aux = x; ep.ldaux = 768; ep.drop_seed = 77; ep.drop_thresh = 429496730u; ep.drop_scale = 1.0f / 0.9f;
      gemm(stx, 1, ctx, 768, W + (size_t)2304 * 768, 768, rows, 768, 768, &ep);
      ep.mode = 1; ep.out = post; ep.out2 = pre; ep.ldo = 3072; ep.aux = nullptr;
      gemm(stx, 1, y, 768, W + (size_t)(2304 + 768) * 768, 768, rows, 3072, 768, &ep);
      ep.mode = 2; ep.out = x; ep.out2 = nullptr; ep.ldo = 768; ep.aux = y; ep.ldaux = 768;
      gemm(stx, 1, post, 3072, W + (size_t)(2304 + 768 + 3072) * 768, 3072, rows, 768, 3072, &ep);
    };
    const int layers = 19 * 8;
    for (int v : {0, 12, 14}) {
      variant(v);
      for (int mode = 0; mode < 3; ++mode) {          // 0: one stream M = 8192; 1: two streams x 4096; 2: one stream, 2 x 4096 back to back
        for (int rep = 0; rep < 2; ++rep) {
          hipDeviceSynchronize();
          hipEventRecord(e0, s1);
          if (mode == 1) { hipEventRecord(ej, s1); hipStreamWaitEvent(s2, ej, 0); }
          for (int l = 0; l < layers; ++l) {
            if (mode == 0) layer(s1, l, 0, 8192);
            else if (mode == 1) { layer(s1, l, 0, 4096); layer(s2, l, 4096, 4096); }
            else { layer(s1, l, 0, 4096); layer(s1, l, 4096, 4096); }
          }
          if (mode == 1) { hipEventRecord(ej, s2); hipStreamWaitEvent(s1, ej, 0); }
          hipEventRecord(e1, s1); hipEventSynchronize(e1);
          float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
          if (rep == 1)
            printf("dual v%-2d mode %d (%s): %.1f us per layer, %.0f TF\n", v, mode,
                   mode == 0 ? "1 stream x 8192" : mode == 1 ? "2 streams x 4096" : "1 stream, 2 x 4096", ms * 1000.0 / layers,
                   2.0 * 8192 * 768 * (2304 + 768 + 3072 + 3072) * layers / (ms * 1e-3) * 1e-12);
        }
      }
    }
    variant(0);
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "cold")) {
    // "cold" timing: every repetition works on its own operand / output set and the sets cycle through > 700 MB, so no operand is
    // found in the 256 MB Infinity Cache or an L2 by the time it is used again - the condition a GEMM meets inside a training step
    // (the warm loop above re-reads the same buffers: everything after the first repetition is a cache hit).
    uint16_t* pool; const size_t pool_elems = (size_t)1600 << 20;        // 3.2 GB of bf16
    hipMalloc(&pool, pool_elems * 2);
    for (size_t off = 0; off < pool_elems; off += maxA) hipMemcpy(pool + off, A, std::min(maxA, pool_elems - off) * 2, hipMemcpyDeviceToDevice);
    auto time_cold = [&](const Shape& sh, int epi, int reps) {
      const size_t a = (size_t)sh.M * sh.K, b = (size_t)sh.N * sh.K, c = (size_t)sh.M * sh.N;
      const size_t set = a + b + c * (epi == 1 ? 2 : 1) + ((epi == 2 || epi == 4) ? c : 0);
      int nsets = (int)(((size_t)400 << 20) / set) + 2;              // >= 800 MB between two uses of a set
      if ((size_t)nsets * set > pool_elems) nsets = (int)(pool_elems / set);
      auto run = [&](int i) {
        uint16_t* base = pool + (size_t)(i % nsets) * set;
        realise_epilogue ep; memset(&ep, 0, sizeof(ep));
        ep.mode = epi; ep.out = base + a + b; ep.ldo = sh.N; ep.alpha = 1.0f; ep.drop_scale = 1.0f; ep.bias = bias;
        if (epi == 1) ep.out2 = base + a + b + c;
        if (epi == 2) { ep.aux = base + a + b + c; ep.ldaux = sh.N; ep.drop_seed = 77; ep.drop_thresh = 429496730u; ep.drop_scale = 1.0f / 0.9f; }
        if (epi == 4) { ep.aux = base + a + b + c; ep.ldaux = sh.N; ep.bias = nullptr; }
        gemm(st, 1, base, sh.K, base + a, sh.K, sh.M, sh.N, sh.K, &ep);
      };
      for (int i = 0; i < nsets; ++i) run(i);
      hipEventRecord(e0, st);
      for (int i = 0; i < reps; ++i) run(i);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
      return ms * 1000.0 / reps;
    };
    std::vector<int> cv;
    for (int i = 4; i < argc; ++i) cv.push_back(atoi(argv[i]));
    if (cv.empty()) cv = {16, 12, 14, 13, 11};
    struct Job { int shape, epi; };
    const Job jobs[] = {{2, 2}, {3, 0}, {4, 1}, {4, 4}, {5, 2}, {5, 0}, {6, 0}, {7, 0}, {8, 0}};
    for (const Job& j : jobs) {
      const Shape& sh = shapes[j.shape];
      for (int v : cv) {
        if (sh.N > 3072 && (v % 10 == 3 || v % 10 == 4)) continue;
        variant(v);
        const int reps = sh.N > 3072 || sh.K > 3072 ? 6 : 24;
        printf("v%-2d %5d x %5d x %5d epi %d %-22s", v, sh.M, sh.N, sh.K, j.epi, sh.what);
        for (int g : {0, 4, 8}) {
          group_m(g);
          if (nt8p_knob) nt8p_knob(0, g == 0 ? 1 : 0);        // variant 50: gm0 column = XCD-owned rows (default), the others = chunked ids
          const double warm = (g == 0) ? time_us(sh, j.epi, reps) : 0.0, cold = time_cold(sh, j.epi, reps);
          if (g == 0) printf(" | warm %7.1f us %5.0f TF", warm, 2.0 * sh.M * sh.N * sh.K / warm * 1e-6);
          printf(" | cold gm%d %7.1f us %5.0f TF", g, cold, 2.0 * sh.M * sh.N * sh.K / cold * 1e-6);
        }
        group_m(0);
        if (nt8p_knob) nt8p_knob(0, 1);
        printf("\n"); fflush(stdout);
      }
    }
    variant(0);
    return 0;
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
