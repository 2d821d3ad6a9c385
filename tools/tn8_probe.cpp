// Probe of the ping-pong 8-wave TN kernel (gemm_tn8.hip) against the 4-wave kernel on the weight-gradient shapes of the BERT stacks.
//   hipcc --offload-arch=gfx950 -O2 tools/tn8_probe.cpp -o tools/_bin/tn8_probe -ldl && tools/_bin/tn8_probe realise_amd/librealise_hip.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/realise_hip.h"

typedef int (*gemm_tn_fn)(void*, int, const void*, int64_t, const void*, int64_t, int, int, int, float*, int64_t, float*, int64_t, float*);
typedef void (*seti_fn)(int);

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "realise_amd/librealise_hip.so";
  void* h = dlopen(path, RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  gemm_tn_fn gemm = (gemm_tn_fn)dlsym(h, "realise_gemm_tn");
  seti_fn variant = (seti_fn)dlsym(h, "realise_set_tn_variant");
  seti_fn split = (seti_fn)dlsym(h, "realise_set_tn_split");
  if (!gemm || !variant || !split) { fprintf(stderr, "missing symbols\n"); return 1; }
  struct Shape { int P, I, J; int64_t lda, ldb; const char* what; };
  const Shape shapes[] = {{8192, 768, 768, 768, 768, "attn-out wgrad"}, {8192, 2304, 768, 2304, 768, "qkv wgrad"}, {8192, 3072, 768, 3072, 768, "ffn1 wgrad"},
                          {8192, 768, 3072, 768, 3072, "ffn2 wgrad"}, {8192, 21128, 768, 21184, 768, "classifier wgrad (padded dlogits)"},
                          {3000, 520, 136, 520, 136, "ragged"}, {1024, 256, 128, 256, 128, "minimal"}};
  const size_t maxA = (size_t)8192 * 21184, maxB = (size_t)8192 * 3072, maxC = (size_t)21128 * 768 > (size_t)3072 * 768 ? (size_t)21128 * 768 : 0;
  uint16_t *A, *B; float *C, *Cref, *slab, *cs, *csref;
  hipMalloc(&A, maxA * 2); hipMalloc(&B, maxB * 2); hipMalloc(&C, maxC * 4); hipMalloc(&Cref, maxC * 4);
  const int64_t slab_elems = 16ll << 20;
  hipMalloc(&slab, slab_elems * 4); hipMalloc(&cs, 21184 * 4); hipMalloc(&csref, 21184 * 4);
  {
    std::vector<uint16_t> hbuf(maxA);
    uint32_t s = 777;
    for (auto& v : hbuf) { s = s * 1664525u + 1013904223u; float f = (((s >> 8) / 8388608.0f) - 1.0f) * 0.1f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    hipMemcpy(A, hbuf.data(), maxA * 2, hipMemcpyHostToDevice);
    for (size_t i = 0; i < maxB; ++i) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) / 8388608.0f) - 1.0f; uint32_t u; memcpy(&u, &f, 4); hbuf[i] = (uint16_t)(u >> 16); }
    hipMemcpy(B, hbuf.data(), maxB * 2, hipMemcpyHostToDevice);
  }
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto call = [&](const Shape& sh, float* out, float* colsum) {
    return gemm(st, 1, A, sh.lda, B, sh.ldb, sh.P, sh.I, sh.J, out, sh.J, slab, slab_elems, colsum);
  };
  size_t bad_total = 0;
  for (const Shape& sh : shapes) {
    const size_t n = (size_t)sh.I * sh.J;
    for (int sp : {0, 1, 2, 5}) {
      if (sp == 5 && sh.I > 4000) continue;
      hipMemsetAsync(C, 0, n * 4, st); hipMemsetAsync(Cref, 0, n * 4, st); hipMemsetAsync(cs, 0, 21184 * 4, st); hipMemsetAsync(csref, 0, 21184 * 4, st);
      variant(0); split(0); call(sh, Cref, csref);
      variant(8); split(sp); const int rc = call(sh, C, cs); split(0);
      // accumulate semantics: a second call doubles the result
      if (sp == 0) call(sh, C, cs);
      hipStreamSynchronize(st);
      std::vector<float> a(n), b(n), ca(sh.I), cb(sh.I);
      hipMemcpy(a.data(), Cref, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), C, n * 4, hipMemcpyDeviceToHost);
      hipMemcpy(ca.data(), csref, sh.I * 4, hipMemcpyDeviceToHost); hipMemcpy(cb.data(), cs, sh.I * 4, hipMemcpyDeviceToHost);
      const float mul = sp == 0 ? 2.0f : 1.0f;
      double mx = 0, md = 0, cmx = 0, cmd = 0;
      for (size_t i = 0; i < n; ++i) { mx = fmax(mx, fabs(a[i])); md = fmax(md, fabs(a[i] * mul - b[i])); }
      for (int i = 0; i < sh.I; ++i) { cmx = fmax(cmx, fabs(ca[i])); cmd = fmax(cmd, fabs(ca[i] * mul - cb[i])); }
      const bool ok = md <= 2e-5 * mx * mul + 1e-6 && cmd <= 2e-5 * cmx * mul + 1e-6;
      if (!ok) ++bad_total;
      printf("  check split %d  P %5d I %5d J %5d %-34s rc %d  max|ref| %.3e  max diff %.3e  colsum diff %.3e (max %.3e)  %s\n", sp, sh.P, sh.I, sh.J, sh.what, rc, mx, md, cmd, cmx,
             ok ? "ok" : "MISMATCH");
    }
  }
  printf("TOTAL mismatching cases: %zu\n", bad_total);
  fflush(stdout);
  auto time_us = [&](const Shape& sh, int reps) {
    for (int i = 0; i < 3; ++i) call(sh, C, cs);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) call(sh, C, cs);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / reps;
  };
  for (int k = 0; k < 5; ++k) {
    const Shape& sh = shapes[k];
    printf("%-34s", sh.what);
    for (int v : {0, 8}) {
      variant(v); split(0);
      const double us = time_us(sh, k == 4 ? 5 : 20);
      printf(" | v%d %7.1f us %5.0f TF", v, us, 2.0 * sh.P * sh.I * sh.J / us * 1e-6);
    }
    variant(8);
    printf(" | tn8 splits:");
    for (int sp : {1, 2, 3, 4, 5, 6, 8, 12, 14, 16}) {
      if (k == 4 && sp > 2) break;
      split(sp);
      printf(" %d:%.1f", sp, time_us(sh, k == 4 ? 5 : 20));
    }
    split(0);
    printf("\n");
    fflush(stdout);
  }
  return bad_total ? 2 : 0;
}
