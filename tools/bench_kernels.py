"""GPU micro-benchmark of the GEMM / conv kernel families on the model's own shapes (C ABI, HIP events)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realise_amd import _capi

lib = _capi.load()
dev = torch.device("cuda", 0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def epi(out, ldo):
    e = _capi.Epilogue()
    e.mode, e.accumulate, e.out, e.ldo, e.alpha, e.drop_scale = 0, 0, out.data_ptr(), ldo, 1.0, 1.0
    return e


def bench_nt(M, N, K, variant):
    lib.realise_set_nt_allow_n96(0 if variant == 1 else 1)
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    e = epi(out, N)
    us = timeit(lambda: lib.realise_gemm_nt(st(), 1, a.data_ptr(), K, b.data_ptr(), K, M, N, K, C.byref(e)))
    return us, 2.0 * M * N * K / us / 1e6


def bench_tn(P, I, J, variant):
    a = torch.randn(P, I, device=dev).bfloat16(); b = torch.randn(P, J, device=dev).bfloat16()
    out = torch.zeros(I, J, device=dev)
    slab = torch.empty(16 << 20, device=dev)
    sp, sn = (slab.data_ptr(), slab.numel()) if variant == 2 else (None, 0)
    us = timeit(lambda: lib.realise_gemm_tn(st(), 1, a.data_ptr(), I, b.data_ptr(), J, P, I, J, out.data_ptr(), J, sp, sn, None))
    return us, 2.0 * P * I * J / us / 1e6


def geom(src, rows, Hr, Hs, Cc, k, stride, pad, mode):
    g = _capi.ConvGeom()
    g.src = src.data_ptr(); g.img_index = None
    g.rows, g.Hr, g.Wr, g.Hs, g.Ws, g.C, g.KH, g.KW, g.stride, g.pad, g.mode = rows, Hr, Hr, Hs, Hs, Cc, k, k, stride, pad, mode
    return g


def bench_conv(N, Hin, Cin, Co, k, stride, pad, variant):
    lib.realise_set_nt_allow_n96(0 if variant == 1 else 1)
    Hout = (Hin + 2 * pad - k) // stride + 1
    x = torch.randn(N, Hin, Hin, Cin, device=dev).bfloat16()
    w = torch.randn(Co, k * k * Cin, device=dev).bfloat16()
    Pn = N * Hout * Hout
    y = torch.empty(Pn, Co, device=dev, dtype=torch.bfloat16)
    g = geom(x, Pn, Hout, Hin, Cin, k, stride, pad, 0)
    e = epi(y, Co)
    K = k * k * Cin
    us_f = timeit(lambda: lib.realise_conv_nt(st(), 1, C.byref(g), w.data_ptr(), K, Pn, Co, K, C.byref(e)), 10)
    dw = torch.zeros(Co, Cin, k, k, device=dev)
    slab = torch.empty(16 << 20, device=dev)
    sp, sn = (slab.data_ptr(), slab.numel()) if variant == 2 else (None, 0)
    us_w = timeit(lambda: lib.realise_conv_tn(st(), 1, y.data_ptr(), Co, C.byref(g), Pn, Co, Cin, dw.data_ptr(), sp, sn), 10)
    fl = 2.0 * Pn * Co * K
    return us_f, fl / us_f / 1e6, us_w, fl / us_w / 1e6


print("== NT GEMM (M,N,K): us / TFLOPs  [NT/conv fwd: v1 = 128x128 tiles only | v2 = 128x96 allowed;  TN/wgrad: v1 = atomics | v2 = slabs + fold]")
for shp in [(8192, 2304, 768), (8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 21128, 768), (8192, 768, 21128),
            (8192, 768, 2304), (3000, 2304, 768)]:
    r1, r2 = bench_nt(*shp, 1), bench_nt(*shp, 2)
    print("  %-22s v1 %8.1f us %7.1f TF | v2 %8.1f us %7.1f TF" % (shp, r1[0], r1[1], r2[0], r2[1]))
print("== TN GEMM (P,I,J)")
for shp in [(8192, 768, 768), (8192, 2304, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 21128, 768), (5000, 64, 2304)]:
    r1, r2 = bench_tn(*shp, 1), bench_tn(*shp, 2)
    print("  %-22s v1 %8.1f us %7.1f TF | v2 %8.1f us %7.1f TF" % (shp, r1[0], r1[1], r2[0], r2[1]))
print("== conv fwd / wgrad (N,Hin,Cin,Co,k,s,p)")
for shp in [(8192, 32, 8, 64, 3, 2, 1), (8192, 16, 64, 64, 3, 1, 1), (8192, 16, 64, 128, 3, 2, 1), (8192, 8, 128, 128, 3, 1, 1),
            (8192, 8, 128, 256, 3, 2, 1), (8192, 4, 256, 256, 3, 1, 1), (8192, 4, 256, 512, 3, 2, 1), (8192, 2, 512, 512, 3, 1, 1),
            (8192, 2, 512, 768, 3, 2, 1), (8192, 1, 768, 768, 3, 1, 1)]:
    r1, r2 = bench_conv(*shp, 1), bench_conv(*shp, 2)
    print("  %-34s v1 fwd %8.1f us %6.1f TF wgrad %8.1f us %6.1f TF | v2 fwd %8.1f us %6.1f TF wgrad %8.1f us %6.1f TF" % ((shp,) + r1 + r2))
