"""Block-1 conv2 (64 -> 64 channels, 3x3 s1 p1, 16x16 maps) forward / input gradient / weight gradient on N distinct images:
the LDS-resident kernels (conv_c64_nt.hip, conv_wgrad_c64.hip) against the generic implicit-GEMM kernels, C ABI + HIP events."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realise_amd import _capi

lib = _capi.load()
dev = torch.device("cuda", 0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


N = int(sys.argv[1]) if len(sys.argv) > 1 else 3740
P = N * 256
x = torch.randn(P, 64, device=dev).bfloat16()
dy = torch.randn(P, 64, device=dev).bfloat16()
wf = (torch.randn(64, 9, 64, device=dev) * 0.05).bfloat16()
y = torch.empty(P, 64, device=dev, dtype=torch.bfloat16)
dw = torch.zeros(64, 64, 3, 3, device=dev)
slab = torch.empty(16 << 20, device=dev)
res = {}
for mode_name, mode, src in (("forward", 0, x), ("dgrad", 1, dy)):
    g = _capi.ConvGeom()
    g.src = src.data_ptr(); g.img_index = None
    g.rows, g.Hr, g.Wr, g.Hs, g.Ws, g.C, g.KH, g.KW, g.stride, g.pad, g.mode = P, 16, 16, 16, 16, 64, 3, 3, 1, 1, mode
    ep = _capi.Epilogue()
    ep.mode, ep.accumulate, ep.out, ep.ldo, ep.alpha, ep.drop_scale = 0, 0, y.data_ptr(), 64, 1.0, 1.0
    fn = lambda: lib.realise_conv_nt(st(), 1, C.byref(g), wf.data_ptr(), 576, P, 64, 576, C.byref(ep))
    outs = []
    for on in (1, 0):
        lib.realise_set_conv_c64(on)
        us = timeit(fn)
        outs.append((us, y.clone()))
    same = torch.equal(outs[0][1], outs[1][1])
    diff = (outs[0][1].float() - outs[1][1].float()).abs().max().item()
    print("%-8s LDS-resident %7.1f us (%.2f TB/s of 2*P*128 B) | generic %7.1f us | identical %s (max diff %.3g)"
          % (mode_name, outs[0][0], 2 * P * 128 / outs[0][0] / 1e6, outs[1][0], same, diff), flush=True)
g = _capi.ConvGeom()
g.src = x.data_ptr(); g.img_index = None
g.rows, g.Hr, g.Wr, g.Hs, g.Ws, g.C, g.KH, g.KW, g.stride, g.pad, g.mode = P, 16, 16, 16, 16, 64, 3, 3, 1, 1, 0
fn = lambda: lib.realise_conv_tn(st(), 1, dy.data_ptr(), 64, C.byref(g), P, 64, 64, dw.data_ptr(), slab.data_ptr(), slab.numel())
outs = []
for on in (1, 0):
    lib.realise_set_conv_c64(on)
    us = timeit(fn)
    dw.zero_(); fn(); torch.cuda.synchronize()
    outs.append((us, dw.clone()))
rel = ((outs[0][1] - outs[1][1]).norm() / outs[1][1].norm()).item()
print("wgrad    LDS-resident %7.1f us | generic %7.1f us | relative difference %.2e" % (outs[0][0], outs[1][0], rel))
lib.realise_set_conv_c64(1)
