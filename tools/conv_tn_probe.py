"""Where the conv weight-gradient time goes: each glyph-ResNet wgrad shape on 3740 distinct images (the dedup'd B=64 batch),
normal / no fetches (probe 2) / no MFMA (probe 3) / no fold (probe 4), through the C ABI with HIP events."""
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


def geom(src, rows, Hr, Hs, Cc, k, stride, pad, mode):
    g = _capi.ConvGeom()
    g.src = src.data_ptr(); g.img_index = None
    g.rows, g.Hr, g.Wr, g.Hs, g.Ws, g.C, g.KH, g.KW, g.stride, g.pad, g.mode = rows, Hr, Hr, Hs, Hs, Cc, k, k, stride, pad, mode
    return g


N = int(sys.argv[1]) if len(sys.argv) > 1 else 3740
slab = torch.empty(64 << 20, device=dev)
shapes = [("b1 conv1", 32, 8, 3, 64, 3, 2, 1), ("b1 short", 32, 8, 3, 64, 1, 2, 0), ("b1 conv2", 16, 64, 64, 64, 3, 1, 1),
          ("b2 conv1", 16, 64, 64, 128, 3, 2, 1), ("b2 short", 16, 64, 64, 128, 1, 2, 0), ("b2 conv2", 8, 128, 128, 128, 3, 1, 1),
          ("b3 conv1", 8, 128, 128, 256, 3, 2, 1), ("b3 short", 8, 128, 128, 256, 1, 2, 0), ("b3 conv2", 4, 256, 256, 256, 3, 1, 1),
          ("b4 conv1", 4, 256, 256, 512, 3, 2, 1), ("b4 short", 4, 256, 256, 512, 1, 2, 0), ("b4 conv2", 2, 512, 512, 512, 3, 1, 1),
          ("b5 conv1", 2, 512, 512, 768, 3, 2, 1), ("b5 short", 2, 512, 512, 768, 1, 2, 0), ("b5 conv2", 1, 768, 768, 768, 3, 1, 1)]
for name, Hin, Cp, Ci, Co, k, s, p in shapes:
    Hout = (Hin + 2 * p - k) // s + 1
    x = torch.randn(N, Hin, Hin, Cp, device=dev).bfloat16()
    Pn = N * Hout * Hout
    dy = torch.randn(Pn, Co, device=dev).bfloat16()
    g = geom(x, Pn, Hout, Hin, Cp, k, s, p, 0)
    dw = torch.zeros(Co, Ci, k, k, device=dev)
    fn = lambda: lib.realise_conv_tn(st(), 1, dy.data_ptr(), Co, C.byref(g), Pn, Co, Ci, dw.data_ptr(), slab.data_ptr(), slab.numel())
    res = []
    for mode in (0, 2, 3, 4):
        lib.realise_set_tn_probe(mode)
        res.append(timeit(fn))
    lib.realise_set_tn_probe(0)
    fl = 2.0 * Pn * Co * k * k * Cp
    mb = (Pn * Co * 2 + N * Hin * Hin * Cp * 2) / 1e6
    print("%-9s P %7d I %4d J %5d | %7.1f us %6.1f TF | no-fetch %7.1f | no-mfma %7.1f | no-fold %7.1f | operands %6.1f MB -> %5.2f TB/s"
          % (name, Pn, Co, k * k * Cp, res[0], fl / res[0] / 1e6, res[1], res[2], res[3], mb, mb / res[0] / 1e6 * 1e6 / 1e6), flush=True)
