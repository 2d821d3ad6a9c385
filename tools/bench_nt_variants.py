import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realise_amd import _capi
lib = _capi.load(); dev = torch.device("cuda", 0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for shp in [(8192, 2304, 768), (8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 21128, 768), (8192, 768, 2304), (5300, 768, 768), (5300, 2304, 768)]:
    M, N, K = shp
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = (a.float() @ b.float().t())
    e = _capi.Epilogue(); e.mode, e.out, e.ldo, e.alpha, e.drop_scale = 0, out.data_ptr(), N, 1.0, 1.0
    res = []
    for big, stages in [(0, 2 | 0x100), (0, 2)]:
        lib.realise_set_nt_allow_n96(0 if stages & 0x100 else 1)
        us = timeit(lambda: lib.realise_gemm_nt(st(), 1, a.data_ptr(), K, b.data_ptr(), K, M, N, K, C.byref(e)))
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        res.append("%s: %6.1fus %5.0fTF err %.1e" % ("128x128 only" if stages & 0x100 else "128x96 allowed", us, 2.0 * M * N * K / us / 1e6, err))
    print(shp, " | ".join(res))
