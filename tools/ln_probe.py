"""LayerNorm forward / backward timings as a training step meets them (bf16, 8192 x 768): operands cycle through > 700 MB so nothing
is found in the Infinity Cache, the backward writes both outputs and the per-workgroup records.
    python tools/ln_probe.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realise_amd import _capi

lib = _capi.load()
dev = torch.device("cuda", 0)
R, H = 8192, 768
NS = 24                                     # sets: 24 x (2 in + 2 out) x 12.6 MB = 1.2 GB
x = [(torch.randn(R, H, device=dev)).bfloat16() for _ in range(NS)]
dy = [(torch.randn(R, H, device=dev) * 1e-2).bfloat16() for _ in range(NS)]
o1 = [torch.empty(R, H, device=dev, dtype=torch.bfloat16) for _ in range(NS)]
o2 = [torch.empty(R, H, device=dev, dtype=torch.bfloat16) for _ in range(NS)]
rstd = torch.rand(R, device=dev) + 0.5
gamma = torch.rand(H, device=dev) + 0.5
beta = torch.rand(H, device=dev)
dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev)
slots = torch.empty(2 * 1024 * 1024, device=dev)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, iters=3 * NS):
    for i in range(NS):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % NS)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def fwd(i):
    lib.realise_layernorm_fwd(st(), 1, p(x[i]), p(gamma), p(beta), C.c_float(1e-12), p(o1[i]), p(o2[i]), p(rstd), R, H)


def bwd(i, drop=True):
    lib.realise_layernorm_bwd_ex(st(), p(dy[i]), p(x[i]), p(rstd), p(gamma), p(o1[i]), p(o2[i]), 77, 429496730 if drop else 0,
                                 C.c_float(1.0 / 0.9), p(dg), p(db), p(slots), R, H)


live = (torch.rand(R, device=dev) < 0.65).to(torch.uint8)


def bwd_live(i):
    lib.realise_layernorm_bwd_live(st(), p(dy[i]), p(x[i]), p(rstd), p(gamma), p(o1[i]), p(o2[i]), 77, 429496730, C.c_float(1.0 / 0.9), p(dg), p(db),
                                   p(slots), p(live), R, H)


# round 5: v2 = asm row loads behind counted waits + DPP row sums + one-barrier epilogue (realise_set_ln(5, 1)); v1 = the round-4 kernels
for v2 in (0, 1):
    lib.realise_set_ln(5, v2)
    print("v2 %d: ln_fwd %.1f us" % (v2, timeit(fwd)))
    for blocks in ((256, 512, 768) if not v2 else (64, 128, 256, 512)):
        lib.realise_set_ln(1, blocks)
        t, tl = timeit(bwd), timeit(bwd_live)
        print("v2 %d blocks %4d: ln_bwd + fold dense %.1f us (50 MB -> %.2f TB/s); 65 %% live rows %.1f us" % (v2, blocks, t, 50.3e6 / t / 1e6, tl))
    lib.realise_set_ln(1, 0)
lib.realise_set_ln(5, 1)
