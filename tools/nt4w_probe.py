"""Round 6 probe: the layer shapes on a 128 x 192 tile computed by FOUR waves of 64 x 96 (two workgroups per
CU; variants 60 / 61 of the 4-wave kernel, probe build) against the shipped 8-wave 128 x 192 two-per-CU kernel (variant 16) and the
4-wave 128 x 128 kernel (variant 9): plain stores, every launch on its own operand / output set (cold), us per launch.
    REALISE_HIP_PROBES=1 python tools/nt4w_probe.py"""
import ctypes as C
import os
import sys
os.environ.setdefault("REALISE_HIP_PROBES", "1")
import torch
sys.path.insert(0, ".")
from realise_amd import _capi

lib = _capi.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [(8192, 768, 768, "attn-out"), (8192, 2304, 768, "qkv"), (8192, 3072, 768, "FFN-up"), (8192, 768, 3072, "FFN-down"), (8192, 768, 2304, "qkv dgrad")]
SETS = 6
for M, N, K, what in SHAPES:
    A = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(SETS)]
    B = [torch.randn(N, K, device="cuda").bfloat16() * 0.05 for _ in range(SETS)]
    O = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(SETS)]
    ref = None
    line = "%-10s %5d x %4d x %4d |" % (what, M, N, K)
    for v, name in ((16, "8-wave 128x192 (shipped)"), (9, "4-wave 128x128"), (60, "4-wave 128x192, 64x96/wave"), (61, "same, spread fetches")):
        lib.realise_set_nt_variant(v)
        ep = _capi.Epilogue(mode=0, accumulate=0, out=O[0].data_ptr(), ldo=N, alpha=1.0, drop_scale=1.0)
        rc = lib.realise_gemm_nt(st, 1, A[0].data_ptr(), K, B[0].data_ptr(), K, M, N, K, C.byref(ep))
        torch.cuda.synchronize()
        if rc != 0:
            line += " %s: rc %d |" % (name, rc)
            continue
        if ref is None:
            ref = O[0].clone()
        same = torch.equal(ref, O[0])
        for _ in range(2):
            for s in range(SETS):
                ep.out = O[s].data_ptr()
                lib.realise_gemm_nt(st, 1, A[s].data_ptr(), K, B[s].data_ptr(), K, M, N, K, C.byref(ep))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            for s in range(SETS):
                ep.out = O[s].data_ptr()
                lib.realise_gemm_nt(st, 1, A[s].data_ptr(), K, B[s].data_ptr(), K, M, N, K, C.byref(ep))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * SETS)
        line += " %s: %.1f us %.0f TF%s |" % (name, us, 2.0 * M * N * K / us * 1e-6, "" if same else " (differs!)")
    lib.realise_set_nt_variant(0)
    print(line, flush=True)
