"""run a few NT / TN GEMM launches for rocprofv3 --pmc collection"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realise_amd import _capi
lib = _capi.load(); dev = torch.device("cuda", 0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 8192, 3072, 768
a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
e = _capi.Epilogue(); e.mode, e.out, e.ldo, e.alpha, e.drop_scale = 0, out.data_ptr(), N, 1.0, 1.0
for _ in range(5):
    lib.realise_gemm_nt(st, 1, a.data_ptr(), K, b.data_ptr(), K, M, N, K, C.byref(e))
P, I, J = 8192, 3072, 768
a2 = torch.randn(P, I, device=dev).bfloat16(); b2 = torch.randn(P, J, device=dev).bfloat16()
o2 = torch.zeros(I, J, device=dev); slab = torch.empty(16 << 20, device=dev)
for _ in range(5):
    lib.realise_gemm_tn(st, 1, a2.data_ptr(), I, b2.data_ptr(), J, P, I, J, o2.data_ptr(), J, slab.data_ptr(), slab.numel(), None)
torch.cuda.synchronize()
