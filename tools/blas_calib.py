"""Calibration only (not a product path): what the vendor GEMM reaches on the hot NT shapes, to judge the headroom of gemm_nt8."""
import torch, time
shapes = [(8192, 768, 768), (8192, 2304, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 768, 2304), (8192, 21128, 768), (8192, 768, 21184)]
dev = "cuda"
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        torch.mm(a, b.t(), out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        torch.mm(a, b.t(), out=c)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    print("NT %5d x %5d x %5d  %7.1f us %5.0f TF" % (M, N, K, us, 2.0 * M * N * K / us * 1e-6), flush=True)
# TN (weight-gradient) shapes: out[I,J] = A[P,I]^T B[P,J]
for (I, J, P) in [(768, 768, 8192), (2304, 768, 8192), (3072, 768, 8192), (768, 3072, 8192)]:
    a = torch.randn(P, I, device=dev, dtype=torch.bfloat16)
    b = torch.randn(P, J, device=dev, dtype=torch.bfloat16)
    c = torch.empty(I, J, device=dev, dtype=torch.float32)
    cb = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        torch.mm(a.t(), b, out=cb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.mm(a.t(), b, out=cb)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print("TN %5d x %5d x %5d  %7.1f us %5.0f TF" % (I, J, P, us, 2.0 * I * J * P / us * 1e-6), flush=True)
