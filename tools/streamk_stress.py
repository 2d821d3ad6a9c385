#!/usr/bin/env python3
"""400 back-to-back stream-K launches per layer shape (realise_gemm_nt_streamk): median / p99 / worst launch time and the give-up flag -
how the rare 30 - 60 ms launches of the first polling forms were found (profiles/round5_streamk_probe.log).  GPU box: python tools/streamk_stress.py"""
import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import streamk_probe as sp
ctx = sp.Ctx()
def stress(N, K, mode, acc, kind, launches=400):
    M = 8192; rng = np.random.default_rng(5); nb = M // 16
    a, b, bias, aux, old = sp.make(M, N, K, mode, 11)
    if kind == "dense": lst = cnt = None; live = np.arange(nb)
    else:
        live = sp.live_blocks(kind, nb, rng)
        lst = torch.full((nb + 8,), -7, dtype=torch.int32, device="cuda"); lst[:len(live)] = torch.from_numpy(live.astype(np.int32)).cuda()
        cnt = torch.tensor([len(live)], dtype=torch.int32, device="cuda")
    out = old.clone(); out2 = old.clone() if mode == 1 else None
    ep = sp.epilogue(mode, out, N, acc, out2, bias if mode != 4 else None, aux, 0.1 if mode == 2 else 0.0)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for e0, e1 in evs:
        e0.record(); sp.launch(ctx, "sk", a, b, M, N, K, ep, lst, cnt); e1.record()
    torch.cuda.synchronize()
    t = np.array([e0.elapsed_time(e1) * 1e3 for e0, e1 in evs])
    print("N %d K %d mode %d %s: %d launches, median %.1f us, p99 %.1f, max %.1f, > 200 us: %s, timeout flag %d" % (N, K, mode, kind, launches, np.median(t), np.percentile(t, 99), t.max(),
          [(int(i), round(float(x))) for i, x in enumerate(t) if x > 200][:10], int(ctx.timeout[0].item())), flush=True)
    ctx.timeout.zero_()
for _ in range(2):
    stress(768, 3072, 2, 0, "bench"); stress(768, 3072, 0, 1, "bench"); stress(2304, 768, 0, 0, "bench"); stress(3072, 768, 1, 0, "dense"); stress(768, 2304, 0, 1, "bench")
