import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import streamk_probe as sp
ctx = sp.Ctx()
def one(M, N, K, mode, acc, use_bias, listed):
    a, b, bias, aux, old = sp.make(M, N, K, mode, 3)
    nb = M // 16
    lst = cnt = None
    if listed:
        lst = torch.arange(nb + 8, dtype=torch.int32, device="cuda"); cnt = torch.tensor([nb], dtype=torch.int32, device="cuda")
    outs = {}
    for which in ("sk", "ref"):
        out = old.clone(); out2 = old.clone() if mode == 1 else None
        ep = sp.epilogue(mode, out, N, acc, out2, bias if use_bias else None, aux, 0.0)
        sp.launch(ctx, which, a, b, M, N, K, ep, lst, cnt); torch.cuda.synchronize()
        outs[which] = out.float()
    d = (outs["sk"] - outs["ref"]).abs()
    bad = d > 0.02
    rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
    print("M %d N %d K %d mode %d acc %d bias %d listed %d: bad %d | rows %s | cols %s" % (M, N, K, mode, acc, use_bias, listed, int(bad.sum()), rows[:24], cols[:40]), flush=True)
    if bad.any() and use_bias:
        r, c = bad.nonzero()[0].tolist()
        print("   first bad (%d, %d): sk %.4f ref %.4f diff %.4f bias[c] %.4f ; diff/bias over bad: %s" % (r, c, outs["sk"][r, c], outs["ref"][r, c], outs["sk"][r, c] - outs["ref"][r, c], bias[c],
              ((outs["sk"] - outs["ref"])[bad][:8]).tolist()))
        # does the difference equal bias[c'] - bias[c] for some shifted column?
        dd = (outs["sk"] - outs["ref"])[r]
        for sh in (-16, -8, -4, 4, 8, 16, 32):
            cc = torch.arange(N, device="cuda"); c2 = (cc + sh).clamp(0, N - 1)
            e = (dd - (bias[c2] - bias[cc])).abs().max().item()
            print("      shift %d: residual %.4f" % (sh, e))
        print("      no-bias residual (sk missing bias): %.4f ; double-bias residual %.4f" % ((dd + bias).abs().max().item(), (dd - bias).abs().max().item()))
for args in [(256, 192, 128, 0, 0, 1, 1), (256, 192, 128, 0, 0, 0, 1), (256, 192, 128, 0, 0, 1, 0), (256, 192, 128, 0, 1, 1, 1), (256, 384, 256, 0, 0, 1, 1), (512, 192, 128, 0, 0, 1, 1),
             (256, 192, 128, 1, 0, 1, 1), (256, 192, 128, 1, 0, 0, 1)]:
    one(*args)
