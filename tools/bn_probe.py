"""Times the BatchNorm passes of the glyph branch alone (bf16, through the C ABI's debug entry points), fast paths against the
generic kernels, and compares their results.  usage: python tools/bn_probe.py   (on an MI355X)
Shapes: block 1 of the CharResNet on the dedup'd batch of configs[1] (~3738 distinct glyphs x 16x16 pixels x 64 channels), block 2
(8x8 x 128), and block 1 of the dense configs[3] batch (32768 glyph stacks)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from realise_amd import _capi  # noqa: E402

lib = _capi.load()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / reps


def run(images, hw, Cc, dedup=True):
    P = images * hw
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: (torch.randn(P, Cc, device="cuda", generator=g) * 1.3 + 0.2).to(torch.bfloat16)
    # rotate over several buffer sets so that no pass finds its inputs in the 256 MB Infinity Cache
    nset = max(2, int(1.2e9 // (P * Cc * 2 * 6)))
    sets = [dict(x=mk(), xs=mk(), dy=mk(), o=mk(), y=torch.empty(P, Cc, device="cuda", dtype=torch.bfloat16),
                 dxa=torch.empty(P, Cc, device="cuda", dtype=torch.bfloat16), dxb=torch.empty(P, Cc, device="cuda", dtype=torch.bfloat16)) for _ in range(nset)]
    counts = (torch.randint(1, 5, (images,), device="cuda", generator=g).float() if dedup else None)
    n_stat = int(counts.sum().item()) * hw if dedup else P
    mean, sq = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    mean_b, rstd = torch.zeros(Cc, device="cuda") + 0.1, torch.ones(Cc, device="cuda") * 0.8
    gamma, beta = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda")
    sums, slots = torch.empty(4 * Cc, device="cuda"), torch.empty(262144, device="cuda")
    dg, db = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    dg2, db2 = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    scratch = torch.empty(4 * Cc, device="cuda")
    sc_, sh_ = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    it = [0]

    def nxt():
        it[0] += 1
        return sets[it[0] % nset]

    def stats():
        s = nxt()
        _capi.check(lib.realise_batchnorm_stats_ex(st(), p(s["x"]), P, Cc, hw, p(counts), n_stat, p(gamma), p(beta), 1e-5, 0.1, p(rm), p(rv), None, p(mean),
                                                   p(sq), p(sc_), p(sh_), p(scratch), p(slots)), "stats")

    def apply1():
        s = nxt()
        _capi.check(lib.realise_batchnorm_fwd(st(), _capi.BF16, p(s["x"]), P, Cc, p(gamma), p(beta), C.c_float(1e-5), C.c_float(0.1), p(rm), p(rv), None, 0, 1,
                                              p(s["y"]), None, None, p(scratch)), "apply")

    def bwd(pair):
        s = nxt()
        _capi.check(lib.realise_batchnorm_bwd_ex(st(), p(s["dy"]), p(s["o"]), P, Cc, hw, p(counts), n_stat, p(s["x"]), p(mean_b), p(rstd), p(gamma), p(s["dxa"]),
                                                 p(dg), p(db), p(s["xs"]) if pair else None, p(mean_b), p(rstd), p(gamma), p(s["dxb"]), p(dg2), p(db2), p(sums),
                                                 p(slots)), "bwd")

    mb = P * Cc * 2 / 1e6
    res = {}
    for fast in (0, 3, 1):          # generic kernels; 16-byte kernels with two-pass statistics; the same with one-pass statistics
        lib.realise_set_ln(2, fast)
        rm.zero_(); rv.fill_(1.0)
        t_stats, t_apply, t_b1, t_b2 = timed(stats), timed(apply1), timed(lambda: bwd(False)), timed(lambda: bwd(True))
        # results on set 0 for the comparison
        it[0] = -1
        rm.zero_(); rv.fill_(1.0)
        stats(); m0, q0 = mean.clone(), sq.clone()
        it[0] = -1
        dg.zero_(); db.zero_(); dg2.zero_(); db2.zero_()
        bwd(True)
        torch.cuda.synchronize()
        res[fast] = (m0, q0, sets[0]["dxa"].float().clone(), sets[0]["dxb"].float().clone(), dg.clone(), dg2.clone(), db.clone())
        print("P %8d C %3d hw %3d %s | fast %d | stats + finalize %7.1f us | apply (1r 1w) %7.1f us %5.2f TB/s | bwd single (6r 1w) %7.1f us %5.2f TB/s |"
              " bwd bn2+shortcut %7.1f us (generic: 12r 2w, fast: 8r 2w) %5.2f TB/s" %
              (P, Cc, hw, "dedup" if dedup else "dense", fast, t_stats, t_apply, 2 * mb / t_apply, t_b1, 7 * mb / t_b1, t_b2,
               (10 if fast else 14) * mb / t_b2), flush=True)
    lib.realise_set_ln(2, 1)
    a, b = res[0], res[1]
    names = ("mean", "rstd", "dx_a", "dx_b", "dgamma_a", "dgamma_b", "dbeta")
    for n, u, v in zip(names, a, b):
        d = (u - v).abs().max().item()
        ref = u.abs().max().item()
        print("   fast vs generic %-9s max |diff| %.3e (max |value| %.3e)" % (n, d, ref))
        assert d <= 2e-2 * max(ref, 1e-6) + 1e-6, n


if __name__ == "__main__":
    print(lib.realise_version().decode())
    run(3738, 256, 64)
    run(3738, 64, 128)
    run(3738, 16, 256)
    run(32768, 256, 64, dedup=False)
