"""GPU tests added in round 5: the incoming loss gradient as a device scalar (no pass over the logits gradient), the safe default of the
fused optimizer's operand copies, live-row steps under the non-default weight-gradient knobs, stale non-finite rows of
a live-row step, and this round's kernels."""
import ctypes as C

import numpy as np
import pytest
import torch

from realise_amd import _capi
from realise_amd.config import RealiseConfig
from realise_amd.data import synthetic_batch
from realise_amd.init import init_state_dict_numpy
from realise_amd.modeling import SpellBertPho2ResArch3

pytestmark = pytest.mark.gpu


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def build(cfg, sd_np, dtype, train=False, **kw):
    m = SpellBertPho2ResArch3(cfg, compute_dtype=dtype, **kw)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()})
    m.to("cuda")
    m.train(train)
    return m


def cuda_batch(B, S, seed):
    b = synthetic_batch(B, S, seed=seed)
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}


def grads_of(m):
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_loss_gradient_is_applied_as_a_device_scalar(dtype):
    """loss.backward() hands d loss to the engine as a DEVICE scalar (realise_engine_set_loss_grad): the classifier head's three
    gradients are scaled in the kernels that store them, the cross-entropy gradient rows are not touched.  (a) grad_output = 1 gives
    bit for bit what assume_unit_loss_grad gives on every order-fixed gradient; (b) (0.5 * loss).backward() gives exactly half of
    every gradient (a power of two commutes with every rounding on the way); (c) the saved logits gradient is left alone."""
    cfg = RealiseConfig(num_hidden_layers=2, pho_layers=1, out_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = init_state_dict_numpy(cfg, seed=5)
    batch = cuda_batch(8, 64, 21)

    def run(scale, assume):
        m = build(cfg, sd, dtype, train=True)
        m.assume_unit_loss_grad = assume
        m.zero_grad()
        loss, _ = m(batch)
        d0 = m.tap_dlogits().clone()
        (loss * scale if scale != 1.0 else loss).backward()
        torch.cuda.synchronize()
        assert torch.equal(d0, m.tap_dlogits()), "the logits gradient rows were rewritten"
        return grads_of(m)

    g_assume, g_one, g_half = run(1.0, True), run(1.0, False), run(0.5, False)
    for n in g_assume:       # (run to run the tensors behind float atomics - embeddings, BatchNorm / gate sums - differ in their last bits)
        s = g_assume[n].abs().max().item()
        assert (g_assume[n] - g_one[n]).abs().max().item() <= 2e-5 * s + 1e-12, n
        assert (0.5 * g_assume[n] - g_half[n]).abs().max().item() <= 2e-5 * s + 1e-12, n
    layer_w = [n for n in g_assume if ".layer." in n and n.endswith("dense.weight")]
    assert layer_w and all(torch.equal(g_assume[n], g_one[n]) for n in layer_w)          # order-fixed kernels: bit for bit
    head = "classifier.bias"
    assert torch.equal(g_assume[head], g_one[head]) or (g_assume[head] - g_one[head]).abs().max().item() <= 1e-6 * g_assume[head].abs().max().item()
    assert g_half[head].abs().max().item() > 0.0


def test_fused_optimizer_default_sees_a_raw_parameter_write():
    """the module default re-derives every operand copy on every forward, also right after a FusedAdamW step, so a
    p.data write between opt.step() and the next forward is seen; a loop that opts into trust_fused_optimizer (trainer.train(),
    bench.py) and makes such a write must call mark_parameters_updated()."""
    from realise_amd.optim import FusedAdamW
    cfg = RealiseConfig(num_hidden_layers=1, pho_layers=1, out_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = init_state_dict_numpy(cfg, seed=2)
    batch = cuda_batch(4, 32, 3)

    def run(trust, mark):
        m = build(cfg, sd, "bf16", train=True)
        assert m.trust_fused_optimizer is False            # the class default
        m.trust_fused_optimizer = trust
        opt = FusedAdamW(m, [{"params": [p for p in m.parameters() if p.requires_grad], "weight_decay": 0.0}], lr=1e-4, eps=1e-8)
        m.zero_grad()
        m(batch)[0].backward()
        opt.step()
        w = dict(m.named_parameters())["bert.encoder.layer.0.intermediate.dense.weight"]
        w.data.zero_()                                     # a write torch's version counters do not see
        if mark:
            m.mark_parameters_updated()
        m.eval()
        with torch.no_grad():
            out = m(batch)[1].float().clone()
        torch.cuda.synchronize()
        return out

    seen_default, stale_trusted, seen_marked = run(False, False), run(True, False), run(True, True)
    # (the arena-level AdamW of the untrusted run and the engine sweep of the trusted one round a step differently in the last bits)
    # the two 'seen' runs differ by isolated single bf16 steps of a logit (2^-7 relative: 0.0625 at |logit| >= 8), the run that missed the
    # write differs everywhere: the maximum is held to two such steps, the comparison is made on the mean
    seen = (seen_default - seen_marked).abs().max().item()
    missed = (seen_default - stale_trusted).abs().max().item()
    seen_mean = (seen_default - seen_marked).abs().mean().item()
    missed_mean = (seen_default - stale_trusted).abs().mean().item()
    print("zeroed FFN weight: default vs marked %.3e (mean %.3e), default vs trusted-unmarked %.3e (mean %.3e)" % (seen, seen_mean, missed, missed_mean))
    assert seen <= max(2e-2, 2.0 ** -6 * seen_default.abs().max().item())
    assert missed_mean > 20 * max(seen_mean, 1e-4), "the trusted run was expected to miss the raw write (that is what the flag trades)"


@pytest.mark.parametrize("knob", ["wgrad_group", "tn_group_ring"])
def test_live_row_step_under_the_other_weight_gradient_forms(knob):
    """only the grouped two-stage weight-gradient launch walks the live-block list.  With the grouped launch off the
    step must not run live rows at all (the dense weight gradients would sum stale rows); with the four-stage ring requested the listed
    launch keeps the two-stage ring.  Either way every gradient equals the default configuration's."""
    lib = _capi.load()
    cfg = RealiseConfig(num_hidden_layers=2, pho_layers=1, out_layers=1)
    sd = init_state_dict_numpy(cfg, seed=11)
    batch = cuda_batch(8, 64, 17)
    setter = lib.realise_set_wgrad_group if knob == "wgrad_group" else lib.realise_set_tn_group_ring
    default, other = (1, 0) if knob == "wgrad_group" else (0, 1)

    def step(v):
        setter(v)
        try:
            m = build(cfg, sd, "bf16", train=True)
            out = []
            for _ in range(2):                 # the second step meets stale rows in the workspace
                m.zero_grad()
                loss, _ = m(batch)
                loss.backward()
                torch.cuda.synchronize()
                out.append((float(loss.item()), grads_of(m)))
            return out
        finally:
            setter(default)

    a, b = step(default), step(other)
    for (l0, g0), (l1, g1) in zip(a, b):
        assert l0 == l1
        for n in g0:
            s = g0[n].abs().max().item()
            assert torch.isfinite(g1[n]).all(), n
            # (ungrouped weight gradients split their reductions: another summation order, so not bit for bit)
            assert (g0[n] - g1[n]).abs().max().item() <= 2e-3 * s + 1e-12, n


def test_stale_non_finite_rows_do_not_reach_a_live_row_step():
    """A live-row step does not rewrite the activation rows behind a sentence's last live position; whatever an earlier step left
    there - here NaN, written through the activation taps between two steps - is selected away, never multiplied by zero
    (masked mean, gate gradients): loss and every gradient equal the undisturbed second step's."""
    cfg = RealiseConfig(num_hidden_layers=2, pho_layers=1, out_layers=1)
    sd = init_state_dict_numpy(cfg, seed=13)
    B, S = 8, 64
    batch = cuda_batch(B, S, 19)
    masks = batch["masks"].bool() | batch["loss_masks"].bool()
    last = torch.where(masks.any(1), S - masks.flip(1).float().argmax(1), torch.zeros(B, dtype=torch.long, device="cuda"))
    dead = ~(torch.arange(S, device="cuda")[None, :] < last[:, None]).reshape(-1)
    # a 16-row block with a live row is computed whole: poison only the rows of blocks that hold no live row
    blk_dead = dead.view(-1, 16).all(1).repeat_interleave(16)
    assert blk_dead.any()

    def run(poison):
        m = build(cfg, sd, "bf16", train=True)
        out = []
        for it in range(2):
            m.zero_grad()
            loss, _ = m(batch)
            loss.backward()
            torch.cuda.synchronize()
            out.append((float(loss.item()), grads_of(m)))
            if poison and it == 0:
                # what a live-row step leaves alone in the blocks without a live row: the GEMM outputs (qkv, the pre-LayerNorm sums, the
                # FFN activations) and the attention output; the LayerNorm outputs of those rows are recomputed from the poisoned sums
                names = ["%s.layer.%d.%s" % (st, l, t) for st, n in (("bert", 2), ("pho_model", 1), ("output_block", 1)) for l in range(n)
                         for t in ("qkv", "ctx", "inter", "sum1", "sum2")]
                for name in names:
                    m.tap(name).view(B * S, -1)[blk_dead] = float("nan")
                torch.cuda.synchronize()
        return out

    clean, dirty = run(False), run(True)
    assert clean[1][0] == dirty[1][0] and np.isfinite(dirty[1][0])
    for n in clean[1][1]:
        s = clean[1][1][n].abs().max().item()
        assert torch.isfinite(dirty[1][1][n]).all(), n
        assert (clean[1][1][n] - dirty[1][1][n]).abs().max().item() <= 5e-5 * s + 1e-12, n


@pytest.mark.parametrize("rows,H,live_frac,drop,blocks", [(8192, 768, 0.65, 0.1, 0), (8192, 768, 1.0, 0.0, 0), (8192, 768, 0.65, 0.1, 128),
                                                          (5000, 768, 0.5, 0.1, 0), (8192, 1024, 0.8, 0.1, 0), (8192, 256, 0.3, 0.0, 0),
                                                          (1024, 768, 0.0, 0.1, 0), (200, 768, 0.7, 0.1, 0)])
def test_layernorm_backward_round5_kernel_against_round4_and_fp32(rows, H, live_frac, drop, blocks):
    """ln_bwd16v2_kernel (realise_set_ln(5, 1), default): asm row loads behind counted vmcnt waits, four rows of a wave in flight,
    DPP row sums, branch-free padding rows (their loads parked beyond the buffer), one-barrier epilogue.  Against the round-4 kernel
    (same arithmetic, another summation order: bf16 outputs within one rounding step, dgamma / dbeta to fp32 rounding) and against
    fp32 autograd of the same LayerNorm; the xhat rows of padding tokens hold NaN - stale as can be - and must not be read into
    anything."""
    lib = _capi.load()
    g = torch.Generator().manual_seed(rows + H)
    x = (torch.randn(rows, H, generator=g) * 1.5 + 0.3).cuda()
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).cuda()
    mean, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    rstd = (1.0 / torch.sqrt(var + 1e-12)).reshape(-1).contiguous()
    xhat = ((x - mean) * rstd[:, None]).bfloat16()
    live = (torch.rand(rows, generator=g) < live_frac).cuda()
    dy = (torch.randn(rows, H, generator=g) * 0.05).cuda().bfloat16()
    dy[~live] = 0
    xh_in = xhat.clone()
    xh_in[~live] = float("nan")
    row_live = live.to(torch.uint8).contiguous()
    slots = torch.empty(2 * 1024 * 1024, device="cuda")
    thresh, scale = (int(drop * 4294967296.0), 1.0 / (1.0 - drop)) if drop > 0 else (0, 1.0)

    def run(v2):
        lib.realise_set_ln(5, v2)
        lib.realise_set_ln(1, blocks)
        try:
            dx = torch.full((rows, H), 7.0, device="cuda").bfloat16()
            dxd = torch.full((rows, H), 7.0, device="cuda").bfloat16()
            dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
            _capi.check(lib.realise_layernorm_bwd_live(stream(), P(dy), P(xh_in), P(rstd), P(gamma), P(dx), P(dxd), 77, thresh, C.c_float(scale),
                                                       P(dg), P(db), P(slots), P(row_live), rows, H), "ln_bwd_live")
            torch.cuda.synchronize()
            return dx, dxd, dg, db
        finally:
            lib.realise_set_ln(1, 0)
            lib.realise_set_ln(5, 1)

    new, old = run(1), run(0)
    for t in new:
        assert torch.isfinite(t.float()).all()
    # padding rows: exact zeros in both outputs
    assert (new[0][~live] == 0).all() and (new[1][~live] == 0).all()
    # against the round-4 kernel
    for a, b, name in zip(new[:2], old[:2], ("dx", "dx_drop")):
        d = (a.float() - b.float()).abs()
        assert (d <= 2.0 ** -7 * b.float().abs() + 1e-6).all(), (name, d.max().item())
        assert (d > 0).float().mean().item() < 0.02, name           # a different last bit is rare
    if drop > 0:
        keep = new[1] != 0
        assert torch.equal(keep, old[1] != 0) or ((keep != (old[1] != 0)) & (new[0] != 0)).sum().item() == 0
        assert abs(keep[live].float().mean().item() - (1 - drop)) < 0.01 or not live.any()
    for a, b, name in zip(new[2:], old[2:], ("dgamma", "dbeta")):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-6, name
    # against fp32 autograd on the same (bf16-rounded) inputs
    xf = xhat.float()[live]
    dyf = dy.float()[live]
    t = dyf * gamma
    ref_dx = rstd[live, None] * (t - t.mean(1, keepdim=True) - xf * (t * xf).mean(1, keepdim=True))
    if live.any():
        assert (new[0].float()[live] - ref_dx).abs().max().item() <= 1e-2 * ref_dx.abs().max().item() + 1e-6
        assert (new[2] - (dyf * xf).sum(0)).abs().max().item() <= 1e-3 * (dyf * xf).sum(0).abs().max().item() + 1e-5
        assert (new[3] - dyf.sum(0)).abs().max().item() <= 1e-3 * dyf.sum(0).abs().max().item() + 1e-5


@pytest.mark.parametrize("rows,H", [(8192, 768), (37, 768), (1000, 256)])
def test_layernorm_forward_dpp_row_sums(rows, H):
    """ln_fwd16 with the DPP half-wave sums (realise_set_ln(5, 1)) against the ds_bpermute form: y / xhat within one bf16 rounding step, rstd
    to fp32 rounding."""
    lib = _capi.load()
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, H, generator=g) * 2 + 0.5).cuda().bfloat16()
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).cuda()
    beta = (0.1 * torch.randn(H, generator=g)).cuda()

    def run(v2):
        lib.realise_set_ln(5, v2)
        try:
            y, xh, rs = torch.empty_like(x), torch.empty_like(x), torch.empty(rows, device="cuda")
            _capi.check(lib.realise_layernorm_fwd(stream(), 1, P(x), P(gamma), P(beta), C.c_float(1e-12), P(y), P(xh), P(rs), rows, H), "ln")
            torch.cuda.synchronize()
            return y, xh, rs
        finally:
            lib.realise_set_ln(5, 1)

    a, b = run(1), run(0)
    assert (a[2] - b[2]).abs().max().item() <= 1e-5 * b[2].abs().max().item()
    for u, v in zip(a[:2], b[:2]):
        d = (u.float() - v.float()).abs()
        assert (d <= 2.0 ** -7 * v.float().abs() + 1e-6).all()
    ref = torch.nn.functional.layer_norm(x.float(), (H,), gamma, beta, 1e-12)
    assert (a[0].float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("gc", [1, 2, 4])
@pytest.mark.parametrize("M,N,K,mode,accumulate,blocks", [(8192, 2304, 768, 0, 0, "ragged"), (8192, 3072, 768, 1, 0, "ragged"),
                                                          (8192, 768, 3072, 2, 0, "ragged"), (2048, 768, 768, 0, 0, "all"),
                                                          (1024, 768, 768, 2, 0, "one"), (1024, 768, 768, 0, 1, "none")])
def test_live_row_gemm_under_every_xcd_split(gc, M, N, K, mode, accumulate, blocks):
    """EpiParams::xcd_gc (realise_set_nt8p(3, gc)): the 2-D XCD split of the live-row GEMM only changes WHICH workgroup computes a
    tile - every listed row carries the dense launch's bits, unlisted rows are untouched - for every split the shapes divide into
    (the default picks 2 column groups for the wide K = 768 outputs)."""
    import test_round4_gpu as r4
    lib = _capi.load()
    lib.realise_set_nt8p(3, gc)
    try:
        r4.test_nt_gemm_over_a_list_of_live_row_blocks(M, N, K, mode, accumulate, blocks)
    finally:
        lib.realise_set_nt8p(3, 0)


def _streamk_probe():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("streamk_probe", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "streamk_probe.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("M,N,K,mode,accumulate,kind", [
    (8192, 2304, 768, 0, 0, "bench"),           # qkv over a batch-shaped live list: tiles cut between workgroups
    (8192, 3072, 768, 1, 0, "dense"),           # FFN-up + GELU, all rows: two whole tiles per workgroup, no cut - bit-identical
    (8192, 768, 3072, 2, 0, "ragged"),          # FFN-down + dropout + residual: a tile's K range spread over three or four workgroups
    (8192, 3072, 768, 4, 0, "bench"),           # GELU' data gradient
    (8192, 768, 2304, 0, 1, "bench"),           # accumulating data gradient
    (1024, 768, 768, 2, 0, "one"),              # one listed block: most workgroups have nothing to do
    (256, 192, 128, 0, 0, "odd"),               # one tile, two K-tiles
    (1008, 776, 128, 0, 0, "dense"),            # ragged N, rows not a multiple of 256
    (512, 200, 256, 0, 1, "dense")])
def test_stream_k_layer_gemm_against_the_one_chain_kernels(M, N, K, mode, accumulate, kind):
    """gemm_nt8s (realise_gemm_nt_streamk; realise_set_engine(11, 1) - off by default, measured slower): one round of 256 workgroups over
    equal K-tile ranges of 256 x 192 tiles, cut tiles folded in-kernel in workgroup order.  Same operands, same dropout masks as the
    128 x 192 kernels: results within one bf16 rounding step (a cut tile sums its K range in two or three fp32 chains; uncut tiles are
    bit-identical), rows of unlisted blocks untouched although their A rows are NaN, two launches bit-identical, no wait gave up."""
    if b"+probes" not in _capi.load().realise_version():
        pytest.skip("the stream-K kernel ships in the probe build only (round 6: python -m realise_amd.build --probes, REALISE_HIP_PROBES=1)")
    sp = _streamk_probe()
    assert sp.check(sp.Ctx(), M, N, K, mode, accumulate, kind, "test")


def test_stream_k_training_step_is_reproducible_and_close_to_the_default():
    """realise_set_engine(11, 1) + (12, 0): every layer GEMM the stream-K kernel supports runs on it (forward and data gradients, live
    rows).  Two runs from the same state give the same bits; against the default kernels the loss and every gradient agree to the
    tolerance the other summation-order knobs are held to; no finisher gave up waiting (the engine's flag word stays 0)."""
    lib = _capi.load()
    if b"+probes" not in lib.realise_version():
        pytest.skip("the stream-K kernel ships in the probe build only")
    cfg = RealiseConfig(num_hidden_layers=2, pho_layers=1, out_layers=1)
    sd = init_state_dict_numpy(cfg, seed=11)
    batch = cuda_batch(16, 128, 23)

    def run(on):
        lib.realise_set_engine(11, on)
        lib.realise_set_engine(12, 0 if on else 10)
        try:
            m = build(cfg, sd, "bf16", train=True)
            out = []
            for _ in range(2):
                m.zero_grad()
                loss, _ = m(batch)
                loss.backward()
                torch.cuda.synchronize()
                out.append((float(loss.item()), grads_of(m)))
            m.check_ids()                      # raises if a stream-K wait gave up
            return out
        finally:
            lib.realise_set_engine(11, 0)
            lib.realise_set_engine(12, 10)

    ref, a, b = run(0), run(1), run(1)
    for (l0, g0), (l1, g1), (l2, g2) in zip(ref, a, b):
        assert l1 == l2
        assert abs(l0 - l1) <= 2e-3 * abs(l0)
        layer_w = [n for n in g0 if ".layer." in n and n.endswith("weight") and "LayerNorm" not in n]
        assert layer_w and all(torch.equal(g1[n], g2[n]) for n in layer_w)      # order-fixed kernels: bit for bit run to run
        for n in g0:
            s = g0[n].abs().max().item()
            assert torch.isfinite(g1[n]).all(), n
            if n.endswith("attention.self.key.bias"):      # (softmax is invariant to it: its gradient is rounding noise around an exact zero)
                continue
            assert (g1[n] - g2[n]).abs().max().item() <= 2e-5 * s + 1e-12, n      # (the tensors behind float atomics: last bits)
            if n.startswith("resnet."):          # (upstream of the BatchNorm backward: bf16 rounding differences of the stacks above are amplified - DESIGN.md section 3)
                cos = torch.nn.functional.cosine_similarity(g0[n].flatten().double(), g1[n].flatten().double(), dim=0).item()
                assert cos >= 0.99, (n, cos)
            else:
                assert (g0[n] - g1[n]).abs().max().item() <= 3e-2 * s + 1e-12, n


@pytest.mark.parametrize("rows,H,S", [(8192, 768, 128), (640, 768, 40), (512, 256, 64)])
def test_layernorm_forward_skips_blocks_without_a_live_row(rows, H, S):
    """ln_fwd with LnFwdArgs::row_live (realise_layernorm_fwd_live), as dense_resid_ln launches it in a live-row step: 16-row blocks
    without a live row are not visited (outputs keep their contents, NaN inputs there do no harm), every other row - dead rows of a
    partly live block included - carries the plain launch's bits."""
    lib = _capi.load()
    g = torch.Generator().manual_seed(rows + H)
    x = torch.randn(rows, H, generator=g).bfloat16().cuda()
    gamma = (1.0 + 0.1 * torch.randn(H, generator=g)).float().cuda()
    beta = (0.1 * torch.randn(H, generator=g)).float().cuda()
    live = torch.zeros(rows + 64, dtype=torch.uint8)
    rng = np.random.default_rng(rows)
    for b0 in range(0, rows, S):                 # a live prefix per sentence (S = 40: blocks straddle sentences)
        live[b0:b0 + int(rng.integers(1, S + 1))] = 1
    live = live.cuda()
    blk_live = (live[:rows].view(-1, 16).sum(1) > 0).repeat_interleave(16)

    def run(use_live):
        y = torch.full((rows, H), 7.0, dtype=torch.bfloat16, device="cuda")
        xh = torch.full((rows, H), 7.0, dtype=torch.bfloat16, device="cuda")
        rs = torch.full((rows,), 7.0, dtype=torch.float32, device="cuda")
        xin = x.clone()
        if use_live:
            xin[~blk_live] = float("nan")
            _capi.check(lib.realise_layernorm_fwd_live(stream(), P(xin), P(gamma), P(beta), 1e-12, P(y), P(xh), P(rs), P(live), rows, H), "ln_fwd_live")
        else:
            _capi.check(lib.realise_layernorm_fwd(stream(), _capi.BF16, P(xin), P(gamma), P(beta), 1e-12, P(y), P(xh), P(rs), rows, H), "ln_fwd")
        torch.cuda.synchronize()
        return y, xh, rs

    y1, xh1, rs1 = run(True)
    y0, xh0, rs0 = run(False)
    assert torch.equal(y1[blk_live], y0[blk_live]) and torch.equal(xh1[blk_live], xh0[blk_live]) and torch.equal(rs1[blk_live], rs0[blk_live])
    dead = ~blk_live
    if dead.any():
        assert (y1[dead].float() == 7.0).all() and (xh1[dead].float() == 7.0).all() and (rs1[dead] == 7.0).all()


@pytest.mark.parametrize("name", ["arch3_b8s256_train", "arch3_b4s512_train"])
def test_long_sequence_train_step_fp32_matches_reference_golden(golden_dir, name):
    """max_seq_length beyond the default 128 (/root/reference/src/run.py:304; position table of 512 rows): the tiled attention kernels
    (attention.hip, S > 128) inside the whole model.  fp32 parity mode, train mode, dropout 0, against the reference's own forward +
    loss.backward() at B = 8, S = 256 and B = 4, S = 512 (oracle/make_golden_full.py train256 / train512): loss, logits, arg-max ids,
    taps, BatchNorm buffers, every parameter gradient - the bars of the B = 64, S = 128 case (tests/test_round3_gpu.py)."""
    import test_round3_gpu as r3
    from helpers import check_summary, golden_case_inputs, load_golden
    g = load_golden(golden_dir, name)
    cfg, sd_np, batch = golden_case_inputs(g, "arch3")
    m = build(cfg, sd_np, "fp32", train=True)
    loss, logits = m(batch)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    check_summary(g, "logits", logits.float(), 1e-3)
    am = logits.argmax(-1).cpu().numpy().astype(np.int32)
    assert (am == g["argmax"]).mean() > 0.999
    check_summary(g, "tap/bert_h", m.tap("bert.layer.11.out").float(), 2e-4, what="tap")
    check_summary(g, "tap/out", m.tap("output_block.layer.2.out").float(), 3e-4, what="tap")
    rows = r3.grad_report(m, g)
    assert len(rows) >= 360, len(rows)
    bad = []
    for n, err, l2r, cos, absmax in rows:
        if absmax < 1e-12 or r3.is_softmax_shift(n):
            continue
        if r3.is_resnet_conv_path(n):
            if abs(l2r - 1.0) > 1e-2 or cos < 0.9999:
                bad.append((n, err, l2r, cos))
        elif err > 2e-3 or abs(l2r - 1.0) > 1e-3:
            bad.append((n, err, l2r, cos))
    assert not bad, "gradient mismatch vs the reference at %s: %s" % (name, sorted(bad, key=lambda r: -r[1])[:10])


def test_long_sequence_bf16_step_and_live_rows(golden_dir):
    """S = 256 in the bf16 speed mode: loss within the bf16 band of the reference's, gradient directions as at S = 128 (cosine >= 0.98 over
    the sampled elements of every transformer / GRU / gate / classifier tensor), and the live-row step (tiled attention with per-sentence
    lengths: key tiles beyond a sentence's last row are not visited) gives the dense step's loss and layer weight gradients bit for bit."""
    import test_round3_gpu as r3
    from helpers import golden_case_inputs, load_golden
    lib = _capi.load()
    g = load_golden(golden_dir, "arch3_b8s256_train")
    cfg, sd_np, batch = golden_case_inputs(g, "arch3")
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}

    def step(live):
        lib.realise_set_engine(10, live)
        try:
            m = build(cfg, sd_np, "bf16", train=True)
            loss, _ = m(batch)
            loss.backward()
            torch.cuda.synchronize()
            return m, float(loss.item()), grads_of(m)
        finally:
            lib.realise_set_engine(10, 2)

    m1, l1, g1 = step(2)
    m0, l0, g0 = step(0)
    assert abs(l1 - float(g["loss"])) < 5e-2
    rows = [r for r in r3.grad_report(m1, g) if r[4] > 1e-9 and int(g["grad/" + r[0] + "/n"]) >= 64 and not r3.is_softmax_shift(r[0]) and not r3.is_resnet_conv_path(r[0])]
    worst = sorted(rows, key=lambda r: r[3])[:5]
    print("bf16 S = 256 worst cosines:", [(n, round(c, 4), round(l, 3)) for n, e, l, c, a in worst])
    assert worst[0][3] >= 0.98, worst
    assert l1 == l0
    layer_w = [n for n in g0 if ".layer." in n and n.endswith("weight") and "LayerNorm" not in n]
    moved = [n for n in layer_w if not torch.equal(g0[n], g1[n])]
    assert layer_w and not moved, moved[:8]


def test_host_built_batches_of_different_pinyin_width_share_one_workspace_plan():
    """build_batch (models.py:797-804) makes pho_idx as wide as the batch's longest pinyin, so the width changes from batch to batch; the
    engine's plan is keyed by (B, S, Tp) and a new key re-zeroes the workspace.  The module widens a narrower batch to the widest seen
    so far (pad columns are beyond every length: those GRU steps are not launched): same loss / logits / gradients as a module that only
    ever saw the narrow batch, and the plan key stays put."""
    cfg = RealiseConfig(num_hidden_layers=1, pho_layers=1, out_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = init_state_dict_numpy(cfg, seed=5)
    wide = synthetic_batch(4, 32, seed=3)
    narrow = synthetic_batch(4, 32, seed=4)
    Tw = wide["pho_idx"].shape[1]
    lens = np.minimum(np.asarray(narrow["pho_lens"]), 3)              # a batch whose longest pinyin has 3 letters
    narrow["pho_lens"] = [int(x) for x in lens]
    narrow["pho_idx"] = narrow["pho_idx"][:, :3].contiguous() * (torch.arange(3)[None, :] < torch.from_numpy(lens)[:, None])
    assert Tw > 3

    def run(m, b):
        m.zero_grad()
        loss, logits = m(b)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.item()), logits.detach().clone(), grads_of(m)

    m1 = build(cfg, sd, "fp32", train=True)
    run(m1, wide)
    key = m1._ws_key
    l1, z1, g1 = run(m1, narrow)
    assert m1._ws_key == key and key[2] == Tw
    m2 = build(cfg, sd, "fp32", train=True)
    l2, z2, g2 = run(m2, narrow)
    assert m2._ws_key[2] == 3
    assert l1 == l2 and torch.equal(z1, z2)
    for n in g2:
        s = g2[n].abs().max().item()
        assert (g1[n] - g2[n]).abs().max().item() <= 1e-5 * s + 1e-12, n
