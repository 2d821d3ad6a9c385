"""Per-kernel parity tests, called through the C ABI (include/realise_hip.h) on a real MI355X.

Each HIP kernel family is compared with a plain PyTorch fp32 statement of the same op on
identical seeded inputs.  Tolerances: fp32 mode (exact-fp32 MFMA) 1e-4 relative to the output
scale; bf16 mode: the comparison uses the SAME bf16-rounded inputs, fp32 math, so the residual is
one output rounding (2^-8 relative) plus accumulation-order noise -> 1.5e-2 of the output scale.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from realise_amd import _capi

pytestmark = pytest.mark.gpu

DT = {"fp32": (_capi.F32, torch.float32, 1e-4), "bf16": (_capi.BF16, torch.bfloat16, 1.5e-2)}


def dev():
    return torch.device("cuda", 0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale)


def close(out, ref, tol, what):
    out = out.float().cpu()
    ref = ref.float().cpu()
    scale = ref.abs().max().item() + 1e-12
    err = (out - ref).abs().max().item()
    assert np.isfinite(err), what + ": non-finite output"
    assert err <= tol * scale, "%s: max err %.3e > %.1e * scale %.3e" % (what, err, tol, scale)


def epilogue(mode, out, ldo, bias=None, aux=None, ldaux=0, out2=None, accumulate=0, alpha=1.0):
    e = _capi.Epilogue()
    e.mode, e.accumulate, e.out, e.ldo = mode, accumulate, out.data_ptr(), ldo
    e.out2 = out2.data_ptr() if out2 is not None else None
    e.bias = bias.data_ptr() if bias is not None else None
    e.aux = aux.data_ptr() if aux is not None else None
    e.ldaux, e.alpha, e.drop_seed, e.drop_thresh, e.drop_scale = ldaux, alpha, 0, 0, 1.0
    return e


# ------------------------------------------------------------------------------------------ GEMM NT
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (256, 768, 768), (96, 768, 2104), (512, 1000, 768),
                                   (33, 2304, 768)])
def test_gemm_nt_bias(dt, M, N, K):
    lib = _capi.load()
    code, tdt, tol = DT[dt]
    a = rnd((M, K), 1).to(dev()).to(tdt)
    b = rnd((N, K), 2).to(dev()).to(tdt)
    bias = rnd((N,), 3).to(dev())
    out = torch.full((M, N), 7.0, device=dev(), dtype=tdt)
    ep = epilogue(_capi.EPI_STORE, out, N, bias=bias)
    _capi.check(lib.realise_gemm_nt(stream(), code, P(a), K, P(b), K, M, N, K, C.byref(ep)), "gemm_nt")
    ref = a.float() @ b.float().t() + bias
    close(out, ref, tol, "gemm_nt %s %dx%dx%d" % (dt, M, N, K))


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_gemm_nt_epilogues(dt):
    lib = _capi.load()
    code, tdt, tol = DT[dt]
    M, N, K = 192, 256, 320
    a = rnd((M, K), 4, 0.5).to(dev()).to(tdt)
    b = rnd((N, K), 5, 0.1).to(dev()).to(tdt)
    bias = rnd((N,), 6).to(dev())
    aux = rnd((M, N), 7).to(dev()).to(tdt)
    pre_ref = a.float() @ b.float().t() + bias
    # GELU: out2 = pre-activation, out = erf-gelu
    out = torch.empty((M, N), device=dev(), dtype=tdt)
    out2 = torch.empty((M, N), device=dev(), dtype=tdt)
    ep = epilogue(_capi.EPI_GELU, out, N, bias=bias, out2=out2)
    _capi.check(lib.realise_gemm_nt(stream(), code, P(a), K, P(b), K, M, N, K, C.byref(ep)), "gelu")
    close(out2, pre_ref, tol, "gelu pre")
    close(out, F.gelu(pre_ref), tol, "gelu post")
    # residual (dropout off)
    ep = epilogue(_capi.EPI_DROP_RESID, out, N, bias=bias, aux=aux, ldaux=N)
    _capi.check(lib.realise_gemm_nt(stream(), code, P(a), K, P(b), K, M, N, K, C.byref(ep)), "resid")
    close(out, pre_ref + aux.float(), tol, "drop_resid")
    # accumulate
    base = rnd((M, N), 8).to(dev()).to(tdt)
    out.copy_(base)
    ep = epilogue(_capi.EPI_STORE, out, N, accumulate=1)
    _capi.check(lib.realise_gemm_nt(stream(), code, P(a), K, P(b), K, M, N, K, C.byref(ep)), "accum")
    close(out, a.float() @ b.float().t() + base.float(), tol, "accumulate")
    # gelu backward: out = acc * gelu'(aux)
    x = aux.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    ep = epilogue(_capi.EPI_GELU_BWD, out, N, aux=aux, ldaux=N)
    _capi.check(lib.realise_gemm_nt(stream(), code, P(a), K, P(b), K, M, N, K, C.byref(ep)), "gelu_bwd")
    close(out, (a.float() @ b.float().t()) * x.grad, tol, "gelu_bwd")


def test_gemm_nt_dropout_statistics():
    lib = _capi.load()
    M, N, K = 256, 512, 64
    a = torch.zeros((M, K), device=dev())
    b = torch.zeros((N, K), device=dev())
    bias = torch.ones((N,), device=dev())
    aux = torch.zeros((M, N), device=dev())
    out = torch.empty((M, N), device=dev())
    ep = epilogue(_capi.EPI_DROP_RESID, out, N, bias=bias, aux=aux, ldaux=N)
    p = 0.1
    ep.drop_seed, ep.drop_thresh, ep.drop_scale = 1234, int(p * 2 ** 32), 1.0 / (1.0 - p)
    _capi.check(lib.realise_gemm_nt(stream(), _capi.F32, P(a), K, P(b), K, M, N, K, C.byref(ep)), "drop")
    vals = out.cpu()
    kept = (vals != 0)
    assert abs(kept.float().mean().item() - (1 - p)) < 0.01                      # keep rate
    assert torch.allclose(vals[kept], torch.tensor(1.0 / (1.0 - p)))             # inverted scaling
    assert abs(kept[:, ::2].float().mean().item() - kept[:, 1::2].float().mean().item()) < 0.02
    out2 = torch.empty_like(out)
    ep.out = out2.data_ptr()
    _capi.check(lib.realise_gemm_nt(stream(), _capi.F32, P(a), K, P(b), K, M, N, K, C.byref(ep)), "drop")
    assert torch.equal(out, out2)                                                # same seed -> same mask
    ep.drop_seed = 999
    _capi.check(lib.realise_gemm_nt(stream(), _capi.F32, P(a), K, P(b), K, M, N, K, C.byref(ep)), "drop")
    assert not torch.equal(out, out2)


# ------------------------------------------------------------------------------------------ GEMM TN
@pytest.mark.parametrize("scratch", [0, 1])
@pytest.mark.parametrize("tr", [0, 1])
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("Pn,I,J", [(256, 128, 128), (1000, 136, 200), (4096, 768, 768), (300, 64, 2304), (70000, 64, 576)])
def test_gemm_tn(dt, tr, Pn, I, J, scratch):
    if dt == "fp32" and tr == 1:
        pytest.skip("transpose-read is a bf16 instruction")
    lib = _capi.load()
    code, tdt, tol = DT[dt]
    lib.realise_set_tn_transpose_read(tr)
    try:
        a = rnd((Pn, I), 11).to(dev()).to(tdt)
        b = rnd((Pn, J), 12).to(dev()).to(tdt)
        base = rnd((I, J), 13).to(dev())
        out = base.clone()
        slab = torch.empty(8 * I * J if scratch else 1, device=dev())        # split reduction: slabs+fold vs atomics
        cs = torch.full((I,), 0.5, device=dev())
        _capi.check(lib.realise_gemm_tn(stream(), code, P(a), I, P(b), J, Pn, I, J, P(out), J,
                                        P(slab) if scratch else None, slab.numel() if scratch else 0, P(cs)), "gemm_tn")
        close(cs, a.float().sum(0) + 0.5, 3e-3 if dt == "bf16" else 2e-4, "fused column sums (bias gradient)")
        ref = a.float().t() @ b.float() + base
        close(out, ref, 3e-3 if dt == "bf16" else 2e-4, "gemm_tn %s tr=%d" % (dt, tr))
    finally:
        lib.realise_set_tn_transpose_read(1)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("Pn,shapes", [(1024, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]), (300, [(136, 200), (64, 128)]), (8192, [(768, 768)])])
def test_gemm_tn_grouped(dt, Pn, shapes):
    """the one-launch form of a transformer layer's weight gradients equals the per-problem launches and torch"""
    lib = _capi.load()
    code, tdt, tol = DT[dt]
    probs = (_capi.TnProblem * len(shapes))()
    keep, outs, refs, css = [], [], [], []
    for k, (I, J) in enumerate(shapes):
        a = rnd((Pn, I), 31 + k).to(dev()).to(tdt)
        b = rnd((Pn, J), 41 + k).to(dev()).to(tdt)
        base = rnd((I, J), 51 + k).to(dev())
        out = base.clone()
        cs = torch.full((I,), 0.25, device=dev()) if k != 1 else None
        keep += [a, b]
        outs.append(out); css.append((cs, a))
        refs.append(a.float().t() @ b.float() + base)
        probs[k].A, probs[k].lda, probs[k].B, probs[k].ldb = a.data_ptr(), I, b.data_ptr(), J
        probs[k].I, probs[k].J, probs[k].out, probs[k].ldo = I, J, out.data_ptr(), J
        probs[k].colsum = cs.data_ptr() if cs is not None else None
    _capi.check(lib.realise_gemm_tn_grouped(stream(), code, len(shapes), probs, Pn), "gemm_tn_grouped")
    for k in range(len(shapes)):
        close(outs[k], refs[k], 3e-3 if dt == "bf16" else 2e-4, "grouped problem %d" % k)
        cs, a = css[k]
        if cs is not None:
            close(cs, a.float().sum(0) + 0.25, 3e-3 if dt == "bf16" else 2e-4, "grouped column sums %d" % k)
    assert lib.realise_gemm_tn_grouped(stream(), code, 5, probs, Pn) != 0          # more than 4 problems is an argument error


@pytest.mark.parametrize("shapes", [[(768, 768), (256, 384), (136, 200)], [(768, 768), (256, 384), (520, 128)]], ids=["4wave", "8wave"])
@pytest.mark.parametrize("n_live16", [0, 1, 6, 37, 70, 128])
def test_gemm_tn_grouped_over_live_16_row_blocks(n_live16, shapes):
    """the engine's weight-gradient reduction over the LIVE rows of a padded batch (bf16): the rows of the dead 16-row blocks are exact
    zeros in dY, the list names the live blocks, four of them (any four) form a reduction tile - equal to the dense reduction; also in
    the whole-tile form (list_rows = 64) and with overwrite."""
    lib = _capi.load()
    code, tdt, tol = DT["bf16"]
    # (every problem with I >= 256 and J >= 128: the 8-wave 256 x 128 kernel, gemm_tn8_group; otherwise the 4-wave 128 x 128 one)
    Pn = 2048
    g = torch.Generator().manual_seed(100 + n_live16)
    live = torch.sort(torch.randperm(Pn // 16, generator=g)[:n_live16]).values.int()
    rowmask = torch.zeros(Pn, 1)
    for b in live.tolist():
        rowmask[16 * b:16 * b + 16] = 1.0
    for list_rows, overwrite in ((16, 0), (16, 1), (64, 0)):
        if list_rows == 64:
            lst = torch.unique(live // 4).int()
        else:
            lst = live
        probs = (_capi.TnProblem * len(shapes))()
        keep, outs, refs, css = [], [], [], []
        for k, (I, J) in enumerate(shapes):
            a = (rnd((Pn, I), 31 + k) * rowmask).to(dev()).to(tdt)
            b = rnd((Pn, J), 41 + k).to(dev()).to(tdt)
            base = rnd((I, J), 51 + k).to(dev())
            out = base.clone()
            cs = torch.full((I,), 0.25, device=dev())
            keep += [a, b]
            outs.append(out); css.append((cs, a))
            refs.append(a.float().t() @ b.float() + (0.0 if overwrite else 1.0) * base)
            probs[k].A, probs[k].lda, probs[k].B, probs[k].ldb = a.data_ptr(), I, b.data_ptr(), J
            probs[k].I, probs[k].J, probs[k].out, probs[k].ldo = I, J, out.data_ptr(), J
            probs[k].colsum = cs.data_ptr()
        lst_d = torch.cat([lst, torch.full((8,), 10 ** 6, dtype=torch.int32)]).to(dev())      # poison behind the count: must never be read as a block
        n_d = torch.tensor([lst.numel()], dtype=torch.int32, device=dev())
        _capi.check(lib.realise_gemm_tn_grouped_live(stream(), code, len(shapes), probs, Pn, P(lst_d), P(n_d), list_rows, overwrite), "gemm_tn_grouped_live")
        for k in range(len(shapes)):
            close(outs[k], refs[k], 3e-3, "live-list problem %d rows %d overwrite %d" % (k, list_rows, overwrite))
            cs, a = css[k]
            close(cs, a.float().sum(0) + 0.25, 3e-3, "live-list column sums %d" % k)


# ------------------------------------------------------------------------------------------ conv
def _geom(src, index, rows, Hr, Hs, Cc, k, stride, pad, mode):
    g = _capi.ConvGeom()
    g.src = src.data_ptr()
    g.img_index = index.data_ptr() if index is not None else None
    g.rows, g.Hr, g.Wr, g.Hs, g.Ws, g.C, g.KH, g.KW, g.stride, g.pad, g.mode = rows, Hr, Hr, Hs, Hs, Cc, k, k, stride, pad, mode
    return g


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("Ci,Cp,Co,Hin,k,stride,pad,use_index", [
    (3, 8, 64, 32, 3, 2, 1, True), (3, 8, 64, 32, 1, 2, 0, True), (64, 64, 64, 16, 3, 1, 1, False),
    (64, 64, 128, 16, 3, 2, 1, False), (128, 128, 256, 2, 3, 2, 1, False), (256, 256, 256, 1, 3, 1, 1, False),
    (64, 64, 128, 8, 1, 2, 0, False), (128, 128, 256, 4, 3, 2, 1, False)])
def test_conv_forward_dgrad_wgrad(dt, Ci, Cp, Co, Hin, k, stride, pad, use_index):
    """implicit-im2col conv forward / input gradient / weight gradient vs torch.nn.functional.conv2d"""
    lib = _capi.load()
    code, tdt, tol = DT[dt]
    N, Vt = 6, 10
    Hout = (Hin + 2 * pad - k) // stride + 1
    table = rnd((Vt if use_index else N, Ci, Hin, Hin), 21).to(dev())
    idx = torch.tensor([3, 0, 9, 3, 7, 1], device=dev(), dtype=torch.int64) if use_index else None
    w = rnd((Co, Ci, k, k), 22, 0.2).to(dev())
    # operands in the kernel's layouts (NHWC with padded channels; [Co][tap][Cp]; [Cp][tap][Co])
    x_nhwc = torch.zeros((table.shape[0], Hin, Hin, Cp), device=dev())
    x_nhwc[..., :Ci] = table.permute(0, 2, 3, 1)
    x_t = x_nhwc.to(tdt).contiguous()
    wf = torch.zeros((Co, k * k, Cp), device=dev())
    wf[..., :Ci] = w.permute(0, 2, 3, 1).reshape(Co, k * k, Ci)
    wf_t = wf.to(tdt).contiguous()
    wd = torch.zeros((Cp, k * k, Co), device=dev())
    wd[:Ci] = w.permute(1, 2, 3, 0).reshape(Ci, k * k, Co)
    wd_t = wd.to(tdt).contiguous()
    # torch reference on the rounded operands
    xr = x_t.float()[..., :Ci].permute(0, 3, 1, 2)
    if use_index:
        xr = xr.index_select(0, idx)
    xr = xr.contiguous().requires_grad_(True)
    wr = wf_t.float()[..., :Ci].reshape(Co, k, k, Ci).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, stride=stride, padding=pad)
    Pn = N * Hout * Hout
    y = torch.empty((Pn, Co), device=dev(), dtype=tdt)
    g = _geom(x_t, idx, Pn, Hout, Hin, Cp, k, stride, pad, 0)
    ep = epilogue(_capi.EPI_STORE, y, Co)
    _capi.check(lib.realise_conv_nt(stream(), code, C.byref(g), P(wf_t), k * k * Cp, Pn, Co, k * k * Cp, C.byref(ep)), "conv fwd")
    close(y.view(N, Hout, Hout, Co), y_ref.permute(0, 2, 3, 1), tol, "conv fwd")
    # backward
    dy = rnd((N, Co, Hout, Hout), 23).to(dev())
    dy_t = dy.permute(0, 2, 3, 1).reshape(Pn, Co).to(tdt).contiguous()
    y_ref.backward(dy_t.float().view(N, Hout, Hout, Co).permute(0, 3, 1, 2))
    if not use_index:
        Pin = N * Hin * Hin
        dx = torch.empty((Pin, Cp), device=dev(), dtype=tdt)
        g2 = _geom(dy_t, None, Pin, Hin, Hout, Co, k, stride, pad, 1)
        ep2 = epilogue(_capi.EPI_STORE, dx, Cp)
        _capi.check(lib.realise_conv_nt(stream(), code, C.byref(g2), P(wd_t), k * k * Co, Pin, Cp, k * k * Co, C.byref(ep2)), "conv dgrad")
        close(dx.view(N, Hin, Hin, Cp)[..., :Ci], xr.grad.permute(0, 2, 3, 1), tol, "conv dgrad")
        if stride == 2 and Hin & (Hin - 1) == 0 and Hin >= 2:
            # the same gradient by parity classes of the input pixel (realise_conv_dgrad_s2): class c = 2*py + px uses the taps
            # kh = ((py + pad) & 1) + 2i, kw likewise; the weight copy holds the tap slots class by class
            order = []
            for c in range(4):
                kh0, kw0 = ((c >> 1) + pad) & 1, ((c & 1) + pad) & 1
                order += [kh * k + kw for kh in range(kh0, k, 2) for kw in range(kw0, k, 2)]
            assert sorted(order) == list(range(k * k))
            wd_c = wd[:, order, :].to(tdt).contiguous()
            dx2 = torch.full((Pin, Cp), 7.0, device=dev(), dtype=tdt)
            ep3 = epilogue(_capi.EPI_STORE, dx2, Cp)
            if k == 1:                                   # only (even, even) pixels are reached: accumulate onto zeros
                dx2.zero_()
                ep3.accumulate = 1
            _capi.check(lib.realise_conv_dgrad_s2(stream(), code, C.byref(g2), P(wd_c), k * k * Co, Cp, C.byref(ep3)), "conv dgrad by parity classes")
            close(dx2.view(N, Hin, Hin, Cp)[..., :Ci], xr.grad.permute(0, 2, 3, 1), tol, "conv dgrad by parity classes")
            if dt == "fp32" and k == 3:
                assert torch.equal(dx2, dx), "class-wise and full-tap data gradients differ (same products, zero taps removed)"
    dw = torch.zeros((Co, Ci, k, k), device=dev())
    g3 = _geom(x_t, idx, Pn, Hout, Hin, Cp, k, stride, pad, 0)
    slab = torch.empty(4 * Co * k * k * Cp, device=dev())
    _capi.check(lib.realise_conv_tn(stream(), code, P(dy_t), Co, C.byref(g3), Pn, Co, Ci, P(dw), P(slab), slab.numel()), "conv wgrad")
    close(dw, wr.grad, 3e-3 if dt == "bf16" else tol, "conv wgrad")


# ------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, mask_add, nh):
    B, S, H = q.shape
    d = H // nh

    def sp(t):
        return t.view(B, S, nh, d).permute(0, 2, 1, 3)
    s = sp(q) @ sp(k).transpose(-1, -2) / 8.0 + mask_add[:, None, None, :]
    p = torch.softmax(s, -1)
    return (p @ sp(v)).permute(0, 2, 1, 3).reshape(B, S, H)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("B,nh,S", [(2, 3, 16), (2, 2, 40), (3, 12, 128), (1, 1, 97),
                                    (2, 3, 256), (1, 2, 200), (2, 12, 512), (1, 1, 129), (2, 2, 301)])      # S > 128: the tiled kernels
def test_attention_fwd_bwd(dt, B, nh, S):
    lib = _capi.load()
    code, tdt, tol = DT[dt]
    H = nh * 64
    qkv = rnd((B * S, 3 * H), 31, 1.0).to(dev()).to(tdt)
    masks = torch.ones((B, S), dtype=torch.int64)
    for b in range(B):
        masks[b, S - (b * 5) % max(S // 2, 1):] = 0 if b > 0 else 1
    masks = masks.to(dev())
    madd = torch.empty((B, S), device=dev())
    _capi.check(lib.realise_mask_to_additive(stream(), P(masks), P(madd), B * S), "mask")
    assert torch.equal(madd, (1.0 - masks.float()) * -10000.0)
    ctx = torch.empty((B * S, H), device=dev(), dtype=tdt)
    lse = torch.empty((B, nh, S), device=dev())
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    esz = qkv.element_size()
    _capi.check(lib.realise_attention_fwd(stream(), code, P(qkv), C.c_void_p(qkv.data_ptr() + H * esz),
                                          C.c_void_p(qkv.data_ptr() + 2 * H * esz), 3 * H, P(madd), P(ctx), H, P(lse),
                                          B, nh, S, 0, 0, 1.0), "attn fwd")
    qf = q.float().reshape(B, S, H).clone().requires_grad_(True)
    kf = k.float().reshape(B, S, H).clone().requires_grad_(True)
    vf = v.float().reshape(B, S, H).clone().requires_grad_(True)
    ref = _attn_ref(qf, kf, vf, madd, nh)
    close(ctx.view(B, S, H), ref, tol, "attention fwd")
    dctx = rnd((B * S, H), 32).to(dev()).to(tdt)
    ref.backward(dctx.float().view(B, S, H))
    dqkv = torch.full((B * S, 3 * H), 5.0, device=dev(), dtype=tdt)
    rowdot = torch.empty((B, nh, S), device=dev())
    _capi.check(lib.realise_attention_bwd(stream(), code, P(qkv), C.c_void_p(qkv.data_ptr() + H * esz),
                                          C.c_void_p(qkv.data_ptr() + 2 * H * esz), 3 * H, P(madd), P(ctx), P(dctx), H, P(lse),
                                          P(rowdot), P(dqkv), C.c_void_p(dqkv.data_ptr() + H * esz),
                                          C.c_void_p(dqkv.data_ptr() + 2 * H * esz), 3 * H, B, nh, S, 0, 0, 1.0), "attn bwd")
    btol = 3e-2 if dt == "bf16" else 2e-4
    close(dqkv[:, 2 * H:].reshape(B, S, H), vf.grad, btol, "attention dV")
    close(dqkv[:, H:2 * H].reshape(B, S, H), kf.grad, btol, "attention dK")
    close(dqkv[:, :H].reshape(B, S, H), qf.grad, btol, "attention dQ")


@pytest.mark.parametrize("S", [48, 45, 160, 261])
def test_attention_dropout_backward_consistency(S):
    """with dropout ON the backward must use the same regenerated mask as the forward:
    directional finite differences of sum(ctx * R) in exact-fp32 mode.  S % 4 == 0: one dropout hash per four keys (forward, dQ) and
    the quad exchange between the four lanes of a key quad (dK / dV); S = 45: the per-element form; S = 160 / 261: the tiled kernels
    (two and three tiles of 128 keys / queries; the hash index is the absolute (query, key) pair in every kernel)."""
    lib = _capi.load()
    B, nh = 1, 2
    H = nh * 64
    qkv = rnd((B * S, 3 * H), 41, 0.5).to(dev())
    madd = torch.zeros((B, S), device=dev())
    R = rnd((B * S, H), 42).to(dev())
    seed, thresh, scale = 77, int(0.1 * 2 ** 32), 1.0 / 0.9

    def fwd(x):
        ctx = torch.empty((B * S, H), device=dev())
        lse = torch.empty((B, nh, S), device=dev())
        _capi.check(lib.realise_attention_fwd(stream(), _capi.F32, P(x), C.c_void_p(x.data_ptr() + H * 4),
                                              C.c_void_p(x.data_ptr() + 2 * H * 4), 3 * H, P(madd), P(ctx), H, P(lse),
                                              B, nh, S, seed, thresh, scale), "fwd")
        return ctx, lse
    ctx, lse = fwd(qkv)
    dqkv = torch.empty_like(qkv)
    rowdot = torch.empty((B, nh, S), device=dev())
    _capi.check(lib.realise_attention_bwd(stream(), _capi.F32, P(qkv), C.c_void_p(qkv.data_ptr() + H * 4),
                                          C.c_void_p(qkv.data_ptr() + 2 * H * 4), 3 * H, P(madd), P(ctx), P(R), H, P(lse),
                                          P(rowdot), P(dqkv), C.c_void_p(dqkv.data_ptr() + H * 4),
                                          C.c_void_p(dqkv.data_ptr() + 2 * H * 4), 3 * H, B, nh, S, seed, thresh, scale), "bwd")
    direction = rnd(qkv.shape, 43).to(dev())
    eps = 1e-2
    lp = (fwd(qkv + eps * direction)[0].double() * R.double()).sum()
    lm = (fwd(qkv - eps * direction)[0].double() * R.double()).sum()
    fd = ((lp - lm) / (2 * eps)).item()
    an = (dqkv.double() * direction.double()).sum().item()
    assert abs(fd - an) <= 2e-3 * max(1.0, abs(fd)), (fd, an)
    nodrop = torch.empty((B * S, H), device=dev())
    _capi.check(lib.realise_attention_fwd(stream(), _capi.F32, P(qkv), C.c_void_p(qkv.data_ptr() + H * 4),
                                          C.c_void_p(qkv.data_ptr() + 2 * H * 4), 3 * H, P(madd), P(nodrop), H, P(lse),
                                          B, nh, S, 0, 0, 1.0), "fwd")
    assert not torch.allclose(nodrop, ctx)


# ------------------------------------------------------------------------------------------ LN / CE
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("rows,H", [(37, 768), (512, 768), (9, 192)])
def test_layernorm_fwd_bwd(dt, rows, H):
    lib = _capi.load()
    code, tdt, tol = DT[dt]
    x = (rnd((rows, H), 51) * 2 + 0.5).to(dev()).to(tdt)
    gamma = (1 + 0.1 * rnd((H,), 52)).to(dev())
    beta = (0.1 * rnd((H,), 53)).to(dev())
    y = torch.empty_like(x)
    xhat = torch.empty_like(x)
    rstd = torch.empty((rows,), device=dev())
    _capi.check(lib.realise_layernorm_fwd(stream(), code, P(x), P(gamma), P(beta), 1e-12, P(y), P(xhat), P(rstd), rows, H), "ln")
    xr = x.float().clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (H,), gr, br, 1e-12)
    close(y, ref, tol, "ln fwd")
    dy = rnd((rows, H), 54).to(dev()).to(tdt)
    ref.backward(dy.float())
    dx = torch.empty_like(x)
    dg = torch.zeros((H,), device=dev())
    db = torch.zeros((H,), device=dev())
    _capi.check(lib.realise_layernorm_bwd(stream(), code, P(dy), P(xhat), P(rstd), P(gamma), P(dx), P(dg), P(db), rows, H), "lnb")
    close(dx, xr.grad, 2e-2 if dt == "bf16" else 1e-4, "ln dx")
    close(dg, gr.grad, 2e-2 if dt == "bf16" else 1e-4, "ln dgamma")
    close(db, br.grad, 1e-2 if dt == "bf16" else 1e-4, "ln dbeta")


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_masked_cross_entropy(dt):
    lib = _capi.load()
    code, tdt, tol = DT[dt]
    rows, V = 50, 21128
    logits = (rnd((rows, V), 61) * 0.6).to(dev()).to(tdt)
    labels = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(62)).to(dev())
    lm = (torch.arange(rows) % 3 != 0).long().to(dev())
    loss = torch.zeros((), device=dev())
    cnt = torch.zeros((1,), device=dev())
    dl = torch.empty_like(logits)
    _capi.check(lib.realise_masked_ce(stream(), code, P(logits), V, P(labels), P(lm), rows, V, P(loss), P(cnt), P(dl)), "ce")
    lr = logits.float().clone().requires_grad_(True)
    ref = F.cross_entropy(lr[lm == 1], labels[lm == 1])
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    close(dl, lr.grad, 1e-2 if dt == "bf16" else 1e-4, "ce dlogits")
    assert cnt.item() == float((lm == 1).sum().item())


def test_adamw_and_gradnorm_kernels(golden_dir):
    """fused AdamW over a flat arena vs the reference trajectory in tests/golden/adamw_steps.npz"""
    import os
    lib = _capi.load()
    g = dict(np.load(os.path.join(golden_dir, "adamw_steps.npz")))
    p = torch.from_numpy(g["p0"].copy()).to(dev()).reshape(-1)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for i in range(3):
        gr = torch.from_numpy(g["grads"][i]).to(dev()).reshape(-1).contiguous()
        _capi.check(lib.realise_adamw(stream(), P(p), P(gr), P(m), P(v), p.numel(), float(g["lrs"][i]), 0.9, 0.999, 1e-8, 0.01,
                                      i + 1, 1, None, 0.0), "adamw")
        assert np.abs(p.cpu().numpy().reshape(5, 7) - g["traj"][i]).max() < 1e-6
    big = rnd((100003,), 71).to(dev())
    acc = torch.zeros((1,), device=dev())
    _capi.check(lib.realise_sumsq(stream(), P(big), big.numel(), P(acc)), "sumsq")
    assert abs(acc.item() - (big.double() ** 2).sum().item()) < 2e-6 * acc.item()


# ---- eval decode (run.py:262-263): device argmax, first maximum wins like numpy -------------------------------------
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("rows,V,ld", [(37, 21128, 21128), (5, 1000, 1003), (64, 7, 8), (3, 70000, 70000)])
def test_argmax_matches_numpy_first_occurrence(dtype, rows, V, ld):
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(rows * 7 + V)
    x = torch.randn((rows, ld), generator=g).to(tdt)
    x[0, : min(V, 5)] = 3.0                                   # ties on purpose (bf16 rounds many values together anyway)
    if rows > 2:
        x[2, V - 1] = 100.0                                   # maximum in the ragged tail
    xd = x.cuda()
    ids = torch.empty(rows, dtype=torch.int64, device="cuda")
    _capi.check(_capi.load().realise_argmax(stream(), 0 if dtype == "fp32" else 1, P(xd), ld, rows, V, P(ids)), "realise_argmax")
    torch.cuda.synchronize()
    want = np.argmax(x[:, :V].float().numpy(), axis=-1)
    assert np.array_equal(ids.cpu().numpy(), want)


def test_argmax_nan_counts_as_maximum_like_numpy():
    x = torch.randn(4, 512)
    x[1, 77] = float("nan")
    x[1, 300] = float("nan")
    xd = x.cuda()
    ids = torch.empty(4, dtype=torch.int64, device="cuda")
    _capi.check(_capi.load().realise_argmax(stream(), 0, P(xd), 512, 4, 512, P(ids)), "realise_argmax")
    torch.cuda.synchronize()
    assert np.array_equal(ids.cpu().numpy(), np.argmax(x.numpy(), axis=-1))


# ---- device-side build_batch (models.py:797-804): gather + stable length sort + alive counts ------------------------
@pytest.mark.parametrize("T_", [1, 37, 1024, 8192, 20000])
def test_build_pho_matches_host_bookkeeping(T_):
    V, Tw = 500, 7
    g = np.random.default_rng(T_)
    vlens = g.integers(1, Tw + 1, V).astype(np.int32)
    table = np.zeros((V, Tw), np.int64)
    for v in range(V):
        table[v, :vlens[v]] = g.integers(1, 33, vlens[v])
    src = g.integers(0, V, T_).astype(np.int64)
    if T_ > 8:
        src[3:8] = 0                                             # equal lengths next to each other: stability matters
    d = lambda a: torch.from_numpy(a).cuda()
    srcd, tabd, vld = d(src), d(table), d(vlens)
    pho = torch.empty((T_, Tw), dtype=torch.int64, device="cuda")
    perm = torch.empty(T_, dtype=torch.int32, device="cuda")
    ls = torch.empty(T_, dtype=torch.int32, device="cuda")
    alive = torch.empty(Tw, dtype=torch.int32, device="cuda")
    _capi.check(_capi.load().realise_build_pho(stream(), P(srcd), T_, P(tabd), P(vld), V, Tw, P(pho), P(perm), P(ls), P(alive)), "build_pho")
    torch.cuda.synchronize()
    lens = vlens[src]
    want_perm = np.argsort(-lens, kind="stable").astype(np.int32)          # what modeling.py does on the host
    assert np.array_equal(pho.cpu().numpy(), table[src])
    assert np.array_equal(perm.cpu().numpy(), want_perm)
    assert np.array_equal(ls.cpu().numpy(), lens[want_perm])
    assert alive.cpu().tolist() == [int((lens > t).sum()) for t in range(Tw)]


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 14, 16, 50])
def test_gemm_nt_tile_variants(variant):
    """realise_set_nt_variant: every NT kernel the library holds must give the default choice's results.  The production library
    ships 9 (4-wave), 12 (8-wave 256x192), 14 (8-wave 128x192, three stages), 16 (8-wave 128x192, two workgroups per CU) and 50
    (persistent 256x192); the measured-and-
    rejected shapes 1..8 (8-wave 128x192 / 256x128 tiles of the 4-wave family, 3-stage rings, spread fetch issue, the phase-shifted
    two-group kernel) exist in the probe build only (REALISE_HIP_PROBES=1)."""
    lib = _capi.load()
    if variant < 9 and b"+probes" not in lib.realise_version():
        pytest.skip("probe build only")
    code, tdt, tol = DT["bf16"]
    outs = []
    try:
        for v in (0, variant):
            lib.realise_set_nt_variant(v)
            for (M, N, K) in ((520, 768, 768), (256, 3072, 128)):
                a, b = rnd((M, K), 1).to(tdt).cuda(), rnd((N, K), 2, 0.05).to(tdt).cuda()
                bias = rnd((N,), 3).cuda()
                out = torch.empty((M, N), dtype=tdt, device="cuda")
                ep = _capi.Epilogue()
                ep.mode, ep.out, ep.ldo, ep.bias, ep.alpha, ep.drop_scale = 0, out.data_ptr(), N, bias.data_ptr(), 1.0, 1.0
                _capi.check(lib.realise_gemm_nt(stream(), code, P(a), K, P(b), K, M, N, K, C.byref(ep)), "gemm_nt")
                torch.cuda.synchronize()
                outs.append(out.float().cpu())
    finally:
        lib.realise_set_nt_variant(0)
    for ref, got in zip(outs[:2], outs[2:]):
        assert torch.allclose(ref, got, rtol=1e-2, atol=1e-3)       # same math; at most a bf16 rounding apart
