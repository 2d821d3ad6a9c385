"""GPU tests added in round 6: workspace slots (a plan per batch shape, zero-filled once), the trainer's run.py:193-230 call sites
(gradient accumulation, windowed loss log, checkpoints every save_steps), live-block lists beyond 16384 token rows,
and this round's kernels / fusions."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from realise_amd import _capi, trainer
from realise_amd.config import RealiseConfig
from realise_amd.data import synthetic_batch
from realise_amd.init import init_state_dict_numpy
from realise_amd.modeling import SpellBertPho2ResArch3
from realise_amd.optim import FusedAdamW, get_linear_schedule_with_warmup

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


SMALL = dict(num_hidden_layers=2, pho_layers=1, out_layers=1)


# ---------------------------------------------------------------------------------------------- workspace slots
def test_alternating_batch_shapes_zero_fill_each_workspace_once():
    """a plan change re-zeroed the whole workspace (2.7 GB at the bench shape) on every (B, S, Tp) switch.
    Now every shape owns a workspace buffer (model.workspace_slots most recent ones) and the engine remembers the plan installed in
    it: train (B = 8) -> eval (B = 4) -> train -> eval -> the short last batch (B = 3) counts THREE installs, and the results of the
    revisited shapes are those of a module that never left them (the self-cleaning accumulators and stale finite rows of a plan
    survive in its own buffer)."""
    lib = _capi.load()
    cfg = RealiseConfig(**SMALL, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = init_state_dict_numpy(cfg, seed=3)
    tb, eb, lb = cuda_batch(8, 64, 31), cuda_batch(4, 64, 32), cuda_batch(3, 64, 33)

    def train_step(m, b):
        m.train()
        m.zero_grad()
        loss, _ = m(b)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.item()), grads_of(m)

    def eval_step(m, b):
        m.eval()
        with torch.no_grad():
            loss, logits = m(b)
        return float(loss.item()), logits.float().clone()

    m = build(cfg, sd, "bf16", train=True)
    l0, g0 = train_step(m, tb)
    assert lib.realise_engine_plan_installs(m._engine) == 1
    e0 = eval_step(m, eb)
    assert lib.realise_engine_plan_installs(m._engine) == 2
    e1 = eval_step(m, eb)                  # (before the next training forward moves the BatchNorm running statistics)
    l1, g1 = train_step(m, tb)
    train_step(m, lb)
    assert lib.realise_engine_plan_installs(m._engine) == 3
    l2, g2 = train_step(m, tb)
    assert lib.realise_engine_plan_installs(m._engine) == 3, "a revisited shape must not be re-installed"
    assert len(m._ws_cache) == 3
    assert l0 == l1 == l2 and e0[0] == e1[0] and torch.equal(e0[1], e1[1])
    ref = build(cfg, sd, "bf16", train=True)
    lr, gr = train_step(ref, tb)
    assert lr == l2
    for n in gr:
        s = gr[n].abs().max().item()
        assert (gr[n] - g2[n]).abs().max().item() <= 5e-5 * s + 1e-12, n
    # a fourth shape evicts the least recently used buffer (the engine is told before the buffer is freed) and everything still works
    train_step(m, cuda_batch(2, 64, 34))
    assert len(m._ws_cache) == 3 and lib.realise_engine_plan_installs(m._engine) == 4
    eval_step(m, eb)                       # (eb's buffer was the least recently used one: evicted, re-installed)
    assert lib.realise_engine_plan_installs(m._engine) == 5
    l3, _ = train_step(m, tb)
    assert l3 == l0


# ---------------------------------------------------------------------------------------------- trainer call sites (run.py:193-230)
def _items(n, S, seed):
    b = synthetic_batch(n, S, seed=seed)
    items = []
    for i in range(n):
        L = int(b["loss_masks"][i].sum())
        items.append({"id": "s%d" % i, "src": None, "tgt": None, "tokens_size": [1] * L, "lengths": L,
                      "src_idx": b["src_idx"][i, :L + 2].tolist(), "tgt_idx": b["tgt_idx"][i, :L + 2].tolist()})
    return items


class _Tok:
    """build_batch only needs convert_ids_to_tokens for the pinyin lookup; the device-side table replaces it here"""


def _bb(batch, tokenizer):
    return batch


def test_trainer_reproduces_accumulation_logging_and_checkpoint_call_sites(tmp_path):
    """run.py:193-230 transcribed by hand against trainer.train(): with gradient_accumulation_steps = 2 the loss is halved before
    backward(), the optimizer steps every second batch, t_total counts optimizer steps, `Step / LR / Loss` lines carry the windowed
    mean (tr_loss - logging_loss) / logging_steps, and every save_steps a saved_ckpt-N directory appears with the model files and
    training_args.bin.  Dropout off: the first windows agree to fp32 summation order; after the first update with a non-zero rate the
    two runs agree to ~1e-5 (the few gradient sums behind float atomics - embedding rows, bias column sums - differ in their last bits
    between any two runs, and Adam's m / sqrt(v) turns a last-bit difference of a near-zero gradient element into a full +-lr step)."""
    from realise_amd.data import synthetic_pinyin_table
    cfg = RealiseConfig(**SMALL, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = init_state_dict_numpy(cfg, seed=9)
    S, bs, gas = 64, 4, 2
    items = _items(24, S, 77)
    ptable = synthetic_pinyin_table(cfg.vocab_size)

    def fresh():
        m = build(cfg, sd, "bf16", train=True)
        m.set_pinyin_table(ptable)
        return m

    # ---- the trainer
    m1 = fresh()
    lines = []
    out_dir = str(tmp_path / "out")
    args_obj = {"per_gpu_train_batch_size": bs, "gradient_accumulation_steps": gas, "learning_rate": 1e-4}
    gs, mean1 = trainer.train(m1, items, batch_size=bs, max_seq_length=S, epochs=1, lr=1e-4, warmup_steps=1, build_batch=_bb,
                              gradient_accumulation_steps=gas, logging_steps=1, save_steps=2, output_dir=out_dir, training_args=args_obj,
                              log_fn=lines.append, seed=5, return_global_step=True)
    assert gs == 24 // bs // gas == 3
    assert len(lines) == 3 and all(l.startswith("Step: %d, LR: " % (i + 1)) for i, l in enumerate(lines))
    ck = os.path.join(out_dir, "saved_ckpt-2")
    assert os.path.isdir(ck) and not os.path.exists(os.path.join(out_dir, "saved_ckpt-1")) and not os.path.exists(os.path.join(out_dir, "saved_ckpt-3"))
    assert sorted(os.listdir(ck)) == ["config.json", "pytorch_model.bin", "training_args.bin"]
    assert torch.load(os.path.join(ck, "training_args.bin"), weights_only=False) == args_obj
    assert m1.trust_fused_optimizer is False

    # ---- run.py:184-221 by hand on a second module
    m2 = fresh()
    m2.trust_fused_optimizer = True
    no_decay = ["bias", "LayerNorm.weight"]
    groups = [{"params": [p for n, p in m2.named_parameters() if p.requires_grad and not any(nd in n for nd in no_decay)], "weight_decay": 0.0},
              {"params": [p for n, p in m2.named_parameters() if p.requires_grad and any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    opt = FusedAdamW(m2, groups, lr=1e-4, eps=1e-8, max_grad_norm=1.0)
    t_total = len(items) // bs // gas * 1
    sched = get_linear_schedule_with_warmup(opt, 1, t_total)
    tr_loss, logging_loss, global_step, want = 0.0, 0.0, 0, []
    m2.zero_grad()
    for step, batch in enumerate(trainer.data_helper(items, bs, S, _bb, None, seed=5 + 0)):
        m2.train()
        for t in batch:
            if t not in ["id", "src", "tgt", "lengths", "tokens_size", "pho_lens"]:
                batch[t] = batch[t].to("cuda")
        loss = m2(batch)[0]
        if gas > 1:
            loss = loss / gas
        loss.backward()
        tr_loss += loss.item()
        if (step + 1) % gas == 0:
            opt.step()
            sched.step()
            m2.zero_grad()
            global_step += 1
            want.append((global_step, sched.get_last_lr()[0], (tr_loss - logging_loss) / 1))
            logging_loss = tr_loss
    assert global_step == gs
    print("trainer lines:", lines, "by hand:", want, "means:", mean1, tr_loss / global_step)
    for line, (g_, lr_, l_) in zip(lines, want):
        f = line.replace(",", "").split()
        # (later windows: the two loops' backward passes add their fp32 atomics in another order - the trainer skips the training logits,
        # its timing differs - and bf16 forwards turn that 1e-8 noise in the weights into ~1e-4 of a loss within two updates)
        assert int(f[1]) == g_ and abs(float(f[3]) - lr_) < 1e-12 and abs(float(f[5]) - l_) < 1e-3 * abs(l_) + 1e-6, (line, g_, lr_, l_)
    f1 = lines[0].replace(",", "").split()
    assert abs(float(f1[5]) - want[0][2]) < 1e-6 * abs(want[0][2])         # the first window: same weights, same batches, no update yet
    assert abs(mean1 - tr_loss / global_step) < 1e-3 * abs(mean1)
    torch.cuda.synchronize()
    dp = (m1.flat_parameters() - m2.flat_parameters()).abs()
    assert dp.max().item() <= 4.1e-4 and dp.mean().item() < 1e-6, (dp.max().item(), dp.mean().item())      # (two updates of at most lr each)
    # the checkpoint of step 2 loads into a fresh module (from_pretrained round trip, run.py:468 / 520)
    m3 = SpellBertPho2ResArch3.from_pretrained(ck, config=cfg, compute_dtype="bf16")
    assert set(m3.state_dict()) == set(m1.state_dict())


def test_trainer_restores_the_trust_flag_when_the_loop_raises():
    """an exception inside train() left model.trust_fused_optimizer = True (the safe default silently lost)."""
    cfg = RealiseConfig(**SMALL)
    m = build(cfg, init_state_dict_numpy(cfg, seed=9), "bf16", train=True)
    items = _items(8, 64, 78)
    items[5]["src_idx"][3] = cfg.vocab_size + 5               # nn.Embedding would raise IndexError (modeling_bert.py:183-186)
    m.strict_ids = True

    def bb(batch, tok):
        b = synthetic_batch(len(batch["src_idx"]), 64, seed=1)
        batch["pho_idx"], batch["pho_lens"] = b["pho_idx"], b["pho_lens"]
        return batch

    with pytest.raises(IndexError):
        trainer.train(m, items, batch_size=4, max_seq_length=64, build_batch=bb, seed=1)
    assert m.trust_fused_optimizer is False


# ---------------------------------------------------------------------------------------------- live-block lists beyond 16384 rows
def test_live_row_step_at_32768_token_rows_matches_the_dense_step():
    """the live-block list of the grouped weight gradients lived in a fixed 4 KB of LDS = 16384 token rows; B = 256,
    S = 128 failed in backward.  The list area is now sized per launch (up to 65536 rows; beyond that the step is dense)."""
    lib = _capi.load()
    cfg = RealiseConfig(num_hidden_layers=1, pho_layers=1, out_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = init_state_dict_numpy(cfg, seed=13)
    batch = cuda_batch(256, 128, 41)

    def step(on):
        lib.realise_set_engine(10, on)
        lib.realise_set_engine(5, 1 if on else 0)
        try:
            m = build(cfg, sd, "bf16", train=True)
            m.zero_grad()
            loss, _ = m(batch)
            loss.backward()
            torch.cuda.synchronize()
            return float(loss.item()), grads_of(m)
        finally:
            lib.realise_set_engine(10, 2)
            lib.realise_set_engine(5, 1)

    (l0, g0), (l1, g1) = step(0), step(1)
    assert l0 == l1
    for n in g0:
        s = g0[n].abs().max().item()
        assert (g0[n] - g1[n]).abs().max().item() <= 5e-5 * s + 1e-12, n


# ---------------------------------------------------------------------------------------------- row-granular live GEMMs
def _epilogue(mode, out, N, accumulate=0, out2=None, bias=None, aux=None, drop=0.0, seed=1234):
    ep = _capi.Epilogue()
    ep.mode, ep.accumulate, ep.out, ep.ldo, ep.alpha, ep.drop_scale = mode, accumulate, out.data_ptr(), N, 1.0, 1.0
    if out2 is not None:
        ep.out2 = out2.data_ptr()
    if bias is not None:
        ep.bias = bias.data_ptr()
    if aux is not None:
        ep.aux, ep.ldaux = aux.data_ptr(), N
    if drop > 0.0:
        ep.drop_seed, ep.drop_thresh, ep.drop_scale = seed, int(drop * 4294967296.0), 1.0 / (1.0 - drop)
    return ep


@pytest.mark.parametrize("M,N,K,mode,accumulate,kind", [
    (8192, 2304, 768, 0, 0, "ragged"),          # qkv
    (8192, 3072, 768, 1, 0, "ragged"),          # FFN-up + GELU (+ the pre-activation copy)
    (8192, 768, 3072, 2, 0, "ragged"),          # FFN-down + dropout + residual
    (8192, 3072, 768, 4, 0, "ragged"),          # GELU' data gradient
    (8192, 768, 2304, 0, 1, "ragged"),          # accumulating data gradient
    (1024, 768, 768, 2, 0, "one"),              # a single listed row
    (1024, 768, 768, 0, 1, "none"),             # an empty list: nothing is touched
    (2048, 768, 768, 0, 0, "all"),              # every row listed: the dense product
    (512, 192, 128, 0, 0, "odd"),               # every other row
    (1024, 768, 768, 0, 0, "129")])             # one row into the second tile
def test_nt_gemm_over_a_list_of_live_rows(M, N, K, mode, accumulate, kind):
    """gemm_nt8_live with EpiParams::live_unit = 1 (realise_gemm_nt_live_rows): the layer GEMMs of a round-6 training step.  A tile is
    ANY 128 listed rows; the listed rows carry EXACTLY the dense launch's values (same kernel, same accumulation order per row, the
    dropout hash indexed by the original row), read and written at their original positions; the unlisted rows - A rows NaN - are
    neither read nor written."""
    lib = _capi.load()
    g = torch.Generator().manual_seed(M + N + K + mode)
    rng = np.random.default_rng(M + mode)
    if kind == "ragged":            # sentences of 128 rows with a live prefix of 3 .. 128 rows
        live = np.concatenate([np.arange(s * 128, s * 128 + int(rng.integers(3, 129))) for s in range(M // 128)])
    elif kind == "one":
        live = np.array([601])
    elif kind == "none":
        live = np.zeros(0, np.int64)
    elif kind == "all":
        live = np.arange(M)
    elif kind == "129":
        live = np.sort(rng.choice(M, 129, replace=False))
    else:
        live = np.arange(1, M, 2)
    rows = torch.from_numpy(live).long().cuda()
    dead = torch.ones(M, dtype=torch.bool, device="cuda")
    dead[rows] = False
    a = (torch.randn(M, K, generator=g) * 0.1).bfloat16().cuda()
    b = (torch.randn(N, K, generator=g) * 0.1).bfloat16().cuda()
    bias = (torch.randn(N, generator=g) * 0.1).float().cuda()
    aux = torch.randn(M, N, generator=g).bfloat16().cuda() if mode in (2, 4) else None
    old = torch.randn(M, N, generator=g).bfloat16().cuda()
    drop = 0.1 if mode == 2 else 0.0

    def run(live_form):
        out = old.clone()
        out2 = old.clone() if mode == 1 else None
        x = a.clone()
        if live_form:
            x[dead] = float("nan")
        ep = _epilogue(mode, out, N, accumulate, out2, bias if mode != 4 else None, aux, drop)
        if live_form:
            lst = torch.full((M + 128,), -7, dtype=torch.int32, device="cuda")
            lst[:len(live)] = torch.from_numpy(live.astype(np.int32)).cuda()
            cnt = torch.tensor([len(live)], dtype=torch.int32, device="cuda")
            _capi.check(lib.realise_gemm_nt_live_rows(stream(), P(x), K, P(b), K, M, N, K, C.byref(ep), P(lst), P(cnt)), "gemm_nt_live_rows")
        else:
            _capi.check(lib.realise_gemm_nt(stream(), _capi.BF16, P(x), K, P(b), K, M, N, K, C.byref(ep)), "gemm_nt")
        torch.cuda.synchronize()
        return out, out2

    out_l, out2_l = run(True)
    out_d, out2_d = run(False)
    assert torch.isfinite(out_l.float()).all()
    assert torch.equal(out_l[rows], out_d[rows])
    assert torch.equal(out_l[dead], old[dead])
    if mode == 1:
        assert torch.equal(out2_l[rows], out2_d[rows]) and torch.equal(out2_l[dead], old[dead])
    out_l2, _ = run(True)
    assert torch.equal(out_l, out_l2)


@pytest.mark.parametrize("B,S,layers", [(8, 64, 2), (64, 128, 1), (8, 40, 1), (8, 256, 1)])
@pytest.mark.parametrize("form", [2, 1])
def test_live_row_step_forms_equal_the_dense_step(B, S, layers, form):
    """realise_set_engine(10, 2) (round-6 default: the layer GEMMs walk the list of live ROWS) and (10, 1) (rounds 4 / 5: live 16-row
    blocks) against the dense step (10, 0), two consecutive steps on one module: the same loss, bit-identical logits on every live row,
    every order-fixed gradient bit-identical.  With the row list the rows that merely complete a sentence's last 16-row block are no
    longer produced by the GEMMs; the one GEMM whose output feeds a block-wise reduction as dY (the GELU' data gradient) keeps the block
    list, so those rows of it are recomputed as the exact zeros they are."""
    lib = _capi.load()
    cfg = RealiseConfig(num_hidden_layers=layers, pho_layers=1, out_layers=1)
    sd = init_state_dict_numpy(cfg, seed=11)
    batch = cuda_batch(B, S, 17)
    masks = batch["masks"].bool() | batch["loss_masks"].bool()
    last = torch.where(masks.any(1), S - masks.flip(1).float().argmax(1), torch.zeros(B, dtype=torch.long, device="cuda"))
    live = (torch.arange(S, device="cuda")[None, :] < last[:, None])

    def step(on):
        lib.realise_set_engine(10, on)
        try:
            m = build(cfg, sd, "bf16", train=True)
            out = []
            for _ in range(2):
                m.zero_grad()
                loss, logits = m(batch)
                loss.backward()
                torch.cuda.synchronize()
                out.append((float(loss.item()), logits.detach().clone(), grads_of(m)))
            return out
        finally:
            lib.realise_set_engine(10, 2)

    dense, lived = step(0), step(form)
    for (l0, z0, g0), (l1, z1, g1) in zip(dense, lived):
        assert l0 == l1
        assert torch.isfinite(z1.float()).all()
        assert torch.equal(z0[live], z1[live])
        moved = [n for n in g0 if not torch.equal(g0[n], g1[n])]
        for n in g0:
            scale = g0[n].abs().max().item()
            assert (g0[n] - g1[n]).abs().max().item() <= 5e-5 * scale + 1e-12, n
        layer_weights = [n for n in g0 if ".layer." in n and n.endswith("weight") and "LayerNorm" not in n]
        assert layer_weights and not [n for n in layer_weights if n in moved], [n for n in layer_weights if n in moved][:8]


def test_row_list_trajectory_with_changing_lengths():
    """Ten optimizer steps over batches whose sentence lengths change every step (rows switch between live and padding, so the row-list
    GEMMs keep meeting stale rows inside live 16-row blocks): losses finite and on the dense trajectory."""
    lib = _capi.load()
    cfg = RealiseConfig(num_hidden_layers=2, pho_layers=1, out_layers=1)
    sd = init_state_dict_numpy(cfg, seed=29)
    batches = [cuda_batch(8, 64, 100 + k) for k in range(10)]

    def run(on):
        lib.realise_set_engine(10, on)
        try:
            m = build(cfg, sd, "bf16", train=True)
            opt = FusedAdamW(m, [{"params": [p for p in m.parameters() if p.requires_grad], "weight_decay": 0.01}], lr=2e-4, eps=1e-8, max_grad_norm=1.0)
            losses = []
            for b in batches:
                m.zero_grad()
                loss, logits = m(b)
                loss.backward()
                opt.step()
                assert torch.isfinite(logits.float()).all()
                losses.append(float(loss.item()))
            return losses
        finally:
            lib.realise_set_engine(10, 2)

    d, r = run(0), run(2)
    assert all(np.isfinite(r))
    assert max(abs(a - b) for a, b in zip(d, r)) < 5e-3 * max(d), (d, r)


def test_glyph_lookup_fused_into_block1_loaders_matches_the_gathered_form():
    """K7 (realise_set_engine(13, 1), default): block 1's two forward convolutions gather their taps from the NHWC glyph table through the
    list of distinct ids; (13, 0) gathers the images first and runs the dense loaders.  Same fetches, same order: bit-identical
    activations, loss and gradients, training and evaluation."""
    lib = _capi.load()
    cfg = RealiseConfig(num_hidden_layers=1, pho_layers=1, out_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = init_state_dict_numpy(cfg, seed=23)
    batch = cuda_batch(8, 64, 55)

    def run(on, dtype):
        lib.realise_set_engine(13, on)
        try:
            m = build(cfg, sd, dtype, train=True)
            m.zero_grad()
            loss, _ = m(batch)
            loss.backward()
            torch.cuda.synchronize()
            b1 = m.tap("resnet.block1").clone()
            g = grads_of(m)
            m.eval()
            with torch.no_grad():
                le, ze = m(batch)
            return float(loss.item()), b1, g, float(le.item()), ze.clone()
        finally:
            lib.realise_set_engine(13, 1)

    for dtype in ("bf16", "fp32"):
        (l0, a0, g0, e0, z0), (l1, a1, g1, e1, z1) = run(0, dtype), run(1, dtype)
        assert l0 == l1 and e0 == e1 and torch.equal(a0, a1) and torch.equal(z0, z1)
        # gradients: bit-identical wherever the gathered form is itself bit-stable from run to run at this size (the small-shape
        # column reductions of the BatchNorm backward add with fp32 atomics - DESIGN 3, "reproducibility exceptions"); a tensor that
        # moves between two runs of the SAME form is held to its L2 distance instead
        g0b = run(0, dtype)[2]
        for n in g0:
            if n.startswith("resnet."):
                if torch.equal(g0[n], g0b[n]):
                    assert torch.equal(g0[n], g1[n]), n
                else:
                    ref = (g0[n].float() - g0b[n].float()).norm().item()
                    d = (g0[n].float() - g1[n].float()).norm().item()
                    assert d <= 4.0 * ref + 1e-6 * g0[n].float().norm().item(), (n, d, ref)


# ---------------------------------------------------------------------------------------------- K13: no training logits
def test_training_forward_without_logits_matches_the_two_buffer_form():
    """K13 (`model.train_logits = False`, what trainer.train() and bench.py set: run.py:191 reads outputs[0] alone): the classifier runs
    over the rows that enter the loss, writes their logits rows into the gradient buffer, and the cross-entropy kernel turns every row
    into its gradient in place.  Same accumulators, same bf16 rounding, same row kernel as the form that writes [B, S, V] logits first:
    the loss and every gradient are bit-identical, dropout on, over three batches of different lengths; the tuple's second entry is None.
    The persistent classifier kernel takes the device-side row count here (gemm_nt8p m_dev): a full-vocabulary head is part of it."""
    cfg = RealiseConfig(**SMALL)
    sd = init_state_dict_numpy(cfg, seed=41)
    batches = [cuda_batch(16, 128, 300 + k) for k in range(3)]

    def run(train_logits):
        m = build(cfg, sd, "bf16", train=True)
        m.train_logits = train_logits
        out = []
        for b in batches:
            m.zero_grad()
            loss, logits = m(b)
            assert (logits is None) == (not train_logits)
            loss.backward()
            torch.cuda.synchronize()
            out.append((float(loss.item()), grads_of(m)))
        return out

    a, a2, b = run(True), run(True), run(False)
    for (la, ga), (_, ga2), (lb, gb) in zip(a, a2, b):
        assert np.isfinite(la) and la == lb
        assert set(ga) == set(gb)
        for n in ga:
            # (the tied embedding / classifier gradient takes the embedding scatter's fp32 atomics: not bit-stable between two runs of
            # the SAME form - such a tensor is held to the distance between those two runs instead)
            # (embed_bwd and the gate network's weight / bias reductions add with fp32 atomics: equal only by chance)
            atomics = "embeddings" in n or n == "classifier.weight" or n.startswith("gate_net")
            if not atomics and torch.equal(ga[n], ga2[n]):
                assert torch.equal(ga[n], gb[n]), n
            else:
                ref = (ga[n].float() - ga2[n].float()).norm().item()
                d = (ga[n].float() - gb[n].float()).norm().item()
                assert d <= 4.0 * ref + 1e-5 * ga[n].float().norm().item(), (n, d, ref)


def test_training_forward_without_logits_keeps_the_reference_tuple_where_it_cannot_apply():
    """fp32 parity mode and evaluation keep the reference's (loss, logits) tuple whatever `train_logits` says."""
    cfg = RealiseConfig(**SMALL)
    sd = init_state_dict_numpy(cfg, seed=42)
    b = cuda_batch(4, 64, 310)
    m = build(cfg, sd, "fp32", train=True)
    m.train_logits = False
    loss, logits = m(b)
    assert logits is not None and logits.shape == (4, 64, cfg.vocab_size)
    m2 = build(cfg, sd, "bf16", train=False)
    m2.train_logits = False
    with torch.no_grad():
        loss, logits = m2(b)
    assert logits is not None and logits.dtype == torch.float32


# ---------------------------------------------------------------------------------------------- fp32 logits from the classifier's epilogue
@pytest.mark.parametrize("B,S", [(16, 128), (4, 64)])
def test_eval_fp32_logits_come_from_the_classifier_epilogue_bit_for_bit(B, S):
    """The reference returns fp32 logits (src/models.py:859).  A bf16 engine used to cast its bf16 logits in a pass of its own; now the
    persistent classifier kernel stores the fp32 copy itself (realise_batch.logits_f32_out; B*S = 2048 rows takes that kernel, 256 rows the
    cast fallback inside the engine).  Either way the fp32 tensor holds exactly the bf16 logits widened, and the loss is the same."""
    cfg = RealiseConfig(**SMALL)
    sd = init_state_dict_numpy(cfg, seed=51)
    b = cuda_batch(B, S, 400 + B)
    m = build(cfg, sd, "bf16", train=False)
    with torch.no_grad():
        loss_w, wide = m(b)
    m16 = build(cfg, sd, "bf16", train=False, logits_dtype="bf16")
    with torch.no_grad():
        loss_n, narrow = m16(b)
    torch.cuda.synchronize()
    assert wide.dtype == torch.float32 and narrow.dtype == torch.bfloat16
    assert float(loss_w.item()) == float(loss_n.item())
    assert torch.equal(wide, narrow.float())


# ---------------------------------------------------------------------------------------------- K9 (evaluation): BatchNorm in the conv epilogues
@pytest.mark.parametrize("dtype,B,S", [("bf16", 8, 64), ("fp32", 4, 64), ("bf16", 64, 128)])
def test_eval_batchnorm_folded_into_the_conv_epilogues_matches_the_separate_kernels(dtype, B, S):
    """realise_set_engine(14, 1), default: an evaluation forward applies every BatchNorm of the glyph ResNet (running statistics: a
    per-channel affine map, src/char_cnn.py:15-32) in the epilogue of the convolution in front of it.  Against the separate scale / shift +
    apply kernels (14, 0): fp32 agrees to rounding order (the affine map runs on the fp32 accumulator either way), bf16 differs by where
    the bf16 rounding falls (the raw convolution output is no longer rounded before the map) - block outputs inside 2 bf16 ulps of their
    scale, the same arg-max ids wherever the margin is not a rounding step, the same loss to 1e-3."""
    lib = _capi.load()
    cfg = RealiseConfig(**SMALL)
    sd = init_state_dict_numpy(cfg, seed=61)
    b = cuda_batch(B, S, 500 + B)

    def run(on):
        lib.realise_set_engine(14, on)
        try:
            m = build(cfg, sd, dtype, train=False)
            with torch.no_grad():
                loss, logits = m(b)
            torch.cuda.synchronize()
            return float(loss.item()), logits.float().clone(), [m.tap("resnet.block%d" % k).float().clone() for k in range(1, 6)], m.tap("res_h").float().clone()
        finally:
            lib.realise_set_engine(14, 1)

    l0, z0, t0, r0 = run(0)
    l1, z1, t1, r1 = run(1)
    tol = 1e-5 if dtype == "fp32" else 2.0 ** -6
    for k, (a, c) in enumerate(zip(t0, t1)):
        assert torch.isfinite(c).all()
        assert (a - c).abs().max().item() <= tol * max(1.0, a.abs().max().item()), ("block", k + 1)
    # (res_h is the LayerNorm of block 5's output: values up to ~4, where a bf16 step is 0.03 - a few steps)
    assert (r0 - r1).abs().max().item() <= (1e-4 if dtype == "fp32" else 0.15)
    assert abs(l0 - l1) <= (1e-5 if dtype == "fp32" else 2e-3) * max(1.0, abs(l0))
    assert (z0 - z1).abs().max().item() <= (1e-3 if dtype == "fp32" else 8e-2)


# ---------------------------------------------------------------------------------------------- pipelined optimizer sweep
@pytest.mark.parametrize("layers,model_type", [(5, "arch3"), (2, "arch3"), (4, "spellbert")])
def test_pipelined_optimizer_sweep_gives_the_plain_sweeps_bits(layers, model_type):
    """`model.pipeline_optimizer = True` (trainer.train(), bench.py): FusedAdamW's engine sweep runs on the engine's side stream in the
    order the next forward consumes the parameters, the forward waiting piece by piece (realise_engine_adamw_pipelined).  Same kernels,
    same per-element arithmetic: after six steps over batches of changing shape - an evaluation forward, a state_dict() and an optimizer
    state_dict() in between - the parameters after the first step are those of the plain sweep (bit for bit up to the atomics noise two
    plain runs show), the later losses and parameters stay inside the drift that noise allows."""
    from realise_amd.modeling import SpellBert
    cfg = RealiseConfig(num_hidden_layers=layers, pho_layers=1, out_layers=1)
    sd = init_state_dict_numpy(cfg, seed=71)
    batches = [cuda_batch(8, 64, 600 + k) for k in range(4)] + [cuda_batch(4, 128, 610), cuda_batch(8, 64, 611)]

    def run(pipe):
        if model_type == "arch3":
            m = build(cfg, sd, "bf16", train=True)
        else:
            m = SpellBert(cfg, compute_dtype="bf16")
            m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items() if k in m.state_dict()}, strict=False)
            m.to("cuda"); m.train()
        m.trust_fused_optimizer = True
        m.pipeline_optimizer = pipe
        no_decay = ["bias", "LayerNorm.weight"]
        groups = [{"params": [p for n, p in m.named_parameters() if p.requires_grad and not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
                  {"params": [p for n, p in m.named_parameters() if p.requires_grad and any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
        opt = FusedAdamW(m, groups, lr=3e-4, eps=1e-8, max_grad_norm=1.0)
        losses, mid, first = [], None, None
        for k, b in enumerate(batches):
            m.train()
            m.zero_grad()
            loss = m(b)[0]
            loss.backward()
            opt.step()
            losses.append(loss)
            if k == 0:
                first = {n: v.detach().clone() for n, v in m.state_dict().items()}
            if k == 1:                       # an evaluation forward right behind a step
                m.eval()
                with torch.no_grad():
                    losses.append(m(batches[0])[0])
            if k == 2:                       # a checkpoint right behind a step
                mid = ({n: v.detach().clone() for n, v in m.state_dict().items()}, opt.state_dict()["realise_flat"]["m"].clone())
        torch.cuda.synchronize()
        fin = {n: v.detach().clone() for n, v in m.state_dict().items()}
        return [float(x.item()) for x in losses], mid, fin, opt.state_dict()["realise_flat"], first

    l0, mid0, fin0, o0, f0 = run(False)
    l1, mid1, fin1, o1, f1 = run(True)

    def same(a, b, what, tol):
        if torch.equal(a, b):
            return
        a, b = a.float(), b.float()
        assert (a - b).abs().max().item() <= tol * (1.0 + a.abs().max().item()), what

    # After ONE step: bit for bit, or inside the noise of the backward's fp32 atomics (embedding scatter, gate reductions: DESIGN 3; at
    # this size two runs of the PLAIN sweep differ the same way - tools/opt_pipe_check.py: 7e-9 on a weight, the clip coefficient carries
    # it into every tensor).  A sweep that skipped a tile or a forward that read a weight too early would be off by the learning rate.
    for n in f0:
        same(f0[n], f1[n], ("first step", n), 2e-7)
    # Later: bf16 forwards turn that noise into single rounding steps of activations, losses agree to ~1e-4, and AdamW moves an element
    # whose gradient is all noise by up to lr per step in either direction - bounded drift, not bits
    assert all(abs(a - b) <= 1e-3 * max(1.0, abs(a)) for a, b in zip(l0, l1)), (l0, l1)
    for n in fin0:
        if "running_" in n or "num_batches" in n:      # (BatchNorm running statistics follow the activations, not the optimizer: block 5
            continue                                   #  averages 8 x 64 samples per channel and moves with every rounding step upstream)
        same(fin0[n], fin1[n], n, 6 * 3e-4 * 1.5)
        same(mid0[0][n], mid1[0][n], ("mid", n), 3 * 3e-4 * 1.5)
    # (no bound on the mean drift: a tensor whose true gradient is zero - the attention key bias, which softmax cancels - is moved by AdamW
    # on rounding noise alone, +-lr per step, in both runs)
    assert o0["step"] == o1["step"]


# ---------------------------------------------------------------------------------------------- live-row evaluation forward
@pytest.mark.parametrize("B,S,with_labels", [(8, 64, True), (16, 128, True), (8, 64, False)])
def test_live_row_evaluation_forward_matches_the_dense_one_on_every_real_token(B, S, with_labels):
    """`model.eval_live_rows = True` (opt-in): an evaluation forward computes the transformer stacks over the rows up to every sentence's
    last attended / loss position only, as bf16 training steps do.  Same kernels per row: the logits of every such row and the loss are
    bit-identical to the dense forward's; the padding rows behind that position are finite (run.py:262-270 never reads them)."""
    cfg = RealiseConfig(**SMALL)
    sd = init_state_dict_numpy(cfg, seed=81)
    b = cuda_batch(B, S, 700 + B)
    if not with_labels:
        b = {k: v for k, v in b.items() if k not in ("tgt_idx", "loss_masks")}
    m = build(cfg, sd, "bf16", train=False)

    def run(live):
        m.eval_live_rows = live
        with torch.no_grad():
            out = m(b)
        torch.cuda.synchronize()
        return (float(out[0].item()) if with_labels else None), out[-1].clone()

    l0, z0 = run(False)
    l1, z1 = run(True)
    l2, z2 = run(False)                      # (and the dense form again afterwards: stale rows in the workspace change nothing)
    mk = b["masks"] == 1
    if with_labels:
        mk = mk | (b["loss_masks"] == 1)
    pos = torch.arange(1, S + 1, device="cuda")[None, :]
    last = (mk * pos).max(dim=1).values
    real = (torch.arange(S, device="cuda")[None, :] < last[:, None])
    assert real.float().mean().item() < 0.95, "the batch needs padding for this test to mean anything"
    assert l0 == l1 == l2
    assert torch.equal(z0[real], z1[real]) and torch.equal(z0, z2)
    assert torch.isfinite(z1).all()
