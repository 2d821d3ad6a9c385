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
    l1, g1 = train_step(m, tb)
    e1 = eval_step(m, eb)
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
    assert eval_step(m, eb)[0] == e0[0] or lib.realise_engine_plan_installs(m._engine) == 5      # (eb may have been the evicted one)
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
    training_args.bin.  Dropout off, so both sides are deterministic and the weights after the run must agree bit for bit."""
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
    assert abs(mean1 - tr_loss / global_step) < 1e-6 * abs(mean1) + 1e-7
    for line, (g_, lr_, l_) in zip(lines, want):
        f = line.replace(",", "").split()
        assert int(f[1]) == g_ and abs(float(f[3]) - lr_) < 1e-12 and abs(float(f[5]) - l_) < 2e-6 * abs(l_) + 1e-6, (line, g_, lr_, l_)
    torch.cuda.synchronize()
    assert torch.equal(m1.flat_parameters(), m2.flat_parameters())
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
            lib.realise_set_engine(10, 1)
            lib.realise_set_engine(5, 1)

    (l0, g0), (l1, g1) = step(0), step(1)
    assert l0 == l1
    for n in g0:
        s = g0[n].abs().max().item()
        assert (g0[n] - g1[n]).abs().max().item() <= 5e-5 * s + 1e-12, n
