"""Thin trainer reproducing the call sites of the reference's ``src/run.py`` around the hot path:
``make_features`` (run.py:68-101), ``data_helper`` (run.py:104-123), the hot loop (run.py:184-211) and
``evaluate`` (run.py:239-280; the label-line writer and the sentence-level scores live in ``metric.py``).  Everything between
``model(batch)`` and ``optimizer.step()`` runs in librealise_hip.so.
"""
import random

import torch

from .ddp import DistributedDataParallel
from .optim import FusedAdamW, get_linear_schedule_with_warmup


def make_features(examples, max_seq_length):
    """run.py:68-101: truncate / pad ``src_idx`` and ``tgt_idx`` to max_seq_length; ``masks`` = 1 on the kept tokens
    ([CLS] chars [SEP]); ``loss_masks`` = 1 on positions 1 .. min(lengths, max_seq_length - 1) (the characters only)."""
    batch = {k: [] for k in ("id", "src", "tgt", "tokens_size", "lengths", "src_idx", "tgt_idx", "masks", "loss_masks")}
    for e in examples:
        for k in ("id", "src", "tgt", "tokens_size", "lengths"):
            batch[k].append(e.get(k))
        for k in ("src_idx", "tgt_idx"):
            seq = list(e[k])[:max_seq_length]
            pad = max_seq_length - len(seq)
            batch[k].append(seq + [0] * pad)
            if k == "src_idx":
                batch["masks"].append([1] * len(seq) + [0] * pad)
        loss_mask = [0] * max_seq_length
        for i in range(1, min(1 + e["lengths"], max_seq_length)):
            loss_mask[i] = 1
        batch["loss_masks"].append(loss_mask)
    for k in ("src_idx", "tgt_idx", "masks", "loss_masks"):
        batch[k] = torch.tensor(batch[k], dtype=torch.long)
    return batch


def data_helper(items, batch_size, max_seq_length, build_batch, tokenizer=None, is_eval=False, seed=None):
    """run.py:104-123: shuffle (train), then yield feature batches."""
    items = list(items)
    if not is_eval:
        (random.Random(seed) if seed is not None else random).shuffle(items)
    for i in range(0, len(items), batch_size):
        yield build_batch(make_features(items[i:i + batch_size], max_seq_length), tokenizer)


def shard(items, rank, world):
    """run.py:130-137: rank r keeps items r, r+W, ... (tail dropped so every rank sees the same count)."""
    n = len(items) // world * world
    return items[rank:n:world]


def train(model, items, *, batch_size=64, max_seq_length=128, epochs=1, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.0,
          warmup_steps=0, max_grad_norm=1.0, device="cuda", distributed=False, build_batch=None, tokenizer=None,
          log_every=0, seed=17):
    """the hot loop of run.py:125-237 with the fused optimizer; returns the mean loss."""
    build_batch = build_batch or type(model).build_batch
    model.to(device)
    if distributed:                    # run.py:130-137: every rank trains on its own strided shard
        import torch.distributed as dist
        items = shard(list(items), dist.get_rank(), dist.get_world_size())
    wrapped = DistributedDataParallel(model) if distributed else model
    no_decay = ["bias", "LayerNorm.weight"]
    groups = [{"params": [p for n, p in model.named_parameters() if p.requires_grad and not any(nd in n for nd in no_decay)],
               "weight_decay": weight_decay},
              {"params": [p for n, p in model.named_parameters() if p.requires_grad and any(nd in n for nd in no_decay)],
               "weight_decay": 0.0}]
    opt = FusedAdamW(model, groups, lr=lr, eps=adam_epsilon, max_grad_norm=max_grad_norm)
    # this loop owns every parameter write between opt.step() and the next forward: the step's operand copies can be trusted
    # (modeling.py: trust_fused_optimizer; the module default re-derives every copy on every forward)
    trusted_before = model.trust_fused_optimizer
    model.trust_fused_optimizer = True
    steps_total = max(1, len(items) // batch_size * epochs)          # t_total of run.py:142-144 (floor)
    sched = get_linear_schedule_with_warmup(opt, warmup_steps, steps_total)
    tr_loss = torch.zeros((), device=device)
    step = 0
    model.zero_grad()
    for ep in range(epochs):
        for batch in data_helper(items, batch_size, max_seq_length, build_batch, tokenizer, seed=seed + ep):
            model.train()
            for k, v in batch.items():
                if torch.is_tensor(v):
                    batch[k] = v.to(device)
            loss = wrapped(batch)[0]
            loss.backward()
            tr_loss += loss.detach()
            # nn.Embedding raises on an out-of-range id before anything is updated (modeling_bert.py:183-186).  The device-side check
            # of this step ran at the head of its forward; poll its flag (host-mapped, no synchronisation) before the weights move,
            # or wait for it when the model runs in strict mode
            if getattr(model, "strict_ids", False):
                model.check_ids()
            elif hasattr(model, "_raise_on_bad_ids"):
                model._raise_on_bad_ids()
            opt.step()
            sched.step()
            model.zero_grad()
            step += 1
            if log_every and step % log_every == 0:
                print("Step: %d, LR: %.3e, Loss: %.5f" % (step, sched.get_last_lr()[0], tr_loss.item() / step))
    model.trust_fused_optimizer = trusted_before
    model.mark_parameters_updated(frozen=False)       # whatever runs next re-derives every operand copy
    if hasattr(model, "check_ids"):
        model.check_ids()                  # a bad id in the last batches must not go unreported
    return tr_loss.item() / max(1, step)


@torch.no_grad()
def evaluate(model, items, *, batch_size=64, max_seq_length=128, device="cuda", build_batch=None, tokenizer=None,
             vocab_path=None, label_path=None, output_dir=None, prefix=""):
    """run.py:239-280: mean eval loss and the arg-max ids.  The arg-max runs on the device (``model.decode``), so only
    [B, S] ids cross PCIe instead of the [B, S, 21128] logits.  With ``vocab_path`` + ``label_path`` + ``output_dir`` the
    predictions are also written as ``preds.txt`` / ``labels.txt`` and scored like the reference (``results`` dict)."""
    from .metric import Metric
    build_batch = build_batch or type(model).build_batch
    model.eval()
    losses, preds, batches = [], [], []
    # nothing updates the parameters inside this loop: the operand copies are derived once, not per batch (~8 B / parameter each time)
    was_static = getattr(model, "static_weights", None)
    if was_static is not None:
        model.mark_parameters_updated()
        model.static_weights = True
    try:
        for batch in data_helper(items, batch_size, max_seq_length, build_batch, tokenizer, is_eval=True):
            with torch.no_grad():
                loss, logits = model(batch)[:2]
            losses.append(loss.detach())
            ids = model.decode(logits).cpu()
            preds.append(ids)
            if label_path is not None:
                batch["src_idx"] = batch["src_idx"].cpu().numpy()
                batch["pred_idx"] = ids.numpy()
                batches.append(batch)
    finally:
        if was_static is not None:
            model.static_weights = was_static
    if hasattr(model, "check_ids"):
        model.check_ids()
    mean_loss = torch.stack(losses).mean().item()
    if label_path is None:
        return mean_loss, torch.cat(preds)
    import os
    results = Metric(vocab_path).metric(batches, os.path.join(output_dir, prefix, "preds.txt"),
                                        os.path.join(output_dir, prefix, "labels.txt"), label_path)
    return mean_loss, torch.cat(preds), results
