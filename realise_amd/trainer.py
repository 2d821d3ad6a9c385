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
          gradient_accumulation_steps=1, max_steps=-1, logging_steps=0, save_steps=0, output_dir=None, training_args=None,
          log_every=0, log_fn=None, seed=17, return_global_step=False):
    """the hot loop of run.py:125-237 with the fused optimizer.  Returns the mean loss per optimizer step (``tr_loss / global_step``,
    run.py:237); with ``return_global_step=True`` the reference's pair ``(global_step, tr_loss / global_step)``.

    Call sites reproduced (same names, same arithmetic):
    * ``gradient_accumulation_steps`` (run.py:193-194, 203): the loss is divided before ``backward()`` (the division reaches the
      engine as the device-side scalar d loss - no pass over the gradients), the optimizer steps every N-th batch, ``t_total``
      counts optimizer steps (run.py:139-143), batches left over at the end of an epoch carry into the next one like the
      reference's (its ``step`` restarts per epoch: so does this one);
    * ``max_steps`` (run.py:139-141, 232-236): overrides the epoch count, the loop stops once ``global_step > max_steps``;
    * ``logging_steps`` (run.py:214-221): every N optimizer steps, ``(tr_loss - logging_loss) / logging_steps`` and the
      scheduler's rate go to ``log_fn`` (default ``print``) as "Step: {}, LR: {}, Loss: {}" - rank 0 only;
      ``log_every`` is the older name of the same knob;
    * ``save_steps`` (run.py:223-230): every N optimizer steps rank 0 writes ``output_dir/saved_ckpt-{global_step}`` with
      ``save_pretrained`` and ``training_args.bin`` (``torch.save(training_args, ...)``: whatever object the caller hands over,
      the reference saves its argparse namespace).
    The per-step ``loss.item()`` of run.py:202 is a device-side accumulation here (no host synchronisation in the loop); the
    windowed log line reads it back once per ``logging_steps``."""
    import os
    build_batch = build_batch or type(model).build_batch
    gas = max(1, int(gradient_accumulation_steps))
    logging_steps = int(logging_steps or log_every or 0)
    log_fn = log_fn or print
    model.to(device)
    rank = 0
    if distributed:                    # run.py:130-137: every rank trains on its own strided shard
        import torch.distributed as dist
        rank = dist.get_rank()
        items = shard(list(items), rank, dist.get_world_size())
    wrapped = DistributedDataParallel(model) if distributed else model
    no_decay = ["bias", "LayerNorm.weight"]
    groups = [{"params": [p for n, p in model.named_parameters() if p.requires_grad and not any(nd in n for nd in no_decay)],
               "weight_decay": weight_decay},
              {"params": [p for n, p in model.named_parameters() if p.requires_grad and any(nd in n for nd in no_decay)],
               "weight_decay": 0.0}]
    opt = FusedAdamW(model, groups, lr=lr, eps=adam_epsilon, max_grad_norm=max_grad_norm)
    per_epoch = len(items) // batch_size // gas                       # optimizer steps per epoch (run.py:139-143, floor)
    if max_steps > 0:
        steps_total = max_steps
        epochs = max_steps // max(1, per_epoch) + 1
    else:
        steps_total = max(1, per_epoch * epochs)
    sched = get_linear_schedule_with_warmup(opt, warmup_steps, steps_total)
    tr_loss = torch.zeros((), device=device)
    logging_loss = 0.0
    global_step = 0
    # this loop owns every parameter write between opt.step() and the next forward: the step's operand copies can be trusted
    # (modeling.py: trust_fused_optimizer; the module default re-derives every copy on every forward).  Restored on ANY exit
    # (an exception inside the loop left the module trusting the optimizer's copies).
    trusted_before = model.trust_fused_optimizer
    model.trust_fused_optimizer = True
    # the loop reads outputs[0] alone (run.py:191): the training forwards skip the [B, S, V] logits (modeling.py: train_logits)
    logits_before = getattr(model, "train_logits", True)
    model.train_logits = False
    # (modeling.py: pipeline_optimizer stays as the caller set it - the pipelined optimizer sweep measured slower than the plain one;
    # whatever it is, the sweep is joined before this function returns)
    pipe_before = getattr(model, "pipeline_optimizer", False)
    try:
        model.zero_grad()
        stop = False
        for ep in range(epochs):
            for step, batch in enumerate(data_helper(items, batch_size, max_seq_length, build_batch, tokenizer, seed=seed + ep)):
                model.train()
                for k, v in batch.items():
                    if torch.is_tensor(v):
                        batch[k] = v.to(device)
                loss = wrapped(batch)[0]
                if gas > 1:
                    loss = loss / gas
                loss.backward()
                tr_loss += loss.detach()
                if (step + 1) % gas == 0:
                    # nn.Embedding raises on an out-of-range id before anything is updated (modeling_bert.py:183-186).  The device-side
                    # check of this step ran at the head of its forward; poll its flag (host-mapped, no synchronisation) before the
                    # weights move, or wait for it when the model runs in strict mode
                    if getattr(model, "strict_ids", False):
                        model.check_ids()
                    elif hasattr(model, "_raise_on_bad_ids"):
                        model._raise_on_bad_ids()
                    opt.step()                 # clip_grad_norm_(max_grad_norm) + AdamW in the same sweep (run.py:207-209)
                    sched.step()
                    model.zero_grad()
                    global_step += 1
                    if rank == 0 and logging_steps > 0 and global_step % logging_steps == 0:
                        now = tr_loss.item()
                        log_fn("Step: {}, LR: {}, Loss: {}".format(global_step, sched.get_last_lr()[0], (now - logging_loss) / logging_steps))
                        logging_loss = now
                    if rank == 0 and save_steps > 0 and global_step % save_steps == 0:
                        if output_dir is None:
                            raise ValueError("save_steps needs output_dir")
                        ckpt = os.path.join(output_dir, "saved_ckpt-{}".format(global_step))
                        os.makedirs(ckpt, exist_ok=True)
                        model.save_pretrained(ckpt)
                        torch.save(training_args, os.path.join(ckpt, "training_args.bin"))
                if max_steps > 0 and global_step > max_steps:
                    stop = True
                    break
            if stop:
                break
    finally:
        model.trust_fused_optimizer = trusted_before
        model.train_logits = logits_before
        model.pipeline_optimizer = pipe_before
        if hasattr(model, "sync_optimizer"):
            model.sync_optimizer()
        model.mark_parameters_updated(frozen=False)       # whatever runs next re-derives every operand copy
    if hasattr(model, "check_ids"):
        model.check_ids()                  # a bad id in the last batches must not go unreported
    mean = tr_loss.item() / max(1, global_step)
    return (global_step, mean) if return_global_step else mean


@torch.no_grad()
def evaluate(model, items, *, batch_size=64, max_seq_length=128, device="cuda", build_batch=None, tokenizer=None,
             vocab_path=None, label_path=None, output_dir=None, prefix="", live_rows=False):
    """run.py:239-280: mean eval loss and the arg-max ids.  The arg-max runs on the device (``model.decode``), so only
    [B, S] ids cross PCIe instead of the [B, S, 21128] logits.  With ``vocab_path`` + ``label_path`` + ``output_dir`` the
    predictions are also written as ``preds.txt`` / ``labels.txt`` and scored like the reference (``results`` dict).
    ``live_rows=True``: the forwards skip the padding rows behind every sentence's last attended position (``model.eval_live_rows``:
    -12 % per forward at B = 64, S = 128) - the loss and the ids of the real tokens, all the scorer reads (it cuts at ``lengths``), are
    unchanged; the ids returned for padding positions are then arbitrary."""
    from .metric import Metric
    build_batch = build_batch or type(model).build_batch
    model.eval()
    losses, preds, batches = [], [], []
    # nothing updates the parameters inside this loop: the operand copies are derived once, not per batch (~8 B / parameter each time)
    was_static = getattr(model, "static_weights", None)
    if was_static is not None:
        model.mark_parameters_updated()
        model.static_weights = True
    was_live = getattr(model, "eval_live_rows", None)
    if was_live is not None:
        model.eval_live_rows = bool(live_rows)
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
        if was_live is not None:
            model.eval_live_rows = was_live
    if hasattr(model, "check_ids"):
        model.check_ids()
    mean_loss = torch.stack(losses).mean().item()
    if label_path is None:
        return mean_loss, torch.cat(preds)
    import os
    results = Metric(vocab_path).metric(batches, os.path.join(output_dir, prefix, "preds.txt"),
                                        os.path.join(output_dir, prefix, "labels.txt"), label_path)
    return mean_loss, torch.cat(preds), results
