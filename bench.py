#!/usr/bin/env python
"""Headline benchmark: train sentences/sec of the full ReaLiSe model (SpellBertPho2ResArch3, BASELINE.json configs[1]) at
seq_len 128, batch 64 per GPU, bf16 MFMA compute, synthetic SIGHAN-shaped data, random-init weights.

A step = forward + backward (+ gradient all-reduce when N > 1) + global-norm clip + AdamW + LR schedule, i.e. exactly the body
of the reference's hot loop (src/run.py:186-211) minus the host-side logging.  Inputs are resident in HBM before the timed
region starts.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --workload glyph256          # BASELINE configs[3]: the glyph ResNet alone on 256 x 128 glyph stacks

Rank 0 prints ONE JSON line.
* `value`            whole-job train sentences/s (max over ranks of the timed region).
* `roofline`         the dominant kernel family - the bf16 MFMA NT GEMMs (every nn.Linear forward / data gradient, classifier,
                     GRU step) - timed per launch with HIP events recorded on the launch stream during the timed region;
                     `traffic` = HBM bytes per launch from the committed PMC passes of this same command, null when the
                     kernels changed since those passes (profiles/round2_pmc_traffic.json carries a hash of csrc/).
* `forward`          forward-only timings (eval mode and train mode) with their MFMA utilisation - the north-star's own target
                     metric (>= 40 % bf16 MFMA utilisation on the fused forward).
* FLOP accounting    `nominal` = dense FLOPs of the reference's graph; `executed` = what the engine
                     really multiplies: the glyph ResNet runs once per DISTINCT token id of the batch (count-weighted BatchNorm
                     keeps the statistics exact), so its FLOPs scale with U / (B*S), read back from the device after the run.
* `cpu_baseline`     the CPU oracle (a port: oracle/realise_ref.py) on a bounded sample of the same workload on this box's host
                     cores (rank 0, N = 1 only): 1 warm-up + 3 timed iterations of eval-forward and of fwd+bwd+clip+AdamW.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
FAMILIES = ["gemm_nt", "conv_nt", "gemm_tn", "conv_tn", "attn_fwd", "attn_bwd"]
RESNET_FLOPS_PER_STACK = 126418944      # CharResNet forward on one 3 x 32 x 32 glyph stack
# HIP-event pairs around every launch of the MFMA kernel families cost ~4 us of stream time each (~8 % of a step when every
# step is bracketed): the timed region brackets every 10th step, which leaves the averages intact and the cost below 1 %.
PROFILE_EVERY = 10
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "round3_pmc_traffic.json")


def fwd_flops_per_sentence(S, mean_len, resnet_frac=1.0):
    """BASELINE.md section 4 accounting (dense forward FLOPs); resnet_frac scales the glyph-ResNet term (dedup)."""
    per_tok = 19 * (14155776 + 2 * 2 * S * 768) + 32452608 + RESNET_FLOPS_PER_STACK * resnet_frac + 18432 + 7077888 * mean_len
    return per_tok * S


def kernels_sha():
    """hash of the kernel sources: PMC traffic measured on other kernels is reported as stale (null)"""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "realise_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic_per_launch(prefixes):
    """HBM bytes per launch of a kernel family from the committed PMC passes (separate rocprofv3 --pmc runs of this same
    command, FETCH_SIZE x2 and KiB units as MI355X_MICROARCH.md prescribes; tools/gpu_pmc_bench.sh + tools/pmc_summary.py).
    Counters cannot be read inside a timed run: null when the summary is absent or was taken on different kernels."""
    try:
        with open(PMC_TRAFFIC) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, "no PMC summary committed"
    if t.get("kernels_sha") != kernels_sha():
        return None, "stale: PMC passes were taken on kernels %s, this tree is %s" % (t.get("kernels_sha"), kernels_sha())
    n = b = 0.0
    for name, v in t.get("kernels", {}).items():
        if any(name.startswith(p) for p in prefixes):
            n += v["launches"]
            b += v["launches"] * (v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"])
    return (round(b / n) if n else None), "rocprofv3 --pmc FETCH_SIZE(x2)/WRITE_SIZE, kernels %s" % t.get("kernels_sha")


def cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = logical
    return model, physical, logical


def cpu_baseline(sd_cpu, cfg, sample_b, S):
    """the oracle (CPU restatement, fp32) on `sample_b` sentences of the same synthetic workload: 1 warm-up + 3 timed iterations
    of (a) the eval forward and (b) forward + backward + clip_grad_norm_ + AdamW (dropout on), all physical cores"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import realise_ref as R
    from realise_amd.data import synthetic_batch
    model, physical, logical = cpu_info()
    threads = max(1, physical)                  # all physical cores
    torch.set_num_threads(threads)
    sd = {}
    for k, v in sd_cpu.items():
        t = v.clone()
        if t.dtype == torch.float32 and k != "char_images_multifonts" and "running_" not in k:
            t.requires_grad_(True)
        sd[k] = t
    sd["classifier.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    ocfg = dict(cfg)
    ocfg["hidden_dropout_prob"] = 0.1
    batch = synthetic_batch(sample_b, S, seed=98)
    params = [(k, p) for k, p in sd.items() if p.requires_grad and k != "classifier.weight"]
    mstate = {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in params}

    def eval_fwd():
        with torch.no_grad():
            R.arch3_forward(sd, ocfg, batch, training=False)

    def train_step(step):
        for _, p in params:
            p.grad = None
        R.arch3_forward(sd, ocfg, batch, training=True)[0].backward()
        live = [(k, p) for k, p in params if p.grad is not None]
        grads, _ = R.clip_grad_norm([p.grad for _, p in live], 1.0)
        with torch.no_grad():
            for (k, p), g in zip(live, grads):
                m, v = mstate[k]
                pn, mn, vn = R.adamw_step(p, g, m, v, step, lr=5e-5, eps=1e-8)
                p.copy_(pn)
                mstate[k] = (mn, vn)

    def timed(fn, n):
        ts = []
        for i in range(n + 1):
            t0 = time.perf_counter()
            fn(i + 1) if fn is train_step else fn()
            ts.append(time.perf_counter() - t0)
        return ts[0], ts[1:]

    warm_e, te = timed(eval_fwd, 3)
    # keep the default run bounded on slow hosts: one train iteration is ~5x an eval forward
    n_train = 3 if 5.0 * warm_e * 4 < 90.0 else 1
    warm_t, tt = timed(train_step, n_train)
    tr = sum(tt) / len(tt)
    ev = sum(te) / len(te)
    return {"value": round(sample_b / tr, 4), "unit": "sentences/s", "cores": threads, "kind": "port",
            "cpu_model": model, "physical_cores": physical, "logical_cpus": logical,
            "eval_forward_sentences_per_s": round(sample_b / ev, 4),
            "sample": "oracle/realise_ref.py fp32, %d sentences x seq_len %d, %d torch threads: train step (forward + backward + "
                      "clip_grad_norm + AdamW, dropout on) 1 warm-up (%.1f s) + %d timed (mean %.2f s); eval forward 1 warm-up + 3 timed "
                      "(mean %.2f s)" % (sample_b, S, threads, warm_t, len(tt), tr, ev)}


def read_families(lib, sampled):
    fams = {}
    cnt, ms, work = C.c_longlong(), C.c_double(), C.c_double()
    for i, name in enumerate(FAMILIES):
        lib.realise_profile_read(i, C.byref(cnt), C.byref(ms), C.byref(work))
        if cnt.value:
            fams[name] = {"launches_per_step": cnt.value / sampled, "ms_per_step": ms.value / sampled,
                          "avg_launch_us": 1e3 * ms.value / cnt.value,
                          "tflops": work.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0}
    return fams


def dump_launches(lib, sampled, path):
    out = {}
    ms = (C.c_float * 8192)()
    work = (C.c_double * 8192)()
    for i, name in enumerate(FAMILIES):
        n = lib.realise_profile_dump(i, 8192, ms, work)
        per = n // max(1, sampled)
        out[name] = [{"us": round(ms[k] * 1e3, 2), "gflop": round(work[k] * 1e-9, 3),
                      "tflops": round(work[k] / (ms[k] * 1e-3) * 1e-12, 1) if ms[k] > 0 else 0.0} for k in range(per)]
    with open(path, "w") as f:
        json.dump(out, f, indent=0)


def timed_loop(fn, steps, world, dev):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--workload", default="train", choices=["train", "glyph256"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: CPU-staged collectives (N ranks may share one GPU)")
    ap.add_argument("--grad-dtype", default="fp32", choices=["fp32", "bf16"], help="dtype of the gradient buckets on the wire (N > 1)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="N = 1: still initialise the process group and wrap the model in the data-parallel wrapper (RCCL initialisation, the "
                         "communication stream and the per-bucket events are exercised with a one-rank all-reduce)")
    ap.add_argument("--dist-timeout", type=int, default=300, help="seconds before a rendezvous / collective gives up")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--profile-markers", action="store_true",
                    help="time launches with event markers recorded around them (round 1/2 form) instead of the dispatch's own timestamps")
    ap.add_argument("--no-glyph256", action="store_true", help="skip the BASELINE configs[3] sub-measurement of the default run")
    ap.add_argument("--dump-launches", default=None, help="write the per-launch durations of the first sampled step to this JSON file")
    ap.add_argument("--no-forward", action="store_true", help="skip the forward-only measurements")
    ap.add_argument("--no-overlap", action="store_true", help="run the bert / pho / glyph branches serially on one stream")
    ap.add_argument("--knob", action="append", default=[], help="diagnostic knob as name:key=value, e.g. engine:0=1 (realise_set_engine(0, 1)); names: engine, ln, nt8p")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU")
    if args.backend == "nccl" and world > ndev:
        raise SystemExit("%d ranks but %d GPUs: RCCL needs one GPU per rank (use --backend gloo to share a GPU)" % (world, ndev))
    local_dev = local % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    ddp = world > 1 or args.force_ddp
    if ddp:
        import datetime
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        for k in ("MASTER_ADDR", "MASTER_PORT"):
            if k not in os.environ:
                raise SystemExit("%s is not set: launch through torch.distributed.run (--master-addr 127.0.0.1 --master-port P)" % k)
        try:
            if args.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=args.dist_timeout))
            else:
                dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=args.dist_timeout))
        except Exception as e:      # noqa: BLE001 - whatever the backend raises, say which rank / rendezvous it was
            raise SystemExit("rank %d/%d: %s process group did not come up within %d s at %s:%s (%s: %s)"
                             % (rank, world, args.backend, args.dist_timeout, os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"),
                                type(e).__name__, e))

    from realise_amd import _capi
    from realise_amd.config import RealiseConfig
    from realise_amd.data import synthetic_batch
    from realise_amd.ddp import DistributedDataParallel
    from realise_amd.modeling import SpellBertPho2ResArch3
    from realise_amd.optim import FusedAdamW, get_linear_schedule_with_warmup
    lib = _capi.load()
    lib.realise_profile_mode(0 if args.profile_markers else 1)
    for kn in args.knob:
        name, kv = kn.split(":")
        k, v = kv.split("=")
        getattr(lib, "realise_set_" + name)(int(k), int(v))

    cfg = RealiseConfig()                                   # full model: 12 + 4 + 3 layers, 3 fonts, dropout 0.1
    model = SpellBertPho2ResArch3(cfg, compute_dtype=args.dtype, seed=0)
    if args.workload == "glyph256":
        out = glyph_workload(args, model, dev, lib, world, rank)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if ddp:
            dist.destroy_process_group()
        return
    B = args.batch or 64
    sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_sample > 0:
        sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(dev)
    model.train()
    model.assume_unit_loss_grad = True                      # plain loss.backward(), as in run.py:200
    wrapped = DistributedDataParallel(model, grad_dtype=args.grad_dtype) if ddp else model
    no_decay = ["bias", "LayerNorm.weight"]                 # run.py:146-151
    groups = [{"params": [p for n, p in model.named_parameters() if p.requires_grad and not any(nd in n for nd in no_decay)],
               "weight_decay": 0.0},
              {"params": [p for n, p in model.named_parameters() if p.requires_grad and any(nd in n for nd in no_decay)],
               "weight_decay": 0.0}]
    opt = FusedAdamW(model, groups, lr=5e-5, eps=1e-8, max_grad_norm=1.0)         # train.sh / run.py:333-339
    sched = get_linear_schedule_with_warmup(opt, 10000, 1000000)

    batch = synthetic_batch(B, args.seq, seed=1000 + rank)
    mean_len = float(sum(batch["pho_lens"])) / len(batch["pho_lens"])
    for k in ("src_idx", "tgt_idx", "masks", "loss_masks", "pho_idx"):
        batch[k] = batch[k].to(dev)
    tr_loss = torch.zeros((), device=dev)

    def step(i=0):
        loss = wrapped(batch)[0]
        loss.backward()
        tr_loss.add_(loss.detach())
        opt.step()
        sched.step()
        model.zero_grad()

    for _ in range(args.warmup):
        step()
    profile = (rank == 0) and not args.no_profile
    if profile:
        _capi.check(lib.realise_profile_enable((args.steps // PROFILE_EVERY + 1) * 1200 + 64), "realise_profile_enable")
    sampled = [0]

    overlap = 0 if args.no_overlap else 1
    lib.realise_set_branch_overlap(overlap)
    lib.realise_set_wgrad_overlap(overlap)

    def prof_step(i):
        if profile:                                          # bracket launches on every 10th timed step only
            on = (i % PROFILE_EVERY) == 0
            lib.realise_profile_pause(0 if on else 1)
            # per-launch durations are only meaningful when a kernel has the chip to itself: the sampled steps run the three
            # model branches serially (the other steps overlap them on three streams)
            lib.realise_set_branch_overlap(0 if on else overlap)
            lib.realise_set_wgrad_overlap(0 if on else overlap)
            sampled[0] += on
        step()

    elapsed = timed_loop(prof_step, args.steps, world, dev)
    lib.realise_set_branch_overlap(overlap)
    lib.realise_set_wgrad_overlap(overlap)
    final_loss = float(tr_loss.item()) / max(1, args.steps + args.warmup)
    fams = {}
    if profile:
        fams = read_families(lib, sampled[0])
        if args.dump_launches:
            dump_launches(lib, sampled[0], args.dump_launches)
        lib.realise_profile_disable()

    # live glyph rows: distinct token ids of the batch / tokens (device-side dedup bookkeeping of the last step)
    T_ = B * args.seq
    uniq = int(model.tap("glyph.bounds").view(torch.int32)[0].item())
    live = uniq / float(T_)

    fwd = None
    if not args.no_forward:
        fwd = {}
        fsteps = max(5, args.steps)
        nom = fwd_flops_per_sentence(args.seq, mean_len) * B
        exe = fwd_flops_per_sentence(args.seq, mean_len, live) * B
        for mode in ("eval", "train"):
            model.train(mode == "train")
            model.static_weights = True                      # weights do not change between these forwards
            with torch.no_grad():
                for _ in range(2):
                    model(batch)
                t = timed_loop(lambda i: model(batch), fsteps, world, dev)
            ms = 1e3 * t / fsteps
            fwd[mode] = {"ms": round(ms, 3), "sentences_per_s": round(B * world / (ms * 1e-3), 1),
                         "mfma_util_nominal": round(nom / (ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12), 4),
                         "mfma_util_executed": round(exe / (ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12), 4)}
        fwd["note"] = ("forward only, batch %d x seq %d per GPU, loss + logits computed; eval: BatchNorm running statistics, no dropout; "
                       "train: batch statistics + dropout 0.1, no backward; target (BASELINE.json north_star): >= 0.40 bf16 MFMA "
                       "utilisation" % (B, args.seq))
        model.static_weights = False
        model.train()

    if rank == 0:
        sent = world * B * args.steps
        step_nom = 3.0 * fwd_flops_per_sentence(args.seq, mean_len) * B
        step_exe = 3.0 * fwd_flops_per_sentence(args.seq, mean_len, live) * B
        sec = elapsed / args.steps
        out = {
            "metric": "train sentences/sec (seq_len=128)",
            "value": round(sent / elapsed, 2),
            "unit": "sentences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * sec, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]%s: full ReaLiSe SpellBertPho2ResArch3 (12+4+3 BERT layers, pinyin GRU, "
                                   "3-font glyph ResNet, gate, tied 21128-way classifier), train step = fwd+bwd%s+clip+AdamW, "
                                   "dropout 0.1, random-init weights, SIGHAN-shaped synthetic batch"
                                   % ("" if world == 1 else " x%d GPUs (configs[2])" % world,
                                      "" if not ddp else "+%s all-reduce (%s buckets)" % ("RCCL" if args.backend == "nccl" else "gloo", args.grad_dtype)),
                       "per_gpu_batch": B, "global_batch": B * world, "seq_len": args.seq,
                       "parallelism": "dp%d" % world, "backend": args.backend if ddp else None,
                       "mean_pinyin_len": round(mean_len, 3), "mean_loss": round(final_loss, 4),
                       "branch_overlap": bool(overlap),
                       "distinct_glyphs": uniq, "tokens": T_},
            "model_flops_per_step_per_gpu": {"nominal": step_nom, "executed": step_exe,
                                             "note": "nominal = dense reference graph (3 x forward); executed: the glyph ResNet runs on the "
                                                     "%d distinct token ids of the %d tokens" % (uniq, T_)},
            "model_mfma_util": round(step_nom / sec / (PEAK_BF16_TFLOPS * 1e12), 4),
            "model_mfma_util_executed": round(step_exe / sec / (PEAK_BF16_TFLOPS * 1e12), 4),
        }
        if fwd is not None:
            out["forward"] = fwd
        for name in ("conv_nt", "conv_tn"):                  # launch records charge the dense row count: scale to the live rows
            if name in fams:
                fams[name]["tflops_nominal"] = fams[name]["tflops"]
                fams[name]["tflops"] = fams[name]["tflops"] * live
                fams[name]["rows_live_frac"] = live
        if "gemm_nt" in fams:
            f = fams["gemm_nt"]
            traffic, tnote = pmc_traffic_per_launch(["gemm_nt8_kernel", "gemm_nt8p_kernel", "gemm_nt_kernel<bf16_t, DenseLoader<bf16_t>"])
            out["roofline"] = {"bound": "mfma",
                               "kernel": "dense NT GEMM family: gemm_nt8p_kernel<256x192> (persistent 8-wave ping-pong: qkv, FFN-up, FFN-down "
                                         "dgrad, classifier) + gemm_nt8_kernel<128x192, two per CU> (N = 768 outputs) + "
                                         "gemm_nt_kernel<bf16, DenseLoader> (GRU steps), v_mfma_f32_16x16x32_bf16",
                               "achieved": round(f["tflops"], 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(f["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": tnote,
                               "avg_launch_us": round(f["avg_launch_us"], 2),
                               "launches_per_step": f["launches_per_step"],
                               "flops_per_launch": round(f["tflops"] * 1e12 * f["avg_launch_us"] * 1e-6),
                               "timing": "event markers around each launch" if args.profile_markers else
                                         "HIP events attached to each dispatch (hipExtLaunchKernelGGL start/stop = the kernel's own begin/end timestamps)",
                               "note": "per-launch durations from every %dth timed step; those steps run the three model "
                                       "branches serially so each kernel is timed alone" % PROFILE_EVERY}
            out["kernel_families"] = {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in fams.items()}
    if world == 1 and not ddp and not args.no_glyph256:
        # BASELINE configs[3] next to the headline number (same process, same model object): the glyph ResNet alone on 256 x 128 stacks
        import copy
        gargs = copy.copy(args)
        gargs.batch, gargs.steps, gargs.warmup = 256, max(3, min(args.steps, 8)), 2
        g = glyph_workload(gargs, model, dev, lib, world, rank)
        out["glyph256"] = {"ms_per_step_dense": g["ms_per_step"], "stacks_per_s_dense": g["value"],
                           "forward_ms_dense": g["forward"]["ms"], "model_mfma_util_dense": g["model_mfma_util"],
                           "ms_per_step_dedup": g["dedup"]["ms_per_step"], "stacks_per_s_dedup": g["dedup"]["stacks_per_s"],
                           "distinct_glyphs": g["dedup"]["distinct_glyphs"], "steps": gargs.steps,
                           "kernel_families": g.get("kernel_families"),
                           "workload": g["config"]["workload"]}
    if rank == 0:
        if sd_cpu is not None:
            out["cpu_baseline"] = cpu_baseline(sd_cpu, cfg, args.cpu_sample, args.seq)
        print(json.dumps(out), flush=True)
    if ddp:
        dist.destroy_process_group()


def glyph_workload(args, model, dev, lib, world, rank):
    """BASELINE configs[3]: CharResNet alone on batch 256 x seq 128 = 32768 glyph stacks [3, 32, 32]; a step = forward (train-mode
    BatchNorm) + backward of the 15 convolutions / BatchNorms.  Timed DENSE (one image per token, like the reference, so nominal
    == executed FLOPs); the production dedup path (one image per distinct id) is timed beside it."""
    from realise_amd import _capi
    from realise_amd.data import synthetic_batch
    B = args.batch or 256
    S = args.seq
    model.to(dev)
    model.train()
    batch = synthetic_batch(B, S, seed=2000 + rank, with_pho=False)
    src = batch["src_idx"].to(dev)
    dres = torch.randn((B, S, 768), device=dev, dtype=torch.float32).to(torch.bfloat16 if args.dtype == "bf16" else torch.float32) * 1e-3
    stacks = B * S

    def step(i=0):
        model.glyph_forward(src, training=True)
        model.glyph_backward(dres)
        model.zero_grad()

    res = {}
    fams = {}
    for dedup in (0, 1):
        lib.realise_set_glyph_dedup(dedup)
        for _ in range(args.warmup):
            step()
        profile = (rank == 0) and not args.no_profile and dedup == 0
        if profile:
            _capi.check(lib.realise_profile_enable(args.steps * 200 + 64), "realise_profile_enable")
        t = timed_loop(step, args.steps, world, dev)
        if profile:
            fams = read_families(lib, args.steps)
            lib.realise_profile_disable()
        res[dedup] = t / args.steps
        with torch.no_grad():
            tf = timed_loop(lambda i: model.glyph_forward(src, training=True), args.steps, world, dev) / args.steps
        res["fwd%d" % dedup] = tf
    lib.realise_set_glyph_dedup(1)
    uniq = int(model.tap("glyph.bounds").view(torch.int32)[0].item())
    if rank == 0:
        nom_fwd = float(RESNET_FLOPS_PER_STACK) * stacks
        out = {
            "metric": "glyph stacks/sec (CharResNet forward+backward, 3x32x32 stacks, batch 256 x seq_len 128)",
            "value": round(world * stacks / res[0], 1), "unit": "glyph stacks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * res[0], 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: glyph-CNN stress, CharResNet (5 BasicBlocks, BatchNorm batch statistics) forward + "
                                   "backward on 256 x 128 = 32768 glyph stacks, dense (one image per token)", "batch": B, "seq_len": S,
                       "glyph_stacks": stacks, "distinct_glyphs": uniq},
            "model_flops_per_step_per_gpu": {"nominal": 3.0 * nom_fwd, "executed": 3.0 * nom_fwd},
            "model_mfma_util": round(3.0 * nom_fwd / res[0] / (PEAK_BF16_TFLOPS * 1e12), 4),
            "forward": {"ms": round(1e3 * res["fwd0"], 3), "mfma_util_nominal": round(nom_fwd / res["fwd0"] / (PEAK_BF16_TFLOPS * 1e12), 4)},
            "dedup": {"ms_per_step": round(1e3 * res[1], 3), "forward_ms": round(1e3 * res["fwd1"], 3), "distinct_glyphs": uniq,
                      "stacks_per_s": round(world * stacks / res[1], 1),
                      "note": "production path: the ResNet runs once per distinct token id, BatchNorm weighted by multiplicity (identical results)"},
        }
        if "conv_nt" in fams:
            f = fams["conv_nt"]
            traffic, tnote = pmc_traffic_per_launch(["gemm_nt_kernel<bf16_t, ConvLoader<bf16_t>"])
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt_kernel<bf16, ConvLoader> (implicit-im2col conv forward / data gradient, "
                                                          "v_mfma_f32_16x16x32_bf16)",
                               "achieved": round(f["tflops"], 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(f["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": tnote,
                               "avg_launch_us": round(f["avg_launch_us"], 2), "launches_per_step": f["launches_per_step"],
                               "flops_per_launch": round(f["tflops"] * 1e12 * f["avg_launch_us"] * 1e-6)}
            out["kernel_families"] = {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in fams.items()}
        return out
    return None


if __name__ == "__main__":
    main()
