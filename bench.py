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
                     kernels changed since those passes (profiles/round5_pmc_traffic.json carries a hash of csrc/).
* `forward`          forward-only timings (eval mode and train mode) with their MFMA utilisation - the north-star's own target
                     metric (>= 40 % bf16 MFMA utilisation on the fused forward).
* FLOP accounting    `nominal` = dense FLOPs of the reference's graph; `executed` = what the engine
                     really multiplies: the glyph ResNet runs once per DISTINCT token id of the batch (count-weighted BatchNorm
                     keeps the statistics exact), so its FLOPs scale with U / (B*S), read back from the device after the run.
* executed work     launches bounded by a device-side row count (the classifier's gradients over the loss rows, the weight-gradient
                     reductions over the live 16-row blocks of the padded batch, the GRU steps over the alive sequences) book their
                     NOMINAL 2.M.N.K; every family prints `tflops` on the EXECUTED work (counters read back after the run) next to
                     `tflops_nominal`; `roofline.achieved` / `frac` are the executed figures.
* `fp32_parity`      train sentences/s of the fp32 parity mode (the mode that meets the north-star tolerance: logits <= 1e-3,
                     arg-max exact where the reference's margin exceeds fp32 noise) at the same batch / seq_len.
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
def _latest_profile(name):
    """the newest round's committed summary of that name (profiles/roundN_<name>)"""
    for r in (6, 5):
        p = os.path.join(ROOT, "profiles", "round%d_%s" % (r, name))
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", "round6_%s" % name)


PMC_TRAFFIC = _latest_profile("pmc_traffic.json")
ROCPROF_STATS = _latest_profile("final_kernel_stats.md")
NT_KERNEL_PREFIXES = ["gemm_nt8_kernel", "gemm_nt8p_kernel", "gemm_nt_kernel<bf16_t, DenseLoader<bf16_t>", "gemm_nt_ln_kernel"]


def fwd_flops_per_sentence(S, mean_len, resnet_frac=1.0, stack_frac=1.0):
    """BASELINE.md section 4 accounting (dense forward FLOPs); resnet_frac scales the glyph-ResNet term (dedup), stack_frac the 19
    transformer layers (live-row training steps: the rows of the live 16-row blocks over all rows)."""
    per_tok = 19 * (14155776 + 2 * 2 * S * 768) * stack_frac + 32452608 + RESNET_FLOPS_PER_STACK * resnet_frac + 18432 + 7077888 * mean_len
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


def rocprof_family_avg(prefixes):
    """launch-weighted average duration (us) and ms/step of a kernel family in the committed rocprofv3 --kernel-trace --stats summary
    of this same command (profiles/round5_final_kernel_stats.md, branches serial); None when that file is absent.  The summary is a
    record of an earlier run of the same kernels: `roofline.avg_launch_us` (measured live, below) must agree with it."""
    try:
        rows = [l.split("|") for l in open(ROCPROF_STATS) if l.startswith("| `")]
    except OSError:
        return None
    calls = tot = per_step = 0.0
    for r in rows:
        name = r[1].strip().strip("`")
        if any(name.startswith(p) for p in prefixes):
            calls += float(r[2]); tot += float(r[4]); per_step += float(r[5])
    if not calls:
        return None
    return {"kernel_avg_us_rocprof": round(1e3 * tot / calls, 2), "ms_per_step_rocprof": round(per_step, 3), "launches_rocprof": int(calls),
            "source": os.path.relpath(ROCPROF_STATS, ROOT)}


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
    n_train = 2 if 5.0 * warm_e * 3 < 60.0 else 1
    warm_t, tt = timed(train_step, n_train)
    tr = sum(tt) / len(tt)
    ev = sum(te) / len(te)
    # All physical cores is the contract figure (`value`), but a 128-core box is oversubscribed by an 8-sentence sample (rounds 1-3:
    # 2.2 -> 1.1 -> 0.5 sentences/s as the thread count went 8 -> 64 -> 128): sweep smaller thread counts on the eval forward and time
    # the train step again at the best one, so the GPU/CPU ratio is not flattered by a badly configured baseline.
    sweep = {threads: round(sample_b / ev, 4)}
    best_t, best_ev = threads, ev
    for t in (8, 16, 32, 64):
        if t >= threads:
            continue
        torch.set_num_threads(t)
        _, tte = timed(eval_fwd, 1)
        sweep[t] = round(sample_b / tte[0], 4)
        if tte[0] < best_ev:
            best_t, best_ev = t, tte[0]
    best = {"threads": best_t, "eval_forward_sentences_per_s": round(sample_b / best_ev, 4), "value": round(sample_b / tr, 4)}
    if best_t != threads:
        torch.set_num_threads(best_t)
        _, ttb = timed(train_step, 1)
        best["value"] = round(sample_b / ttb[0], 4)
    torch.set_num_threads(threads)
    # `value` / `cores` = the FASTEST configuration found (the thread count actually used for it); the all-physical-cores figure - the
    # pessimal one on a 128-core host, which flattered the GPU / CPU ratio by ~6x in round 4's line - is kept beside it
    return {"value": best["value"], "unit": "sentences/s", "cores": best_t, "kind": "port", "sample_batch": sample_b, "sample_seq_len": S,
            "cpu_model": model, "physical_cores": physical, "logical_cpus": logical,
            "eval_forward_sentences_per_s": best["eval_forward_sentences_per_s"],
            "all_physical_cores": {"threads": threads, "value": round(sample_b / tr, 4), "eval_forward_sentences_per_s": round(sample_b / ev, 4)},
            "best_thread_count": best, "eval_forward_thread_sweep_sentences_per_s": sweep,
            "sample": "oracle/realise_ref.py fp32, %d sentences x seq_len %d, %d torch threads: train step (forward + backward + "
                      "clip_grad_norm + AdamW, dropout on) 1 warm-up (%.1f s) + %d timed (mean %.2f s); eval forward 1 warm-up + 3 timed "
                      "(mean %.2f s); thread sweep: 1 warm-up + 1 timed eval forward per count, train step 1 warm-up + 1 timed at the best"
                      % (sample_b, S, threads, warm_t, len(tt), tr, ev)}


def golden_parity(dev):
    """The parity band of both numeric modes, MEASURED here on the reference's own full-size vector (tests/golden/arch3_b64s128_eval.npz:
    B = 64, S = 128, all 19 layers; arg-max ids, top-1 / top-2 margins and one sampled logit per token of the reference's fp32 CPU
    forward, made by oracle/make_golden_full.py importing the reference): one eval forward per mode.  The fixture is data - the oracle
    is not imported here."""
    import numpy as np
    from realise_amd.config import RealiseConfig
    from realise_amd.data import synthetic_batch
    from realise_amd.init import init_state_dict_numpy
    from realise_amd.modeling import SpellBertPho2ResArch3
    path = os.path.join(ROOT, "tests", "golden", "arch3_b64s128_eval.npz")
    if not os.path.exists(path):
        return None
    g = dict(np.load(path))
    B, S, seed, nl = int(g["meta/B"]), int(g["meta/S"]), int(g["meta/seed"]), int(g["meta/n_layers"])
    cfg = RealiseConfig(num_hidden_layers=nl, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, "arch3", seed=seed, scheme="perturbed")
    batch = synthetic_batch(B, S, seed=seed, with_pho=True)
    real = batch["masks"].numpy() == 1
    slot = torch.from_numpy(g["sample_slot"].astype(np.int64)).to(dev)
    out = {"vector": "tests/golden/arch3_b64s128_eval.npz (the reference's fp32 CPU eval forward at B=%d, S=%d, %d+4+3 layers; "
                     "%d tokens, %d of them real)" % (B, S, nl, B * S, int(real.sum())),
           "reference_min_margin": float(g["margin"].min())}
    for dtype in ("fp32", "bf16"):
        m = SpellBertPho2ResArch3(cfg, compute_dtype=dtype)
        m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()})
        m.to(dev).eval()
        with torch.no_grad():
            loss, logits = m(batch)
        flat = logits.reshape(-1, logits.shape[-1]).float()
        tok = torch.arange(flat.shape[0], device=dev)
        mine = flat[tok, slot].cpu().numpy()
        am = m.decode(logits).cpu().numpy().astype(np.int32)
        eq = am == g["argmax"]
        r = {"max_logit_err": float(np.abs(mine - g["sample_logit"]).max()), "loss_abs_err": abs(float(loss.item()) - float(g["loss"])),
             "argmax_equal_frac": float(eq.mean()), "argmax_equal_frac_real_tokens": float(eq[real].mean()),
             "argmax_mismatches": int((~eq).sum())}
        if dtype == "fp32":
            und = g["margin"] <= 1e-4
            r["undecided_tokens"] = int(und.sum())                       # reference top-1 / top-2 margin within fp32 noise of 19 layers
            r["argmax_mismatches_where_margin_gt_1e-4"] = int((~eq & ~und).sum())
            r["tolerance"] = "north_star: logits <= 1e-3, arg-max equal wherever the reference margin > 1e-4"
        else:
            wide = g["margin"] > 8e-2
            r["argmax_mismatches_where_margin_gt_8e-2_real"] = int((~eq & wide & real).sum())
            r["tolerance"] = "band: logits <= 4e-2, arg-max equal on real tokens wherever the reference margin > 8e-2, agreement > 0.92"
        out[dtype] = r
        del m
    torch.cuda.empty_cache()
    return out


def read_families(lib, sampled):
    fams = {}
    cnt, ms, work, wexe = C.c_longlong(), C.c_double(), C.c_double(), C.c_double()
    for i, name in enumerate(FAMILIES):
        lib.realise_profile_read_ex(i, C.byref(cnt), C.byref(ms), C.byref(work), C.byref(wexe))
        if cnt.value:
            sec = ms.value * 1e-3
            fams[name] = {"launches_per_step": cnt.value / sampled, "ms_per_step": ms.value / sampled,
                          "avg_launch_us": 1e3 * ms.value / cnt.value,
                          "tflops": wexe.value / sec / 1e12 if sec > 0 else 0.0,
                          "tflops_nominal": work.value / sec / 1e12 if sec > 0 else 0.0,
                          "gflop_per_step_executed": wexe.value / sampled * 1e-9, "gflop_per_step_nominal": work.value / sampled * 1e-9}
    return fams


def dump_launches(lib, sampled, path):
    out = {}
    ms = (C.c_float * 8192)()
    work = (C.c_double * 8192)()
    wexe = (C.c_double * 8192)()
    for i, name in enumerate(FAMILIES):
        n = lib.realise_profile_dump_ex(i, 8192, ms, work, wexe)
        per = n // max(1, sampled)
        out[name] = [{"us": round(ms[k] * 1e3, 2), "gflop": round(wexe[k] * 1e-9, 3), "gflop_nominal": round(work[k] * 1e-9, 3),
                      "tflops": round(wexe[k] / (ms[k] * 1e-3) * 1e-12, 1) if ms[k] > 0 else 0.0} for k in range(per)]
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
    ap.add_argument("--no-fp32-parity", action="store_true", help="skip the fp32 parity-mode throughput sub-measurement")
    ap.add_argument("--no-golden-parity", action="store_true", help="skip the measured parity band (`parity`: both modes on the reference's full-size vector)")
    ap.add_argument("--host-batch", action="store_true",
                    help="hand the model the reference's host-built batch (pho_idx + the host list pho_lens: an argsort and an H2D copy per "
                         "forward) instead of the device-side build_batch (model.set_pinyin_table)")
    ap.add_argument("--bucket-cap-mb", type=float, default=None, help="N > 1: split every gradient bucket into collectives of at most this many MB")
    ap.add_argument("--ddp-algo", default="allreduce", choices=["allreduce", "mesh"],
                    help="N > 1 gradient exchange: the library's all-reduce per bucket, or the direct (mesh) all-to-all reduce-scatter + all-gather")
    ap.add_argument("--dense-rows", action="store_true",
                    help="compute the transformer stacks over ALL rows of the padded batch, as the reference does (realise_set_engine(10, 0)); "
                         "default: the live 16-row blocks only - loss, live-row logits and gradients bit-identical (tests/test_round4_gpu.py)")
    ap.add_argument("--no-dense-rows-ab", action="store_true", help="skip the dense-rows sub-measurement of the default run")
    ap.add_argument("--no-overlap", action="store_true", help="run the bert / pho / glyph branches serially on one stream")
    ap.add_argument("--knob", action="append", default=[], help="diagnostic knob as name:key=value, e.g. engine:0=1 (realise_set_engine(0, 1)); names: engine, ln, nt8p; "
                                                                    "opt:fused=0 steps with the arena-level AdamW kernels + full operand refresh")
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
    from realise_amd.data import synthetic_batch, synthetic_pinyin_table
    from realise_amd.ddp import DistributedDataParallel
    from realise_amd.modeling import SpellBertPho2ResArch3
    from realise_amd.optim import FusedAdamW, get_linear_schedule_with_warmup
    lib = _capi.load()
    lib.realise_profile_mode(0 if args.profile_markers else 1)
    py_knobs = {}
    for kn in args.knob:
        name, kv = kn.split(":")
        k, v = kv.split("=")
        if name == "opt":
            py_knobs[k] = int(v)
        else:
            getattr(lib, "realise_set_" + name)(int(k), int(v))

    if args.dense_rows:
        lib.realise_set_engine(10, 0)
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
    # (round 5: the default path - loss.backward() hands a device scalar to the engine, which scales the head's gradients in the
    # kernels that store them: no host read, no pass over the logits gradient; assume_unit_loss_grad is no longer set here)
    wrapped = DistributedDataParallel(model, grad_dtype=args.grad_dtype, bucket_cap_mb=args.bucket_cap_mb, collect_stats=True, algo=args.ddp_algo) if ddp else model
    no_decay = ["bias", "LayerNorm.weight"]                 # run.py:146-151
    groups = [{"params": [p for n, p in model.named_parameters() if p.requires_grad and not any(nd in n for nd in no_decay)],
               "weight_decay": 0.0},
              {"params": [p for n, p in model.named_parameters() if p.requires_grad and any(nd in n for nd in no_decay)],
               "weight_decay": 0.0}]
    opt = FusedAdamW(model, groups, lr=5e-5, eps=1e-8, max_grad_norm=1.0)         # train.sh / run.py:333-339
    opt.fused_operand_copies = bool(py_knobs.get("fused", 1))
    model.trust_fused_optimizer = True                      # this loop makes no parameter write besides opt.step() (trainer.train() does the same)
    # (--knob opt:pipeline=1: the optimizer sweep pipelined under the next forward, realise_engine_adamw_pipelined - measured 0.03-0.2 ms
    # SLOWER per step: its LDS-using tiles cannot share a CU with two 80 KB GEMM workgroups, so the sweep only runs in their gaps; off)
    model.pipeline_optimizer = bool(py_knobs.get("pipeline", 0))
    model.train_logits = bool(py_knobs.get("train_logits", 0))   # the loop reads outputs[0] alone (run.py:191): K13, no [B, S, V] logits in training
    sched = get_linear_schedule_with_warmup(opt, 10000, 1000000)

    # The pinyin of a token is a function of its id: the batch's pho_idx / pho_lens come from a per-vocabulary table, as the reference's
    # build_batch derives them (models.py:797-804).  Default: the table lives on the device (model.set_pinyin_table) and the batch is
    # completed there from src_idx alone - no host work inside the step; --host-batch hands over the host-built tensors + list instead.
    ptable = synthetic_pinyin_table(cfg.vocab_size)
    batch = synthetic_batch(B, args.seq, seed=1000 + rank, pinyin_table=ptable)
    mean_len = float(sum(batch["pho_lens"])) / len(batch["pho_lens"])
    if not args.host_batch:
        model.set_pinyin_table(ptable)
        del batch["pho_idx"], batch["pho_lens"]
    for k in ("src_idx", "tgt_idx", "masks", "loss_masks", "pho_idx"):
        if k in batch:
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
    # live rows of the padded batch: rows before a sentence's last attended / loss position, and the 16-row blocks that hold any
    # (what row_liveness lists on the device; computed here from the same masks)
    mk = ((batch["masks"] == 1) | (batch["loss_masks"] == 1)).cpu()
    pos = torch.arange(1, args.seq + 1)[None, :]
    last = (mk * pos).max(dim=1).values                                  # [B]: 1 + last flagged position
    row_live = (torch.arange(args.seq)[None, :] < last[:, None]).reshape(-1)
    rows_frac = float(row_live.float().mean())
    blk16 = row_live[:T_ - T_ % 16].reshape(-1, 16).any(dim=1)
    blocks16_frac = float(blk16.float().mean())
    live_rows_on = (args.dtype == "bf16") and not args.dense_rows and (T_ % 64) == 0
    # round 6: the layer GEMMs walk the list of live ROWS (whole 128-row tiles of it); LayerNorm / attention / the weight-gradient
    # reductions still visit the live 16-row blocks - the GEMM figure is what the FLOP accounting follows
    row_list = live_rows_on and not any(kn.replace(" ", "") == "engine:10=1" for kn in args.knob)
    rows_tiled = (int(row_live.sum()) + 127) // 128 * 128 / float(T_)
    # executed rows of a layer's train step = forward GEMMs (row list) + data gradients (row list, except the GELU' gradient - a third of
    # their FLOPs - which keeps whole blocks) + weight gradients (blocks), a third of the FLOPs each
    stack_frac = ((rows_tiled * (1.0 + 2.0 / 3.0) + blocks16_frac * (1.0 / 3.0 + 1.0)) / 3.0 if row_list else blocks16_frac) if live_rows_on else 1.0
    dense_ab = None
    if live_rows_on and world == 1 and not ddp and not args.no_dense_rows_ab:
        # the same step with the transformer stacks over all rows (what the reference computes): realise_set_engine(10, 0)
        lib.realise_set_engine(10, 0)
        for _ in range(3):
            step()
        nd = max(5, min(args.steps, 20))
        td = timed_loop(lambda i: step(), nd, world, dev)
        lib.realise_set_engine(10, 2)
        step()
        dense_ab = {"value": round(B * nd / td, 2), "unit": "sentences/s", "ms_per_step": round(1e3 * td / nd, 3), "steps": nd,
                    "note": "same model, batch and step with every row of the padded batch computed in the transformer stacks "
                            "(bench.py --dense-rows / realise_set_engine(10, 0)): the reference's row count"}

    fwd = None
    if not args.no_forward:
        fwd = {}
        fsteps = max(5, args.steps)
        nom = fwd_flops_per_sentence(args.seq, mean_len) * B
        exe = fwd_flops_per_sentence(args.seq, mean_len, live) * B
        exe_live = fwd_flops_per_sentence(args.seq, mean_len, live, rows_tiled if live_rows_on else 1.0) * B
        for mode in ("eval", "train", "eval_live_rows"):
            if mode == "eval_live_rows" and not live_rows_on:
                continue
            model.train(mode == "train")
            model.eval_live_rows = mode == "eval_live_rows"
            model.static_weights = True                      # weights do not change between these forwards
            with torch.no_grad():
                for _ in range(2):
                    model(batch)
                t = timed_loop(lambda i: model(batch), fsteps, world, dev)
            ms = 1e3 * t / fsteps
            ex = exe_live if mode == "eval_live_rows" else exe
            fwd[mode] = {"ms": round(ms, 3), "sentences_per_s": round(B * world / (ms * 1e-3), 1),
                         "mfma_util_nominal": round(nom / (ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12), 4),
                         "mfma_util_executed": round(ex / (ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12), 4)}
        model.eval_live_rows = False
        fwd["note"] = ("forward only, batch %d x seq %d per GPU, loss + logits computed; eval: BatchNorm running statistics, no dropout, every "
                       "row of the padded batch (the reference's contract); train: batch statistics + dropout 0.1, no backward; "
                       "eval_live_rows: the same evaluation forward with model.eval_live_rows = True - the transformer stacks over the rows "
                       "up to every sentence's last attended position only (their logits and the loss bit-identical, the padding rows' "
                       "logits unspecified: run.py:262-270 cuts predictions at `lengths`), nominal = the dense graph's FLOPs over its time; "
                       "target (BASELINE.json north_star): >= 0.40 bf16 MFMA utilisation" % (B, args.seq))
        model.static_weights = False
        model.train()

    ddp_stats = wrapped.module.grad_sync.read_stats() if ddp and getattr(wrapped.module, "grad_sync", None) is not None else None

    parity = None
    if world == 1 and not ddp and not args.no_fp32_parity and args.dtype == "bf16":
        # the exact mode (v_mfma_f32_16x16x4_f32, fp32 everywhere): same model, same batch, same step - what the north-star tolerance costs
        m32 = SpellBertPho2ResArch3(cfg, compute_dtype="fp32", seed=0)
        m32.to(dev)
        m32.train()
        if not args.host_batch:
            m32.set_pinyin_table(ptable)
        g32 = [{"params": [p for n, p in m32.named_parameters() if p.requires_grad and not any(nd in n for nd in no_decay)], "weight_decay": 0.0},
               {"params": [p for n, p in m32.named_parameters() if p.requires_grad and any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
        o32 = FusedAdamW(m32, g32, lr=5e-5, eps=1e-8, max_grad_norm=1.0)

        def step32(i=0):
            m32(batch)[0].backward()
            o32.step()
            m32.zero_grad()

        for _ in range(2):
            step32()
        n32 = max(3, min(args.steps, 6))
        t32 = timed_loop(step32, n32, world, dev)
        with torch.no_grad():
            m32.eval()
            m32.static_weights = True
            m32(batch)
            tf32 = timed_loop(lambda i: m32(batch), n32, world, dev)
        parity = {"dtype": "fp32", "value": round(B * n32 / t32, 2), "unit": "sentences/s", "ms_per_step": round(1e3 * t32 / n32, 3), "steps": n32,
                  "eval_forward_ms": round(1e3 * tf32 / n32, 3), "eval_forward_sentences_per_s": round(B * n32 / tf32, 1),
                  "note": "fp32 parity mode (v_mfma_f32_16x16x4_f32, fp32 storage): the mode tests/test_round2_gpu.py / test_round3_gpu.py hold "
                          "to the north-star tolerance (logits <= 1e-3 of the reference's fp32 CPU path, arg-max ids equal wherever the "
                          "reference's top-1 / top-2 margin exceeds fp32 noise); same model, batch and step as `value`"}
        del m32, o32
        torch.cuda.empty_cache()

    if rank == 0:
        sent = world * B * args.steps
        step_nom = 3.0 * fwd_flops_per_sentence(args.seq, mean_len) * B
        step_exe = 3.0 * fwd_flops_per_sentence(args.seq, mean_len, live, stack_frac) * B
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
                       "branch_overlap": bool(overlap), "build_batch": "host" if args.host_batch else "device",
                       "optimizer": "FusedAdamW (%s)" % ("engine sweep: Linear weights + their bf16 operand copies in one pass" if opt.fused_operand_copies else "arena kernels + full operand refresh"),
                       "distinct_glyphs": uniq, "tokens": T_,
                       "padding_rows": (("layer GEMMs of the transformer stacks over the list of live ROWS (%.3f of the rows precede their sentence's "
                                         "last attended / loss position; LayerNorm, attention and the weight-gradient reductions over the live 16-row "
                                         "blocks: %.3f of the rows)" % (rows_frac, blocks16_frac)) if row_list else
                                        ("transformer stacks over the live 16-row blocks only (%.3f of the rows; %.3f of the rows precede their "
                                         "sentence's last attended / loss position)" % (blocks16_frac, rows_frac))) +
                                        ": loss, live-row logits and gradients bit-identical to the dense pass" if live_rows_on else "all rows computed (dense, as the reference)"},
            "model_flops_per_step_per_gpu": {"nominal": step_nom, "executed": step_exe,
                                             "note": "nominal = dense reference graph (3 x forward); executed: the glyph ResNet runs on the "
                                                     "%d distinct token ids of the %d tokens%s" % (uniq, T_, ", the 19 transformer layers on the live "
                                                     "rows / 16-row blocks (%.3f of the dense rows, FLOP-weighted)" % stack_frac if live_rows_on else "")},
            "model_mfma_util": round(step_nom / sec / (PEAK_BF16_TFLOPS * 1e12), 4),
            "model_mfma_util_executed": round(step_exe / sec / (PEAK_BF16_TFLOPS * 1e12), 4),
        }
        if fwd is not None:
            out["forward"] = fwd
        if parity is not None:
            out["fp32_parity"] = parity
        if world == 1 and not ddp and not args.no_golden_parity and not args.no_fp32_parity and args.dtype == "bf16":
            gp = golden_parity(dev)
            if gp is not None:
                out["parity"] = gp
        if dense_ab is not None:
            out["dense_rows"] = dense_ab
        if ddp_stats is not None:
            out["ddp"] = ddp_stats
        for name in ("conv_nt", "conv_tn"):                  # launch records charge the dense row count: scale to the live rows
            if name in fams:
                fams[name]["tflops_nominal"] = fams[name]["tflops"]
                fams[name]["tflops"] = fams[name]["tflops"] * live
                fams[name]["gflop_per_step_executed"] = fams[name]["gflop_per_step_nominal"] * live
                fams[name]["rows_live_frac"] = live
        if "gemm_nt" in fams:
            f = fams["gemm_nt"]
            traffic, tnote = pmc_traffic_per_launch(NT_KERNEL_PREFIXES)
            out["roofline"] = {"bound": "mfma",
                               "kernel": ("dense NT GEMM family: gemm_nt8_kernel<128x192, two per CU, live-row form> (the layer GEMMs of the three "
                                          "transformer stacks, forward and data gradients, over the live 16-row blocks) + gemm_nt8p_kernel<256x192> "
                                          "(persistent 8-wave ping-pong: classifier) + gemm_nt8_kernel<128x192> (classifier data gradient, GRU steps), "
                                          "v_mfma_f32_16x16x32_bf16") if live_rows_on else
                                         ("dense NT GEMM family: gemm_nt8p_kernel<256x192> (persistent 8-wave ping-pong: qkv, FFN-up, FFN-down "
                                          "dgrad, classifier) + gemm_nt8_kernel<128x192, two per CU> (N = 768 outputs) + "
                                          "gemm_nt_kernel<bf16, DenseLoader> (GRU steps), v_mfma_f32_16x16x32_bf16"),
                               "achieved": round(f["tflops"], 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(f["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": tnote,
                               "achieved_nominal": round(f["tflops_nominal"], 2), "frac_nominal": round(f["tflops_nominal"] / PEAK_BF16_TFLOPS, 4),
                               "avg_launch_us": round(f["avg_launch_us"], 2),
                               "launches_per_step": f["launches_per_step"],
                               "flops_per_launch": round(f["tflops"] * 1e12 * f["avg_launch_us"] * 1e-6),
                               "flops_per_launch_nominal": round(f["tflops_nominal"] * 1e12 * f["avg_launch_us"] * 1e-6),
                               "rocprof": rocprof_family_avg(NT_KERNEL_PREFIXES),
                               "timing": "event markers around each launch" if args.profile_markers else
                                         "HIP events attached to each dispatch (hipExtLaunchKernelGGL start/stop = the kernel's own begin/end timestamps)",
                               "note": "per-launch durations from every %dth timed step; those steps run the three model "
                                       "branches serially so each kernel is timed alone.  achieved / frac = EXECUTED flops (launches bounded "
                                       "by a device-side row count execute fewer rows than they book) / sum of durations; *_nominal = booked "
                                       "2.M.N.K.  frac can be recomputed as flops_per_launch / (avg_launch_us x peak); rocprof.kernel_avg_us_rocprof "
                                       "is the same family's average in the committed rocprofv3 kernel trace" % PROFILE_EVERY}
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
