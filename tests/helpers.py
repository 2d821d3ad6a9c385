"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

from realise_amd.config import RealiseConfig
from realise_amd.data import synthetic_batch
from realise_amd.init import init_state_dict_numpy

N_SAMPLE = 192   # must match oracle/make_golden.py


def load_golden(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def sample_of(t):
    a = t.detach().to("cpu", torch.float64).reshape(-1)
    stride = max(1, a.numel() // N_SAMPLE)
    return a[::stride][:N_SAMPLE].to(torch.float32).numpy(), float(a.sum()), float(a.abs().sum())


def check_summary(g, key, t, atol, rtol=0.0, what=""):
    """Compare tensor t with the golden summary stored under key."""
    s, total, abssum = sample_of(t)
    ref = g[key + "/sample"]
    assert int(g[key + "/n"]) == t.numel(), "%s: numel %d != golden %d" % (key, t.numel(), int(g[key + "/n"]))
    err = np.abs(s - ref).max()
    tol = atol + rtol * np.abs(ref).max()
    assert err <= tol, "%s %s: sample max err %.3e > %.3e" % (what, key, err, tol)
    n = t.numel()
    assert abs(abssum - float(g[key + "/abssum"])) <= (atol + rtol * np.abs(ref).max()) * n, \
        "%s %s: abssum %.6e vs golden %.6e" % (what, key, abssum, float(g[key + "/abssum"]))
    return err


def golden_case_inputs(g, model_type):
    """Regenerate cfg / weights / batch of a golden case from its seeds."""
    B, S, seed, nl = int(g["meta/B"]), int(g["meta/S"]), int(g["meta/seed"]), int(g["meta/n_layers"])
    cfg = RealiseConfig(num_hidden_layers=nl, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd_np = init_state_dict_numpy(cfg, model_type, seed=seed, scheme="perturbed")
    batch = synthetic_batch(B, S, seed=seed, with_pho=(model_type == "arch3"))
    return cfg, sd_np, batch


def oracle_state_dict(sd_np, requires_grad=False):
    sd = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.array(v, copy=True))
        if requires_grad and t.dtype == torch.float32 and k not in ("char_images_multifonts", "char_images.weight") and "running_" not in k:
            t.requires_grad_(True)
        sd[k] = t
    sd["classifier.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    return sd
