"""How much do independent branches gain from running on separate HIP streams?  (a) a 12-layer SpellBert forward next to the glyph
ResNet forward, (b) two SpellBert forwards, (c) train-mode fwd+bwd pairs.  Two module instances = two engines = no shared scratch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realise_amd.config import RealiseConfig
from realise_amd.data import synthetic_batch
from realise_amd.modeling import SpellBert, SpellBertPho2ResArch3

dev = torch.device("cuda")
cfg = RealiseConfig()
bert1 = SpellBert(cfg, compute_dtype="bf16", seed=1).to(dev)
bert2 = SpellBert(RealiseConfig(num_hidden_layers=4), compute_dtype="bf16", seed=2).to(dev)
arch = SpellBertPho2ResArch3(RealiseConfig(num_hidden_layers=1), compute_dtype="bf16", seed=3).to(dev)
for m in (bert1, bert2, arch):
    m.static_weights = True
    m.assume_unit_loss_grad = True
batch = synthetic_batch(64, 128, seed=5, with_pho=False)
for k in ("src_idx", "tgt_idx", "masks", "loss_masks"):
    batch[k] = batch[k].to(dev)
src = batch["src_idx"]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def on(stream, fn):
    def run():
        with torch.cuda.stream(stream):
            fn()
    return run


def fwd(m):
    def run():
        with torch.no_grad():
            m(batch)
    return run


def glyph(train):
    def run():
        with torch.no_grad():
            arch.glyph_forward(src, training=train)
    return run


def trainstep(m):
    def run():
        m(batch)[0].backward()
        m.zero_grad()
    return run


def glyph_train():
    d = torch.full((64, 128, 768), 1e-3, device=dev, dtype=torch.bfloat16)
    def run():
        arch.glyph_forward(src, training=True)
        arch.glyph_backward(d)
        arch.zero_grad()
    return run


for name, a, b in [("bert12 eval fwd || glyph train-mode fwd", fwd(bert1.eval()), glyph(True)),
                   ("bert12 eval fwd || bert4 eval fwd", fwd(bert1.eval()), fwd(bert2.eval())),
                   ]:
    ta, tb = timeit(a), timeit(b)
    a1, b2 = on(s1, a), on(s2, b)
    def both():
        a1(); b2()
    tab = timeit(both)
    print("%-45s  A %.2f ms  B %.2f ms  serial %.2f  concurrent %.2f  (saves %.2f ms = %.0f%% of B)" % (name, ta, tb, ta + tb, tab, ta + tb - tab, 100 * (ta + tb - tab) / tb))
bert1.train(); bert2.train(); arch.train()
for name, a, b in [("bert12 fwd+bwd || glyph fwd+bwd", trainstep(bert1), glyph_train()),
                   ("bert12 fwd+bwd || bert4 fwd+bwd", trainstep(bert1), trainstep(bert2))]:
    ta, tb = timeit(a), timeit(b)
    a1, b2 = on(s1, a), on(s2, b)
    def both():
        a1(); b2()
    tab = timeit(both)
    print("%-45s  A %.2f ms  B %.2f ms  serial %.2f  concurrent %.2f  (saves %.2f ms = %.0f%% of B)" % (name, ta, tb, ta + tb, tab, ta + tb - tab, 100 * (ta + tb - tab) / tb))
