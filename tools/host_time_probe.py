"""Host-side enqueue time of one train step (Python + ctypes + launch calls) against the GPU step time: the host must stay ahead."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realise_amd.config import RealiseConfig
from realise_amd.data import synthetic_batch
from realise_amd.modeling import SpellBertPho2ResArch3
from realise_amd.optim import FusedAdamW, get_linear_schedule_with_warmup

cfg = RealiseConfig()
model = SpellBertPho2ResArch3(cfg, compute_dtype="bf16", seed=0).to("cuda:0")
model.train()
model.assume_unit_loss_grad = True
batch = {k: (v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in synthetic_batch(64, 128, seed=1).items()}
opt = FusedAdamW(model, lr=5e-5, max_grad_norm=1.0)
sched = get_linear_schedule_with_warmup(opt, 10, 1000)
parts = {"forward": 0.0, "backward": 0.0, "optimizer": 0.0, "zero_grad": 0.0}


def step(acc):
    t0 = time.perf_counter()
    loss = model(batch)[0]
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step(); sched.step()
    t3 = time.perf_counter()
    model.zero_grad()
    t4 = time.perf_counter()
    if acc:
        parts["forward"] += t1 - t0; parts["backward"] += t2 - t1; parts["optimizer"] += t3 - t2; parts["zero_grad"] += t4 - t3


for _ in range(5):
    step(False)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N):
    step(True)
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print("host enqueue %.2f ms/step (%s), wall %.2f ms/step" % (host / N * 1e3, ", ".join("%s %.2f" % (k, v / N * 1e3) for k, v in parts.items()), total / N * 1e3))
# without back-pressure: one step at a time on an idle queue (host time stops before the synchronize)
for k in parts:
    parts[k] = 0.0
hs = 0.0
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(True)
    hs += time.perf_counter() - t0
torch.cuda.synchronize()
print("host enqueue on an idle queue %.2f ms/step (%s)" % (hs / 10 * 1e3, ", ".join("%s %.2f" % (k, v / 10 * 1e3) for k, v in parts.items())))
