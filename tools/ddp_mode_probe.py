"""Compute-side cost of the two data-parallel backward forms on ONE GPU (world size 1, RCCL: a one-rank all-reduce is a device copy):
REALISE_SIGNALLED_BACKWARD=1 (one branch-overlapped engine call + a 'bucket final' event per bucket) vs 0 (one engine call per
bucket, branches serial, side stream joined at every bucket).  Isolates what the staged form costs before any communication."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group(sys.argv[1] if len(sys.argv) > 1 else "nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from realise_amd.config import RealiseConfig
from realise_amd.data import synthetic_batch
from realise_amd.ddp import DistributedDataParallel
from realise_amd.modeling import SpellBertPho2ResArch3
from realise_amd.optim import FusedAdamW

cfg = RealiseConfig()
model = SpellBertPho2ResArch3(cfg, compute_dtype="bf16", seed=0).to("cuda:0")
model.train()
model.assume_unit_loss_grad = True
batch = {k: (v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in synthetic_batch(64, 128, seed=1).items()}
ddp = DistributedDataParallel(model)
opt = FusedAdamW(model, lr=5e-5, max_grad_norm=1.0)


def step():
    loss = ddp(batch)[0]
    loss.backward()
    opt.step()
    model.zero_grad()


for mode in ("1", "0", "1", "0"):
    model.signalled_backward = mode == "1"
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15):
        step()
    torch.cuda.synchronize()
    print("signalled_backward=%s: %.2f ms/step" % (mode, (time.perf_counter() - t0) / 15 * 1e3), flush=True)
dist.destroy_process_group()
