#!/usr/bin/env python3
"""Memory / wait / branch skeleton of one kernel in a hipcc -save-temps .s file, outside (default) or inside its MFMA loop.
usage: isa_skeleton.py file.s kernel-index [--loop]   (runs on the CPU; round 5: how gemm_nt8s.hip's tile boundary was checked)"""
import re, sys
f, idx = sys.argv[1], int(sys.argv[2]); loop = "--loop" in sys.argv
L = open(f).read().split("\n")
starts = [i for i, l in enumerate(L) if re.match(r"^_Z\w+:", l) and "kernel" in l]
st = starts[idx]; i = st
while not L[i].startswith(".Lfunc_end"): i += 1
body = L[st:i]
print(L[st][:160])
m = [k for k, l in enumerate(body) if "v_mfma" in l]
keys = ("s_waitcnt", "s_barrier", "buffer_load", "buffer_store", "global_load", "global_store", "scratch_", "s_sleep", "s_cbranch", "s_branch", ".LBB", "s_endpgm", "s_load", "flat_")
prev, run = None, 0
def flush():
    if prev: print("   %s x%d" % (prev, run) if run > 1 else "   " + prev)
rng = range(m[0], m[-1] + 1) if loop else list(range(0, m[0])) + list(range(m[-1] + 1, len(body)))
for k in rng:
    l = body[k].strip()
    if not loop and k == m[-1] + 1:
        flush(); prev = None; print("======== after the MFMA loop")
    if any(l.startswith(x) for x in keys):
        key = re.sub(r"\s+", " ", l.split(";")[0])
        key = re.sub(r"[vs]\[[0-9:]+\]|\b[vs][0-9]+\b", "R", key)
        if key == prev: run += 1
        else:
            flush(); prev, run = key, 1
flush()
