#!/usr/bin/env python3
"""Scan the gfx950 ISA of every kernel for fetch queues the compiler drains by accident:

  * `LDS-DMA fetch ... s_waitcnt vmcnt(0) ... LDS-DMA fetch` with no barrier / MFMA / LDS read between them: a compiler-inserted full drain
    in the middle of a tile's fetches (hipcc emits it when a plain load is - or may be - pending next to the fetches);
  * compiler-inserted `s_waitcnt vmcnt(0)` inside a loop that also issues vector stores and loads (prefetch nullified): reported as a count,
    to be read by hand.

Runs on the CPU (hipcc cross-compiles): python tools/isa_wait_scan.py [file.hip ...]   (default: every realise_amd/csrc/*.hip)
A wait between ;;#ASMSTART / ;;#ASMEND is hand-written and not counted."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-gpu-rdc", "-S", "--cuda-device-only"]
DMA = re.compile(r"(buffer_load_\w+ .* lds|global_load_lds)")


def scan(asm):
    lines = asm.split("\n")
    fn, state, hits, in_asm = None, 0, {}, False
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            fn, state = m.group(1), 0
            continue
        if "#ASMSTART" in l:
            in_asm = True
        if "#ASMEND" in l:
            in_asm = False
            continue
        if fn is None:
            continue
        if DMA.search(l):
            if state == 2:
                hits[fn] = hits.get(fn, 0) + 1
            state = 1
        elif "s_waitcnt vmcnt(0)" in l and not in_asm:
            if state == 1:
                state = 2
        elif "s_barrier" in l or "v_mfma" in l or "ds_read" in l:
            state = 0
    return hits


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "realise_amd", "csrc", "*.hip")))
    bad = 0
    for f in files:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [f, "-o", out], capture_output=True, text=True)
            if r.returncode != 0:
                print("%s: compile failed\n%s" % (f, r.stderr[-500:]))
                bad += 1
                continue
            hits = scan(open(out).read())
        for k, v in sorted(hits.items()):
            dem = k
            for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
                try:
                    dem = subprocess.run([tool, k], capture_output=True, text=True).stdout.strip() or k
                    break
                except OSError:
                    continue
            print("%s: %d x fetch / vmcnt(0) / fetch in %s" % (os.path.basename(f), v, dem[:160]))
        if not hits:
            print("%s: clean" % os.path.basename(f))
    return 0


if __name__ == "__main__":
    sys.exit(main())
