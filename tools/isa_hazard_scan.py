#!/usr/bin/env python3
"""Scan the gfx950 ISA of every kernel for a hazard hipcc does not cover: an INLINE-ASM vector-memory instruction that reads an SGPR
(buffer descriptor, scalar offset) fewer than 5 wait states after a VALU instruction wrote it (v_readlane_b32 - the restore of a spilled
SGPR -, v_readfirstlane_b32, a v_cmp into an SGPR pair).  The hazard recogniser pads compiler-generated memory instructions with s_nop;
the text of an asm statement it does not look into.  Round 5, gemm_nt8s.hip: the first of six asm bias loads of a tile ran right behind
the v_readlane restores of its descriptor and came back as zeros (wrong results in the first 16 columns of every wave's band).
Fix at the source: `s_nop 4` as the first instruction of the asm text.

Runs on the CPU: python tools/isa_hazard_scan.py [file.hip ...]   (default: every realise_amd/csrc/*.hip)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-gpu-rdc", "-S", "--cuda-device-only"]
VMEM = re.compile(r"^(buffer_|global_|flat_|scratch_)")
VALU_SGPR = re.compile(r"^(v_readlane_b32|v_readfirstlane_b32|v_cmp\w*|v_cmpx\w*|v_add_co\w*|v_sub_co\w*|v_addc_co\w*|v_subb_co\w*|v_mad_u64_u32|v_mad_i64_i32|v_div_scale\w*)\b")


def sregs(text):
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bs(\d+)\b", text):
        out.add(int(a))
    if re.search(r"\bvcc\b", text):
        out |= {106, 107}
    return out


def scan(asm):
    lines = asm.split("\n")
    fn, hist, in_asm, hits, nasm = None, [], False, [], 0      # hist: (wait states this instruction provides, SGPRs a VALU wrote)
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            fn, hist = m.group(1), []
            continue
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        ins = l.split(";")[0].strip()
        if not ins or ins.startswith(".") or ins.endswith(":") or fn is None:
            continue
        op = ins.split()[0]
        if in_asm and VMEM.match(op):
            nasm += 1
            ops = ins[len(op):]
            first = ops.split(",")[0]
            reads = sregs(ops) if op.startswith(("buffer_store", "global_store", "flat_store")) else sregs(ops[len(first):])
            ws = 0
            for k in range(len(hist) - 1, -1, -1):
                if ws >= 5:
                    break
                if hist[k][1] & reads:
                    hits.append((fn, ins, ws))
                    break
                ws += hist[k][0]
        wrote = set()
        if VALU_SGPR.match(op):
            dst = ins[len(op):].split(",")
            # SGPR destinations: the first operand (and the carry-out operand of the _co forms)
            wrote = sregs(dst[0]) | (sregs(dst[1]) if "_co" in op or op.startswith(("v_mad_u64", "v_mad_i64", "v_div_scale")) else set())
            if op.startswith(("v_cmp", "v_cmpx")) and "_e64" not in op and not sregs(dst[0]):
                wrote |= {106, 107}
        hist.append(((int(ins.split()[1]) + 1) if op == "s_nop" else 1, wrote))
        hist = hist[-8:]
    return hits, nasm


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "realise_amd", "csrc", "*.hip")))
    total = 0
    for f in files:
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [f, "-o", "-"], capture_output=True, text=True)
        if r.returncode != 0:
            print("%s: hipcc failed\n%s" % (f, r.stderr[-2000:]))
            return 2
        hits, nasm = scan(r.stdout)
        print("%-24s inline-asm vector-memory instructions %4d, behind a VALU write of an SGPR they read (< 5 wait states): %d" % (os.path.basename(f), nasm, len(hits)))
        for fn, ins, ws in hits:
            print("    %s\n        %s   (%d wait states)" % (fn[:150], ins, ws))
        total += len(hits)
    print("TOTAL hazards: %d" % total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
