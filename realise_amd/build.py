"""Build librealise_hip.so (gfx950) from realise_amd/csrc/*.hip with hipcc.

Cross-compiles without a GPU (seconds per file).  The shared object is written IN-TREE next
to this file so it travels with the repository snapshot to the GPU box; objects are cached
under realise_amd/_obj/ keyed by source mtime.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "librealise_hip.so")
SOURCES = ["gemm.hip", "gemm_nt8.hip", "gemm_nt8p.hip", "gemm_tn8.hip", "conv_wgrad_c64.hip", "conv_c64_nt.hip", "attention.hip", "ops.hip", "ops2.hip", "engine.hip", "capi.hip", "prof.hip"]
PROBE_SOURCES = ["gemm_nt8s.hip"]      # measured-and-rejected kernels with their own translation unit: the probe build only
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-fno-gpu-rdc"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newest_header():
    t = 0.0
    for d in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _compile(src, hdr_time, verbose, probes=False):
    obj = os.path.join(OBJ + ("_probes" if probes else ""), src.replace(".hip", ".o"))
    sp = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(sp), hdr_time):
        return obj, False
    cmd = [_hipcc()] + FLAGS + (["-DRL_PROBES=1"] if probes else []) + ["-c", sp, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, True


def build(force=False, verbose=False, probes=False):
    """probes=True builds librealise_hip_probes.so next to the production library: the same sources with -DRL_PROBES=1 (measured-and-
    rejected kernel variants + the no-fetch / no-MFMA probe modes, for tools/*_probe.cpp); the production library has none of them."""
    objdir = OBJ + ("_probes" if probes else "")
    lib = LIB.replace(".so", "_probes.so") if probes else LIB
    sources = SOURCES + (PROBE_SOURCES if probes else [])
    os.makedirs(objdir, exist_ok=True)
    if force:
        for f in os.listdir(objdir):
            os.remove(os.path.join(objdir, f))
    hdr_time = _newest_header()
    with ThreadPoolExecutor(max_workers=6) as ex:
        res = list(ex.map(lambda s: _compile(s, hdr_time, verbose, probes), sources))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(lib):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        # an instantiation the compiler silently dropped shows up only at dlopen time (undefined symbol): check it here
        chk = subprocess.run([sys.executable, "-c", "import ctypes, os, sys; ctypes.CDLL(sys.argv[1], mode=os.RTLD_NOW)", lib], capture_output=True, text=True)
        if chk.returncode != 0:
            raise RuntimeError("librealise_hip.so does not load:\n%s" % chk.stderr[-2000:])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, probes="--probes" in sys.argv))
