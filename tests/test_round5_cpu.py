"""CPU tests added in round 5: the ISA scanners (tools/isa_hazard_scan.py, tools/isa_wait_scan.py) on hand-written instruction
streams - the two compiler behaviours the round's kernel work turned on (DESIGN.md sections 6.5, 6.6) - and the goldens of the
longer sequence lengths."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


HAZARD = """
_ZN2rl6kernelEv:
	v_readlane_b32 s8, v200, 39
	v_readlane_b32 s9, v200, 40
	v_readlane_b32 s10, v200, 41
	v_readlane_b32 s11, v200, 42
	;;#ASMSTART
	buffer_load_dwordx4 v[6:9], v3, s[8:11], 0 offen
	;;#ASMEND
	v_add_u32_e32 v3, 32, v0
	v_lshlrev_b32_e32 v4, 2, v3
	v_cmp_gt_i32_e32 vcc, s13, v3
	s_nop 1
	v_cndmask_b32_e32 v3, v198, v4, vcc
	;;#ASMSTART
	buffer_load_dwordx4 v[10:13], v3, s[8:11], 0 offen
	;;#ASMEND
.Lfunc_end0:
"""

PADDED = HAZARD.replace("\t;;#ASMSTART\n\tbuffer_load_dwordx4 v[6:9]", "\t;;#ASMSTART\n\ts_nop 4\n\tbuffer_load_dwordx4 v[6:9]", 1)

COMPILER_LOAD = """
_ZN2rl6kernelEv:
	v_readlane_b32 s8, v200, 39
	buffer_load_dwordx4 v[6:9], v3, s[8:11], 0 offen
.Lfunc_end0:
"""


def test_hazard_scan_flags_an_asm_load_behind_a_spill_restore_of_its_descriptor():
    """The first stream-K build (gemm_nt8s.hip): four v_readlane_b32 restore s[8:11], the asm buffer load that reads them follows at
    once - fewer than the 5 wait states a vector-memory instruction needs behind a VALU write of an SGPR it reads.  hipcc pads its own
    loads, not asm text.  The second asm load (seven wait states later) is clear; `s_nop 4` as the first asm instruction clears the first."""
    hz = _tool("isa_hazard_scan")
    hits, nasm = hz.scan(HAZARD)
    assert nasm == 2 and len(hits) == 1 and "v[6:9]" in hits[0][1] and hits[0][2] == 0
    hits, nasm = hz.scan(PADDED)
    assert nasm == 2 and hits == []
    hits, nasm = hz.scan(COMPILER_LOAD)              # a compiler-generated load is the hazard recogniser's business, not the scan's
    assert nasm == 0 and hits == []


DRAIN = """
_ZN2rl6kernelEv:
	buffer_load_dwordx4 v1, s[4:7], 0 offen lds
	s_waitcnt vmcnt(0)
	buffer_load_dwordx4 v2, s[4:7], 0 offen lds
	s_barrier
	buffer_load_dwordx4 v1, s[4:7], 0 offen lds
	buffer_load_dwordx4 v2, s[4:7], 0 offen lds
	s_waitcnt vmcnt(0)
	s_barrier
.Lfunc_end0:
"""


def test_wait_scan_flags_a_full_drain_between_two_fetches_of_a_tile():
    """`fetch / s_waitcnt vmcnt(0) / fetch` with nothing but the wait in between is a compiler-inserted drain in the middle
    of a K-tile's LDS-DMA fetches; the same wait behind the last fetch and in front of the barrier is the kernel's own."""
    ws = _tool("isa_wait_scan")
    hits = ws.scan(DRAIN)
    hits = hits[0] if isinstance(hits, tuple) else hits
    assert sum(hits.values()) == 1


def test_long_sequence_goldens_are_the_reference_at_256_and_512_positions():
    """tests/golden/arch3_b8s256_train.npz / arch3_b4s512_train.npz (oracle/make_golden_full.py train256 / train512, made by importing the
    reference): what the GPU tests of the tiled attention kernels are pinned to.  Shapes, finite values, the full gradient census."""
    for name, B, S in (("arch3_b8s256_train", 8, 256), ("arch3_b4s512_train", 4, 512)):
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        assert int(g["meta/B"]) == B and int(g["meta/S"]) == S and int(g["meta/train"]) == 1
        assert g["argmax"].shape == (B, S) and np.isfinite(float(g["loss"]))
        grads = [k for k in g.files if k.startswith("grad/") and k.endswith("/l2")]
        assert len(grads) >= 360 and all(np.isfinite(float(g[k])) for k in grads)
        assert len([k for k in g.files if k.startswith("gradnone/")]) == 9
