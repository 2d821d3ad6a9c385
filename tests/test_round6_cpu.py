"""CPU checks added in round 6: host-side support predicates of the weight-gradient kernels, the probe-only
stream-K entry point, the workspace-slot ABI."""
import ctypes as C

from realise_amd import _capi


def test_tn8_support_predicate_keeps_every_condition():
    """tn8_supported() lost `P >= 1024 && I >= 256 && J >= 128 && alignment of lda / ldb / I` to a mid-expression comment in round 5: the
    8-wave TN kernel then accepted shapes it was never written for instead of handing them to the 4-wave kernel."""
    lib = _capi.load()
    ok = lib.realise_debug_tn8_supported
    assert ok(768, 3072, 8192, 768, 3072, 3072) == 1            # an FFN weight gradient: the shape it was built for
    assert ok(7, 3072, 8192, 768, 3072, 3072) == 0              # unaligned lda: 16-byte buffer loads would straddle rows
    assert ok(768, 3071, 8192, 768, 3072, 3072) == 0            # unaligned ldb
    assert ok(768, 3072, 8192, 64, 3072, 3072) == 0             # I below the 256-row tile
    assert ok(768, 3072, 8192, 768, 64, 64) == 0                # J below the 128-column tile
    assert ok(768, 3072, 512, 768, 3072, 3072) == 0             # too few reduction rows for the three-stage ring
    assert ok(768, 3072, 8192, 772, 3072, 3072) == 0            # I % 8
    assert ok(768, 3072, 8192, 768, 3076, 3076) == 0            # J % 8
    assert ok(768, 3072, 8192, 768, 3072, 3074) == 0            # ldo % 4


def test_live_block_list_lds_is_sized_from_the_batch():
    """the live-block list of a listed TN launch sits in LDS: 4 KB (the bench shape, two 64 KB workgroups per CU) up to 16 KB = 4096
    blocks = 65536 bf16 token rows; longer lists are refused (the engine then builds no liveness tables: dense reductions)."""
    lib = _capi.load()
    f = lib.realise_debug_tn_list_lds
    assert f(1) == 4096 and f(512) == 4096 and f(1024) == 4096
    assert f(1025) == 5120 and f(2048) == 8192 and f(4096) == 16384
    assert f(4097) == -1 and f(1 << 20) == -1


def test_stream_k_entry_point_is_probe_build_only():
    """gemm_nt8s (measured slower on every layer shape) ships in librealise_hip_probes.so only: the production library
    keeps the symbol (include/realise_hip_debug.h) and refuses the call."""
    lib = _capi.load()
    if b"+probes" in lib.realise_version():
        return
    ep = _capi.Epilogue()
    assert lib.realise_gemm_nt_streamk(None, None, 0, None, 0, 256, 192, 128, C.byref(ep), None, None, None, None, 1, None) != 0
    lib.realise_set_engine(11, 1)              # the knob is inert in the production build (nothing to select)
    lib.realise_set_engine(11, 0)
