"""ctypes binding of librealise_hip.so (the C ABI declared in include/realise_hip.h).

This is the reference-side binding a maintainer would add: plain pointers
and sizes, no torch types.  The product path FAILS LOUDLY when the library is missing - there
is no CPU or PyTorch fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librealise_hip.so")

F32, BF16 = 0, 1
EPI_STORE, EPI_GELU, EPI_DROP_RESID, EPI_GELU_BWD = 0, 1, 2, 4


class GruStep(C.Structure):
    _fields_ = [("n_alive", C.c_int32), ("H", C.c_int32), ("Tp", C.c_int32), ("t", C.c_int32),
                ("table", C.c_void_p), ("pho_idx", C.c_void_p), ("perm", C.c_void_p), ("lens", C.c_void_p),
                ("gh", C.c_void_p), ("b_hh", C.c_void_p), ("h_prev", C.c_void_p), ("h_new", C.c_void_p), ("rzn", C.c_void_p), ("out", C.c_void_p),
                ("dout", C.c_void_p), ("dh", C.c_void_p), ("dgi", C.c_void_p), ("dgh", C.c_void_p), ("onehot", C.c_void_p)]


class Gate(C.Structure):
    _fields_ = [("B", C.c_int32), ("S", C.c_int32), ("H", C.c_int32),
                ("bert", C.c_void_p), ("pho", C.c_void_p), ("res", C.c_void_p), ("masks", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p),
                ("mean", C.c_void_p), ("msum", C.c_void_p), ("g", C.c_void_p), ("fused", C.c_void_p),
                ("dfused", C.c_void_p), ("dbert", C.c_void_p), ("dpho", C.c_void_p), ("dres", C.c_void_p), ("dz", C.c_void_p), ("dW", C.c_void_p),
                ("dbias", C.c_void_p)]


class AdamwGroup(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
                ("correct_bias", C.c_int32)]


class Epilogue(C.Structure):
    _fields_ = [("mode", C.c_int32), ("accumulate", C.c_int32), ("out", C.c_void_p), ("ldo", C.c_int64),
                ("out2", C.c_void_p), ("bias", C.c_void_p), ("aux", C.c_void_p), ("ldaux", C.c_int64),
                ("alpha", C.c_float), ("drop_seed", C.c_uint32), ("drop_thresh", C.c_uint32), ("drop_scale", C.c_float)]


class ConvGeom(C.Structure):
    _fields_ = [("src", C.c_void_p), ("img_index", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("rows", "Hr", "Wr", "Hs", "Ws", "C", "KH", "KW", "stride", "pad", "mode")]


class TnProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("I", C.c_int32), ("J", C.c_int32),
                ("out", C.c_void_p), ("ldo", C.c_int64), ("colsum", C.c_void_p)]


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("model_type", "dtype", "hidden", "heads", "intermediate", "vocab", "max_pos",
                                         "type_vocab", "bert_layers", "pho_layers", "out_layers", "num_fonts",
                                         "glyph_size", "pho_vocab")] + \
               [("hidden_dropout", C.c_float), ("attn_dropout", C.c_float), ("ln_eps", C.c_float),
                ("tie_classifier", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("B", C.c_int32), ("S", C.c_int32), ("Tp", C.c_int32), ("training", C.c_int32),
                ("want_dlogits", C.c_int32), ("seed", C.c_uint64),
                ("src_idx", C.c_void_p), ("masks", C.c_void_p), ("loss_masks", C.c_void_p), ("tgt_idx", C.c_void_p),
                ("pho_idx", C.c_void_p), ("pho_perm", C.c_void_p), ("pho_lens_sorted", C.c_void_p),
                ("n_alive", C.POINTER(C.c_int32)), ("loss_out", C.c_void_p), ("logits_out", C.c_void_p),
                ("n_alive_dev", C.c_void_p), ("eval_live_rows", C.c_int32), ("logits_f32_out", C.c_void_p)]


# every symbol include/realise_hip.h declares: name -> (restype, argtypes)
_P, _I, _L, _F, _U = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
SYMBOLS = {
    "realise_version": (C.c_char_p, []),
    "realise_gemm_nt": (_I, [_P, _I, _P, _L, _P, _L, _I, _I, _I, C.POINTER(Epilogue)]),
    "realise_conv_nt": (_I, [_P, _I, C.POINTER(ConvGeom), _P, _L, _I, _I, _I, C.POINTER(Epilogue)]),
    "realise_conv_dgrad_s2": (_I, [_P, _I, C.POINTER(ConvGeom), _P, _L, _I, C.POINTER(Epilogue)]),
    "realise_gemm_tn": (_I, [_P, _I, _P, _L, _P, _L, _I, _I, _I, _P, _L, _P, _L, _P]),
    "realise_gemm_tn_grouped": (_I, [_P, _I, _I, C.POINTER(TnProblem), _I]),
    "realise_gemm_nt_rows": (_I, [_P, _I, _P, _L, _P, _L, _I, _I, _I, C.POINTER(Epilogue), _P]),
    "realise_gemm_nt_live": (_I, [_P, _P, _L, _P, _L, _I, _I, _I, C.POINTER(Epilogue), _P, _P]),
    "realise_gemm_nt_live_rows": (_I, [_P, _P, _L, _P, _L, _I, _I, _I, C.POINTER(Epilogue), _P, _P]),
    "realise_gemm_nt_streamk": (_I, [_P, _P, _L, _P, _L, _I, _I, _I, C.POINTER(Epilogue), _P, _P, _P, _P, _I, _P]),
    "realise_gemm_nt_splitk": (_I, [_P, _P, _L, _P, _L, _I, _I, _I, _I, _P, _L, _P]),
    "realise_gemm_tn_grouped_live": (_I, [_P, _I, _I, C.POINTER(TnProblem), _I, _P, _P, _I, _I]),
    "realise_conv_tn": (_I, [_P, _I, _P, _L, C.POINTER(ConvGeom), _I, _I, _I, _P, _P, _L]),
    "realise_set_tn_transpose_read": (None, [_I]),
    "realise_set_nt_allow_n96": (None, [_I]),
    "realise_set_nt_probe": (None, [_I]),
    "realise_set_nt_variant": (None, [_I]),
    "realise_set_nt_group_m": (None, [_I]),
    "realise_set_nt8p": (None, [_I, _I]),
    "realise_set_ln": (None, [_I, _I]),
    "realise_set_engine": (None, [_I, _I]),
    "realise_layernorm_bwd_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_float, _P, _P, _P, _I, _I]),
    "realise_layernorm_fwd_live": (_I, [_P, _P, _P, _P, C.c_float, _P, _P, _P, _P, _I, _I]),
    "realise_layernorm_bwd_live": (_I, [_P, _P, _P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_float, _P, _P, _P, _P, _I, _I]),
    "realise_batchnorm_stats_ex": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, C.c_float, C.c_float] + [_P] * 9),
    "realise_batchnorm_bwd_ex": (_I, [_P, _P, _P, _I, _I, _I, _P, _I] + [_P] * 7 + [_P] * 7 + [_P, _P]),
    "realise_set_tn_probe": (None, [_I]),
    "realise_set_tn_split": (None, [_I]),
    "realise_set_tn_variant": (None, [_I]),
    "realise_set_tn_group_ring": (None, [_I]),
    "realise_set_conv_c64": (None, [_I]),
    "realise_set_attn_probe": (None, [_I]),
    "realise_set_branch_overlap": (None, [_I]),
    "realise_set_nt_wide_epilogue": (None, [_I]),
    "realise_set_glyph_dedup": (None, [_I]),
    "realise_set_wgrad_overlap": (None, [_I]),
    "realise_set_wgrad_group": (None, [_I]),
    "realise_set_dgrad_parity": (None, [_I]),
    "realise_attention_fwd": (_I, [_P, _I, _P, _P, _P, _L, _P, _P, _L, _P, _I, _I, _I, _U, _U, _F]),
    "realise_attention_bwd": (_I, [_P, _I, _P, _P, _P, _L, _P, _P, _P, _L, _P, _P, _P, _P, _P, _L, _I, _I, _I, _U, _U, _F]),
    "realise_mask_to_additive": (_I, [_P, _P, _P, _I]),
    "realise_layernorm_fwd": (_I, [_P, _I, _P, _P, _P, _F, _P, _P, _P, _I, _I]),
    "realise_layernorm_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I]),
    "realise_masked_ce": (_I, [_P, _I, _P, _L, _P, _P, _I, _I, _P, _P, _P]),
    "realise_gru_step_fwd": (_I, [_P, _I, C.POINTER(GruStep)]),
    "realise_gru_step_bwd": (_I, [_P, _I, C.POINTER(GruStep)]),
    "realise_gate_fwd": (_I, [_P, _I, C.POINTER(Gate)]),
    "realise_gate_bwd": (_I, [_P, _I, C.POINTER(Gate)]),
    "realise_batchnorm_fwd": (_I, [_P, _I, _P, _I, _I, _P, _P, _F, _F, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "realise_batchnorm_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "realise_embedding_bwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "realise_glyph_unique": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "realise_segment_sum": (_I, [_P, _I, _P, _P, _I, _I, _P, _P, _P]),
    "realise_argmax": (_I, [_P, _I, _P, _L, _I, _I, _P]),
    "realise_build_pho": (_I, [_P, _P, _I, _P, _P, _I, _I, _P, _P, _P, _P]),
    "realise_layout_count": (_I, [C.POINTER(Config)]),
    "realise_layout_entry": (_I, [C.POINTER(Config), _I, C.c_char_p, _I, C.POINTER(C.c_int32), C.POINTER(_L),
                                  C.POINTER(C.c_int32), C.POINTER(_L)]),
    "realise_arena_elems": (_L, [C.POINTER(Config), _I]),
    "realise_bucket_count": (_I, [C.POINTER(Config)]),
    "realise_bucket_bounds": (_I, [C.POINTER(Config), _I, C.POINTER(_L), C.POINTER(_L)]),
    "realise_engine_create": (_P, [C.POINTER(Config), _P, _P, _P, _P, _P, _P]),
    "realise_engine_destroy": (None, [_P]),
    "realise_engine_shadow_bytes": (_L, [_P]),
    "realise_engine_workspace_bytes": (_L, [_P, _I, _I, _I]),
    "realise_engine_forget_workspace": (None, [_P, _P]),
    "realise_engine_plan_installs": (_L, [_P]),
    "realise_debug_tn8_supported": (_I, [_L, _L, _I, _I, _I, _L]),
    "realise_debug_tn_list_lds": (_I, [_L]),
    "realise_engine_bind": (_I, [_P, _P, _P, _L]),
    "realise_engine_refresh_shadows": (_I, [_P, _P]),
    "realise_engine_refresh_shadows_ex": (_I, [_P, _P, _I]),
    "realise_engine_adamw": (_I, [_P, _P, _P, _P, _P, C.POINTER(AdamwGroup), _I, _L, _P, _F]),
    "realise_engine_adamw_pipelined": (_I, [_P, _P, _P, _P, _P, C.POINTER(AdamwGroup), _I, _L, _P, _F]),
    "realise_engine_sync_optimizer": (_I, [_P, _P]),
    "realise_engine_invalidate_frozen": (None, [_P]),
    "realise_engine_set_id_flag": (None, [_P, _P]),
    "realise_engine_set_grads_fresh": (None, [_P, _I]),
    "realise_engine_set_loss_grad": (None, [_P, _P]),
    "realise_engine_forward": (_I, [_P, _P, C.POINTER(Batch)]),
    "realise_engine_backward": (_I, [_P, _P, _I, _I]),
    "realise_engine_backward_signalled": (_I, [_P, _P, C.POINTER(C.c_void_p), _I]),
    "realise_engine_glyph_forward": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "realise_engine_glyph_backward": (_I, [_P, _P, _P]),
    "realise_engine_tap": (_I, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(_L)]),
    "realise_sumsq": (_I, [_P, _P, _L, _P]),
    "realise_adamw": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _L, _I, _P, _F]),
    "realise_clip_scale": (_I, [_P, _P, _L, _P, _F]),
    "realise_adamw_grouped": (_I, [_P, _P, _P, _P, _P, _L, _P, C.POINTER(AdamwGroup), _I, _L, _P, _F]),
    "realise_fill_f32": (_I, [_P, _P, _F, _L]),
    "realise_cast_to_f32": (_I, [_P, _I, _P, _P, _L]),
    "realise_profile_enable": (_I, [_I]),
    "realise_profile_disable": (None, []),
    "realise_profile_pause": (None, [_I]),
    "realise_profile_mode": (None, [_I]),
    "realise_profile_dump": (_I, [_I, _I, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "realise_profile_read": (_I, [_I, C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "realise_profile_read_ex": (_I, [_I, C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "realise_profile_dump_ex": (_I, [_I, _I, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


class RealiseHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64; it must be in the process BEFORE this library is loaded so both
    # resolve to ONE HIP runtime (loading /opt/rocm's copy first and torch's second gives two runtimes whose
    # streams / kernels do not mix: every launch on a torch stream then fails).
    import torch  # noqa: F401
    path = LIB_PATH
    if os.environ.get("REALISE_HIP_PROBES") == "1":      # measurement sessions: the probe build (python -m realise_amd.build --probes)
        path = LIB_PATH.replace(".so", "_probes.so")
    if not os.path.exists(path):
        raise RealiseHipError(
            "%s is missing. Build it with `python -m realise_amd.build` "
            "(or __graft_entry__.build()); the HIP path has no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError here == header / library mismatch
        fn.restype = res
        fn.argtypes = args
    # A/B knobs for measurements (defaults are the production settings)
    for env, fn in (("REALISE_WGRAD_OVERLAP", lib.realise_set_wgrad_overlap), ("REALISE_WGRAD_GROUP", lib.realise_set_wgrad_group), ("REALISE_TN_GROUP_RING", lib.realise_set_tn_group_ring), ("REALISE_CONV_C64", lib.realise_set_conv_c64), ("REALISE_DGRAD_PARITY", lib.realise_set_dgrad_parity), ("REALISE_GLYPH_DEDUP", lib.realise_set_glyph_dedup),
                    ("REALISE_NT_VARIANT", lib.realise_set_nt_variant), ("REALISE_NT_GROUP_M", lib.realise_set_nt_group_m)):
        if os.environ.get(env) is not None:
            fn(int(os.environ[env]))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RealiseHipError("%s failed with status %d" % (what, rc))


def make_config(cfg, model_type, dtype, tie=True):
    c = Config()
    c.model_type = 1 if model_type == "arch3" else 0
    c.dtype = dtype
    c.hidden, c.heads, c.intermediate = cfg["hidden_size"], cfg["num_attention_heads"], cfg["intermediate_size"]
    c.vocab, c.max_pos, c.type_vocab = cfg["vocab_size"], cfg["max_position_embeddings"], cfg["type_vocab_size"]
    c.bert_layers, c.pho_layers, c.out_layers = cfg["num_hidden_layers"], cfg["pho_layers"], cfg["out_layers"]
    c.num_fonts, c.glyph_size, c.pho_vocab = cfg["num_fonts"], cfg["glyph_size"], cfg["pho_vocab_size"]
    c.hidden_dropout, c.attn_dropout = cfg["hidden_dropout_prob"], cfg["attention_probs_dropout_prob"]
    c.ln_eps = cfg["layer_norm_eps"]
    c.tie_classifier = 1 if tie else 0
    return c


def layout(c):
    """[(name, arena, offset, shape)], arena sizes, bucket bounds from the library."""
    lib = load()
    n = lib.realise_layout_count(C.byref(c))
    out = []
    name = C.create_string_buffer(256)
    arena, ndim = C.c_int32(), C.c_int32()
    off = C.c_int64()
    shape = (C.c_int64 * 4)()
    for i in range(n):
        check(lib.realise_layout_entry(C.byref(c), i, name, 256, C.byref(arena), C.byref(off), C.byref(ndim), shape),
              "realise_layout_entry")
        out.append((name.value.decode(), arena.value, off.value, tuple(shape[k] for k in range(ndim.value))))
    sizes = [lib.realise_arena_elems(C.byref(c), a) for a in range(5)]
    buckets = []
    b0, b1 = C.c_int64(), C.c_int64()
    for i in range(lib.realise_bucket_count(C.byref(c))):
        check(lib.realise_bucket_bounds(C.byref(c), i, C.byref(b0), C.byref(b1)), "realise_bucket_bounds")
        buckets.append((b0.value, b1.value))
    return out, sizes, buckets
