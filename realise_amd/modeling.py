"""Drop-in ``nn.Module`` shells for the reference's ``SpellBert`` / ``SpellBertPho2ResArch3``.

Same constructor (``Model(config)``), same ``forward(batch) -> (loss, logits) | (logits,)``
contract (src/models.py:50-73, 806-870), same ``state_dict`` keys, same
trainer-facing methods (``tie_cls_weight``, ``from_pretrained``, ``save_pretrained``,
``build_batch``; src/run.py:429-445).  All arithmetic happens in librealise_hip.so: the module
owns five flat arenas (parameters, gradients, ...) that the C engine indexes directly, every
``nn.Parameter`` is a *view* into them, and one ``torch.autograd.Function`` spans the whole
model so ``loss.backward()`` is a single C call that accumulates into the gradient arena.

There is no PyTorch / CPU fallback: forward on a non-CUDA device or without the built library
raises.
"""
import collections
import ctypes as C
import json
import os

import numpy as np
import torch
from torch import nn

from . import _capi
from .config import RealiseConfig
from .init import tensor_init, tensor_specs, synth_glyph_table

_DTYPES = {"bf16": (_capi.BF16, torch.bfloat16), "fp32": (_capi.F32, torch.float32)}
_AR_TRAIN, _AR_UNUSED, _AR_FROZEN, _AR_BUF_F32, _AR_BUF_I64 = range(5)


class _EngineLoss(torch.autograd.Function):
    """Whole-model autograd node: forward already ran inside the engine; backward runs the
    engine's backward stages, which ACCUMULATE into the module's gradient arena."""

    @staticmethod
    def forward(ctx, anchor, loss_value, module):
        ctx.module = module
        ctx.gen = module._fwd_gen
        return loss_value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        # the engine keeps the activations of ONE forward: a backward through an older loss (two forwards before one
        # backward, micro-batch sums, a second backward of the same loss) would silently reuse the wrong activations
        if ctx.gen != ctx.module._fwd_gen or ctx.module._bwd_done_gen == ctx.gen:
            raise RuntimeError("realise_amd: backward() of a loss whose forward is no longer the engine's latest (run forward and "
                               "backward in pairs; the workspace holds one batch's activations)")
        ctx.module._run_backward(grad_out)
        ctx.module._bwd_done_gen = ctx.gen
        return None, None, None


class _Container(nn.Module):
    """name-space node so parameters carry the reference's dotted names"""


class RealiseModule(nn.Module):
    model_type = "arch3"

    def __init__(self, config, compute_dtype=None, seed=0, init_scheme="reference", tie=True, logits_dtype=None):
        super().__init__()
        if not isinstance(config, RealiseConfig):
            config = RealiseConfig(**{k: getattr(config, k) for k in RealiseConfig.DEFAULTS if hasattr(config, k)})
        config.validate()
        self.config = config
        self.vocab_size = config.vocab_size
        self.compute_dtype = compute_dtype or os.environ.get("REALISE_DTYPE", "bf16")
        if self.compute_dtype not in _DTYPES:
            raise ValueError("compute_dtype must be 'bf16' or 'fp32'")
        self._tie = bool(tie)
        # dtype of the returned logits.  "auto" (default): fp32 in eval mode - the reference contract, `logits.cpu().numpy()` in
        # run.py:262 / test.py:140 needs it - and the compute dtype in training mode, where the trainer only reads the loss
        # (run.py:191).  "fp32" / "bf16" force one dtype in both modes.
        self.logits_dtype = logits_dtype or os.environ.get("REALISE_LOGITS_DTYPE") or "auto"
        # False: a training forward returns (loss, None) and never writes the [B, S, V] logits (K13; forward())
        self.train_logits = True
        if self.logits_dtype not in ("auto", "bf16", "fp32"):
            raise ValueError("logits_dtype must be 'auto', 'bf16' or 'fp32'")
        if self.compute_dtype == "fp32" and self.logits_dtype == "bf16":
            raise ValueError("bf16 logits need compute_dtype='bf16'")
        self.static_weights = False         # serving: promise that parameters only change through tracked paths (see _ensure_engine)
        self._fwd_gen = 0
        self._bwd_done_gen = -1
        self._ccfg = _capi.make_config(config, self.model_type, _DTYPES[self.compute_dtype][0], tie=self._tie)
        self._entries, self._sizes, self._buckets = _capi.layout(self._ccfg)
        self._engine = None
        self._shadow = None
        self._ws = None
        self._ws_key = None
        self._ws_cache = collections.OrderedDict()      # (B, S, Tp) -> workspace buffer, least recently used first
        self.workspace_slots = 3
        self._shadow_version = None
        self._linear_copies_current = False
        self._step_seed = int(seed) * 1000003 + 12345
        self.assume_unit_loss_grad = False
        self.grad_sync = None               # optional object with bucket_ready(i) / finish(), set by the DDP wrapper
        self.signalled_backward = os.environ.get("REALISE_SIGNALLED_BACKWARD", "1") != "0"    # DDP: one-call backward + bucket events
        self._last = None
        # ---- arenas (CPU first, like the reference: Model(config) -> load -> .to(device)) ----
        self._arenas = [torch.zeros(max(self._sizes[0], 1), dtype=torch.float32),
                        torch.zeros(max(self._sizes[1], 1), dtype=torch.float32),
                        torch.zeros(max(self._sizes[2], 1), dtype=torch.float32),
                        torch.zeros(max(self._sizes[3], 1), dtype=torch.float32),
                        torch.zeros(max(self._sizes[4], 1), dtype=torch.int64)]
        self._grads = torch.zeros(max(self._sizes[0], 1), dtype=torch.float32)
        self._anchor = torch.zeros((), requires_grad=True)
        self._views = {}
        self._register_views()
        self.init_weights(seed=seed, scheme=init_scheme)

    # ------------------------------------------------------------------ parameter plumbing
    def _view(self, arena, off, shape):
        n = int(np.prod(shape)) if len(shape) else 1
        return self._arenas[arena][off:off + n].view(shape)

    def _register_views(self):
        seen = {}
        for name, arena, off, shape in self._entries:
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Container())
                mod = mod._modules[p]
            leaf = parts[-1]
            key = (arena, off)
            if arena in (_AR_TRAIN, _AR_UNUSED, _AR_FROZEN):
                if key in seen:                       # tied classifier.weight
                    param = seen[key]
                else:
                    param = nn.Parameter(self._view(arena, off, shape), requires_grad=(arena != _AR_FROZEN))
                    seen[key] = param
                if leaf in mod._parameters:
                    del mod._parameters[leaf]
                mod.register_parameter(leaf, param)
                self._views[name] = (arena, off, shape, param)
            else:
                mod.register_buffer(leaf, self._view(arena, off, shape))
                self._views[name] = (arena, off, shape, None)

    def _rebind(self):
        """after the arenas moved (``.to(device)``): point every parameter/buffer at its slice again"""
        for name, (arena, off, shape, param) in self._views.items():
            v = self._view(arena, off, shape)
            if param is not None:
                param.data = v
                if param.grad is not None and arena == _AR_TRAIN:
                    n = v.numel()
                    param.grad = self._grads[off:off + n].view(shape)
            else:
                parts = name.split(".")
                mod = self
                for p in parts[:-1]:
                    mod = mod._modules[p]
                mod._buffers[parts[-1]] = v

    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, dtype=torch.float32))
        if probe.dtype != torch.float32:
            raise RuntimeError("master parameters stay fp32; choose the compute dtype with compute_dtype='bf16'|'fp32'")
        self._arenas = [fn(a) if a.is_floating_point() else a.to(probe.device) for a in self._arenas]
        self._grads = fn(self._grads)
        self._anchor = torch.zeros((), requires_grad=True, device=probe.device)
        self._rebind()
        self._drop_engine()
        return self

    def _drop_engine(self):
        if self._engine is not None:
            _capi.load().realise_engine_destroy(self._engine)
        self._engine = None
        self._shadow = None
        self._ws = None
        self._ws_key = None
        self._ws_cache = collections.OrderedDict()
        self._shadow_version = None
        self._linear_copies_current = False
        self._frozen_version = None

    def __del__(self):
        try:
            self._drop_engine()
        except Exception:
            pass

    @property
    def device(self):
        return self._arenas[0].device

    # ------------------------------------------------------------------ reference-facing API
    def init_weights(self, seed=0, scheme="reference"):
        """transformers/modeling_bert.py:496-506 + PyTorch defaults, from a torch-independent generator."""
        glyph = None
        with torch.no_grad():
            for name, shape, kind in tensor_specs(self.config, self.model_type):
                if name == "classifier.weight" and self._tie:
                    continue
                arena, off, shp, param = self._views[name]
                if kind == "glyph":
                    val = synth_glyph_table(shape[0], self.config.num_fonts, self.config.glyph_size, seed).reshape(shape)
                else:
                    val = tensor_init(name, shape, kind, self.config, seed, scheme)
                self._view(arena, off, shp).copy_(torch.from_numpy(np.asarray(val)).view(shp))

    def tie_cls_weight(self):
        """src/models.py:700-701.  The classifier is tied from construction in this implementation."""
        if not self._tie:
            raise RuntimeError("constructed with tie=False; build the module with tie=True to share the weight")

    def load_state_dict(self, state_dict, strict=True):
        self.sync_optimizer()
        sd = dict(state_dict)
        if self._tie and "classifier.weight" in sd and "bert.embeddings.word_embeddings.weight" in sd:
            del sd["classifier.weight"]        # the reference re-points it at the embedding after loading (run.py:431)
        missing, unexpected = [], []
        with torch.no_grad():
            for name, (arena, off, shape, param) in self._views.items():
                if name == "classifier.weight" and self._tie:
                    continue
                if name not in sd:
                    missing.append(name)
                    continue
                src = sd.pop(name)
                if src.numel() == 1 and len(shape) == 0:
                    src = src.reshape(())
                if tuple(src.shape) != tuple(shape):
                    raise RuntimeError("size mismatch for %s: %s vs %s" % (name, tuple(src.shape), tuple(shape)))
                self._view(arena, off, shape).copy_(src.to(self._arenas[arena].dtype))
        unexpected = list(sd.keys())
        if strict and (missing or unexpected):
            raise RuntimeError("load_state_dict: missing %s unexpected %s" % (missing[:5], unexpected[:5]))
        self._shadow_version = None
        self._linear_copies_current = False
        self._frozen_version = None
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    @classmethod
    def from_pretrained(cls, path, config=None, **kw):
        """transformers/modeling_utils.py:254-492: config.json + pytorch_model.bin."""
        if config is None:
            config = RealiseConfig.from_pretrained(path)
        for k in ("cache_dir", "from_tf", "force_download", "proxies", "output_loading_info"):      # HF plumbing run.py passes along
            kw.pop(k, None)
        model = cls(config, **kw)
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        model.load_state_dict(sd, strict=False)
        return model

    def save_pretrained(self, d):
        """transformers/modeling_utils.py:236-251."""
        os.makedirs(d, exist_ok=True)
        self.config.save_pretrained(d)
        torch.save({k: v.detach().cpu() for k, v in self.state_dict().items()}, os.path.join(d, "pytorch_model.bin"))

    def _glyph_param(self):
        name = "char_images.weight" if self.config.num_fonts == 1 else "char_images_multifonts"
        if name not in self._views:
            raise RuntimeError("this model has no glyph table (model_type %r)" % self.model_type)
        return self._views[name][3]

    def set_glyph_table(self, table):
        """install a pre-rendered glyph table: [V, F, 32, 32] (or [V, 1024] for the single-font model)"""
        p = self._glyph_param()
        with torch.no_grad():
            p.copy_(torch.as_tensor(np.asarray(table), dtype=torch.float32).reshape(p.shape))
        self._shadow_version = None
        self._linear_copies_current = False
        self._frozen_version = None

    def build_glyce_embed(self, vocab_dir, font_path, font_size=32):
        """src/models.py:703-734 (single-font model, ``char_images.weight [V, 1024]``): render ``vocab.txt`` with one font,
        blank unless the token is exactly one CJK character, standardise, install."""
        from . import glyph
        if self.config.num_fonts != 1:
            raise RuntimeError("build_glyce_embed is the num_fonts == 1 path (run.py:433-435); use build_glyce_embed_multifonts")
        vocab = glyph.read_vocab(vocab_dir)
        if len(vocab) != self.vocab_size:
            raise ValueError("vocab.txt has %d entries, the model %d" % (len(vocab), self.vocab_size))
        self.set_glyph_table(glyph.render_font_table(vocab, font_path, font_size, cjk_only=True).reshape(len(vocab), -1))

    def build_glyce_embed_multifonts(self, vocab_dir, num_fonts=None, use_traditional_font=False, font_size=32, font_paths=None,
                                     to_traditional=None):
        """src/models.py:737-758, called as ``model.build_glyce_embed_multifonts(vocab_dir, num_fonts, use_traditional_font)``
        (run.py:436-440).  The reference opens ``simhei.ttf`` / ``xiaozhuan.ttf`` from the working directory and converts to
        traditional forms with OpenCC; neither ships with the tree, so ``font_paths`` (list of paths, or (path, traditional)
        pairs) and ``to_traditional`` (str -> str) can be given explicitly.  A pre-rendered table goes through
        ``set_glyph_table``."""
        from . import glyph
        if not isinstance(vocab_dir, (str, bytes, os.PathLike)):
            raise TypeError("build_glyce_embed_multifonts(vocab_dir, num_fonts, use_traditional_font): vocab_dir must be a path; "
                            "install a pre-rendered table with set_glyph_table(table)")
        nf = self.config.num_fonts if num_fonts is None else int(num_fonts)
        if nf != self.config.num_fonts:
            raise ValueError("num_fonts %d does not match config.num_fonts %d" % (nf, self.config.num_fonts))
        vocab = glyph.read_vocab(vocab_dir)
        if len(vocab) != self.vocab_size:
            raise ValueError("vocab.txt has %d entries, the model %d" % (len(vocab), self.vocab_size))
        self.set_glyph_table(glyph.render_multifont_table(vocab, nf, bool(use_traditional_font), font_size, font_paths, to_traditional))

    @staticmethod
    def build_batch(batch, tokenizer=None):
        """SpellBert.build_batch (src/models.py:46-48): the BERT-only model needs nothing beyond make_features' tensors."""
        return batch

    def zero_grad(self, set_to_none=True):
        """run.py:211.  Default (``set_to_none=True``, torch's own default since 2.0): on the GPU nothing is zeroed here - the
        gradients are detached (``p.grad = None``) and the engine is told that the arena holds nothing to keep: its next backward
        zero-fills only what it accumulates into and stores the big Linear weight gradients instead of adding to them (no 680 MB
        memset, no read-modify-write).  ``flat_gradients()`` / ``bucket_views()`` / ``clip_grad_norm_()`` materialise the zeros if
        someone asks before that.  An explicit ``set_to_none=False`` is the strict form: the arena is zero-filled now and every
        ``p.grad`` view stays attached (holders of an old view read zeros), as is ``lazy_zero_grad = False`` for every call."""
        if set_to_none and self._engine is not None and self._grads.is_cuda and self.lazy_zero_grad:
            self._zero_pending = True
            for p in self._grad_params():
                p.grad = None
            return
        self.sync_optimizer()                # (a pipelined optimizer sweep may still be reading the gradients)
        self._grads.zero_()
        self._zero_pending = False
        if set_to_none:
            for p in self.parameters():
                p.grad = None

    lazy_zero_grad = True

    def _grad_params(self):
        if getattr(self, "_grad_param_list", None) is None:
            self._grad_param_list = [param for name, (arena, off, shape, param) in self._views.items() if arena == _AR_TRAIN and param is not None]
        return self._grad_param_list

    def _materialize_zero(self):
        if getattr(self, "_zero_pending", False):
            self._grads.zero_()
            self._zero_pending = False

    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ over the flat gradient arena (run.py:207): two kernels."""
        lib = _capi.load()
        self._materialize_zero()
        st = self._stream()
        nsq = torch.zeros(1, dtype=torch.float32, device=self.device)
        _capi.check(lib.realise_sumsq(st, self._grads.data_ptr(), self._grads.numel(), nsq.data_ptr()), "realise_sumsq")
        _capi.check(lib.realise_clip_scale(st, self._grads.data_ptr(), self._grads.numel(), nsq.data_ptr(), float(max_norm)), "realise_clip_scale")
        return nsq.sqrt()[0]

    # ------------------------------------------------------------------ engine management
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ensure_engine(self, B, S, Tp):
        lib = _capi.load()
        if self.device.type != "cuda":
            raise _capi.RealiseHipError("the ReaLiSe HIP path needs a GPU: move the module with .to('cuda') "
                                        "(there is no CPU fallback)")
        if self._engine is None:
            a = self._arenas
            self._engine = lib.realise_engine_create(C.byref(self._ccfg), a[0].data_ptr(), self._grads.data_ptr(),
                                                     a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), a[4].data_ptr())
            if not self._engine:
                raise _capi.RealiseHipError("realise_engine_create rejected the configuration")
            nbytes = lib.realise_engine_shadow_bytes(self._engine)
            self._shadow = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)      # padded rows rely on zero fill
            # id-range flag in host-mapped pinned memory: the device sets it, the host reads it without a synchronisation
            self._id_flag = torch.zeros(2, dtype=torch.int32).pin_memory()        # [0] id range, [1] fused-LayerNorm wait gave up
            lib.realise_engine_set_id_flag(self._engine, self._id_flag.data_ptr())
            self._ws = None
            self._ws_key = None
            self._ws_cache = collections.OrderedDict()
            self._frozen_version = None
        # One workspace buffer per (B, S, Tp) key, the `workspace_slots` most recently used ones kept (round 6): the engine installs a
        # plan INTO a buffer (one zero fill, ~0.3 - 2 ms) and remembers it per buffer, so a loop that alternates shapes - training
        # batches and evaluation batches of another size, the short last batch of a window (run.py:104-123), the glyph-only plan -
        # pays that fill once per shape, not at every switch.  9 GB per slot at B = 64, S = 128 (288 GB of HBM per GPU).
        key = (B, S, Tp)
        if key != self._ws_key or self._ws is None:
            cache = self._ws_cache
            if key in cache:
                cache.move_to_end(key)
            else:
                while len(cache) >= max(1, int(self.workspace_slots)):
                    _, old = cache.popitem(last=False)
                    lib.realise_engine_forget_workspace(self._engine, old.data_ptr())
                    del old
                need = lib.realise_engine_workspace_bytes(self._engine, B, S, Tp)
                cache[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws = cache[key]
            _capi.check(lib.realise_engine_bind(self._engine, self._shadow.data_ptr(), self._ws.data_ptr(), self._ws.numel()),
                        "realise_engine_bind")
        self._ws_key = key
        # Operand shadows (bf16 W / W^T copies): re-derived on EVERY forward, training or eval.  After `.to(device)` the
        # parameters are views whose version counters are detached from the arena's, so no torch-side counter sees a stock
        # optimizer, `p.data.copy_()`, an EMA swap or a manual re-init; one cast launch (~8 B / parameter) is cheaper than a
        # silently stale weight.  Serving code that never touches the parameters can set `model.static_weights = True`: the
        # refresh then only follows tracked changes (load_state_dict, FusedAdamW, mark_parameters_updated()).
        # (The frozen glyph table - 65 M floats, requires_grad=False - keeps the tracked rule: its NHWC image is rebuilt after
        # load_state_dict / build_glyce_embed* / set_glyph_table, or an in-place op the arena's version counter sees.)
        ver = (self._arenas[0]._version, self._arenas[2]._version)
        if getattr(self, "_frozen_version", None) != ver[1]:
            lib.realise_engine_invalidate_frozen(self._engine)
            self._frozen_version = ver[1]
        if self.training or not self.static_weights or self._shadow_version != ver:
            # Only a loop that OWNS every parameter write opts into `trust_fused_optimizer` (trainer.train(), bench.py): FusedAdamW's step
            # then wrote the Linear weights' operand copies in the pass that updated them (realise_engine_adamw) and what is left to
            # re-derive is the conv-weight copies.  The default (False) keeps the guarantee of the comment above: every copy, every forward.
            linear_current = 1 if (self._linear_copies_current and self.trust_fused_optimizer) else 0
            _capi.check(lib.realise_engine_refresh_shadows_ex(self._engine, self._stream(), linear_current), "realise_engine_refresh_shadows")
            self._linear_copies_current = False
            self._shadow_version = ver

    def _raise_on_bad_ids(self):
        """nn.Embedding's IndexError, raised lazily: the engine replaced an out-of-range src_idx / pho_idx by 0 in an earlier
        step and flagged it (include/realise_hip.h: realise_engine_set_id_flag)"""
        f = getattr(self, "_id_flag", None)
        if f is not None and int(f[1]) != 0:
            f[1] = 0                          # (only the flag being reported: a bad id seen in the same window still raises below / next time)
            raise RuntimeError("a workgroup of a fused dense + LayerNorm launch (or a stream-K layer GEMM: realise_set_engine(11, 1)) gave up waiting for another workgroup in an "
                               "earlier step (results of that step are invalid); realise_set_engine(8, 0) runs the two-launch form")
        if f is not None and int(f[0]) != 0:
            f[0] = 0
            raise IndexError("index out of range in self: a src_idx outside [0, %d) or a pho_idx outside [0, %d) reached the "
                             "model in an earlier forward" % (self.vocab_size, self._ccfg.pho_vocab))

    # strict_ids = True: backward() waits for the forward's range check before it enqueues anything, so a batch with an out-of-range id
    # raises before a single gradient or weight is touched (the reference's behaviour) at the price of one host synchronisation per
    # step; the default polls the flag without waiting and may report a bad batch one step late.
    strict_ids = False

    def check_ids(self):
        """wait for the queued steps and raise IndexError now if one of them saw an out-of-range id"""
        torch.cuda.synchronize(self.device)
        self._raise_on_bad_ids()

    # False (default): every forward re-derives all operand copies, also right after a FusedAdamW step - a p.data write, an EMA swap or a
    # second optimizer between step() and forward() is always seen.  True (set by trainer.train() and bench.py, whose loops make no such
    # write): the forward after a FusedAdamW step trusts the copies that step wrote and re-derives the conv-weight copies only.
    trust_fused_optimizer = False
    _linear_copies_current = False
    # True (needs trust_fused_optimizer; nobody sets it by default): FusedAdamW's engine sweep runs PIPELINED across the step boundary - on
    # the engine's side stream, in the order the next forward consumes the parameters, that forward waiting piece by piece
    # (realise_engine_adamw_pipelined).  Built for the 1.1 ms the sweep holds the caller's stream at the full model size; MEASURED 0.03-0.2
    # ms/step slower than the plain sweep (the sweep's tiles use LDS and cannot share a CU with two 80 KB GEMM workgroups -
    # it runs in their gaps, and what overlaps competes for the same memory system), so it stays opt-in.  Until the next forward
    # nothing but the engine may then touch parameters, gradients or optimizer moments; state_dict() / save_pretrained() /
    # load_state_dict() / zero_grad(set_to_none=False) / sync_optimizer() order the caller's stream behind the sweep first.
    pipeline_optimizer = False
    # True: evaluation forwards compute the transformer stacks over the live rows only, as bf16 training steps do (realise_batch.
    # eval_live_rows): the logits of a sentence's real tokens and the loss are bit-identical to the dense forward's, the logits rows of the
    # padding behind its last attended position are finite and meaningless - what run.py:262-270 cuts off at `lengths` anyway.  Off by
    # default (the reference returns defined values there); bench.py times both forms.
    eval_live_rows = False

    def sync_optimizer(self):
        """order the current stream behind a pending pipelined optimizer sweep (no-op without one)"""
        if getattr(self, "_engine", None) is not None and self._grads.is_cuda:
            _capi.check(_capi.load().realise_engine_sync_optimizer(self._engine, self._stream()), "realise_engine_sync_optimizer")

    def state_dict(self, *args, **kwargs):
        self.sync_optimizer()
        return super().state_dict(*args, **kwargs)

    def mark_parameters_updated(self, frozen=True, linear_copies_current=False):
        """call after mutating parameters through a path torch's version counter cannot see: raw pointers, or ``p.data.copy_()``
        on a parameter view (``.data`` carries its own version counter).  Weight operands are re-derived at the next forward and
        so is the NHWC image of the frozen glyph table - the one tensor that is NOT refreshed on every forward (65 M floats), so
        writing ``char_images_multifonts.data`` by hand instead of ``set_glyph_table()`` / ``build_glyce_embed*()`` needs this call.
        ``frozen=False`` (what FusedAdamW passes every step) leaves the glyph table's image alone: an optimizer never writes it."""
        self._shadow_version = None
        self._linear_copies_current = bool(linear_copies_current)
        if frozen:
            self._frozen_version = None

    def tap(self, name):
        """named internal activation of the last forward as a torch tensor (parity tests)"""
        lib = _capi.load()
        ptr, n = C.c_void_p(), C.c_int64()
        _capi.check(lib.realise_engine_tap(self._engine, name.encode(), C.byref(ptr), C.byref(n)), "tap " + name)
        esz = 2 if self.compute_dtype == "bf16" else 4
        off = ptr.value - self._ws.data_ptr()
        return self._ws[off:off + n.value * esz].view(_DTYPES[self.compute_dtype][1])

    # ------------------------------------------------------------------ forward / backward
    def _dev(self, t):
        if not torch.is_tensor(t):
            t = torch.as_tensor(t)
        return t.to(device=self.device, dtype=torch.int64).contiguous()

    def forward(self, batch):
        self._raise_on_bad_ids()
        src = self._dev(batch["src_idx"])
        B, S = src.shape
        masks = self._dev(batch["masks"])
        tgt = self._dev(batch["tgt_idx"]) if "tgt_idx" in batch else None
        loss_masks = self._dev(batch["loss_masks"]) if tgt is not None else None
        keep = [src, masks, tgt, loss_masks]
        cb = _capi.Batch()
        cb.B, cb.S, cb.Tp = B, S, 1
        training = bool(self.training)
        need_grad = training and tgt is not None and torch.is_grad_enabled()
        cb.training = 1 if training else 0
        cb.want_dlogits = 1 if need_grad else 0
        cb.eval_live_rows = 1 if (not training and self.eval_live_rows) else 0
        self._step_seed += 1
        cb.seed = self._step_seed & 0xFFFFFFFFFFFFFFFF
        cb.src_idx, cb.masks = src.data_ptr(), masks.data_ptr()
        cb.tgt_idx = tgt.data_ptr() if tgt is not None else None
        cb.loss_masks = loss_masks.data_ptr() if loss_masks is not None else None
        if self.model_type == "arch3" and "pho_idx" not in batch and getattr(self, "_pho_table", None) is not None:
            batch = self.build_batch_device(batch)
        if self.model_type == "arch3" and "_pho_device" in batch:
            pho_idx = batch["pho_idx"]
            perm_d, lens_d, alive_d = batch["_pho_device"]
            if pho_idx.shape[0] != B * S:
                raise ValueError("pho_idx must have B*S rows")
            keep += [pho_idx, perm_d, lens_d, alive_d]
            cb.Tp = int(pho_idx.shape[1])
            cb.pho_idx, cb.pho_perm, cb.pho_lens_sorted = pho_idx.data_ptr(), perm_d.data_ptr(), lens_d.data_ptr()
            cb.n_alive = None
            cb.n_alive_dev = alive_d.data_ptr()
        elif self.model_type == "arch3":
            pho_idx = self._dev(batch["pho_idx"])
            lens = np.asarray(batch["pho_lens"], dtype=np.int32)         # stays a HOST list in the reference (run.py:189)
            if pho_idx.shape[0] != B * S or lens.shape[0] != B * S:
                raise ValueError("pho_idx / pho_lens must have B*S rows")
            Tp = int(pho_idx.shape[1])
            if lens.min() < 1 or lens.max() > Tp:
                raise ValueError("pho_lens out of range")            # pack_padded_sequence would raise too
            # build_batch makes pho_idx as wide as the batch's longest pinyin (models.py:797-804: 4 .. 7 columns from batch to batch), and the
            # engine's workspace plan is keyed by (B, S, Tp): every new key re-zeroes the workspace (~2 ms at B = 64, S = 128).  Widen to the
            # widest batch seen so far with pad columns (id 0, beyond every length: those GRU steps have no live row and are not launched),
            # so the key settles after a few batches.
            tp_high = max(Tp, getattr(self, "_tp_high", 0))
            self._tp_high = tp_high
            if tp_high > Tp:
                pho_idx = torch.nn.functional.pad(pho_idx, (0, tp_high - Tp))
                Tp = tp_high
            perm = np.argsort(-lens, kind="stable").astype(np.int32)
            lens_sorted = lens[perm]
            alive = (C.c_int32 * Tp)(*[int((lens > t).sum()) for t in range(Tp)])
            perm_d = torch.from_numpy(perm).to(self.device)
            lens_d = torch.from_numpy(np.ascontiguousarray(lens_sorted)).to(self.device)
            keep += [pho_idx, perm_d, lens_d, alive]
            cb.Tp = Tp
            cb.pho_idx, cb.pho_perm, cb.pho_lens_sorted = pho_idx.data_ptr(), perm_d.data_ptr(), lens_d.data_ptr()
            cb.n_alive = alive
        self._ensure_engine(B, S, cb.Tp)
        tdt = _DTYPES[self.compute_dtype][1]
        self._fwd_gen += 1
        # K13 (round 6): a training forward whose caller reads the loss alone (`model.train_logits = False`, what realise_amd.trainer
        # sets: run.py:191 takes outputs[0]) does not produce the [B, S, V] logits at all - the classifier runs over the loss rows
        # straight into the gradient buffer and the tuple's second entry is None.  The default keeps the reference's tuple.
        no_logits = (need_grad and not self.train_logits and self.compute_dtype == "bf16" and self.vocab_size % 8 == 0
                     and self.vocab_size <= 22528 and B * S <= 65536)
        loss = torch.zeros((), dtype=torch.float32, device=self.device) if tgt is not None else None
        if no_logits:
            logits = None
            cb.logits_out = None
        else:
            logits = torch.empty((B, S, self.vocab_size), dtype=tdt, device=self.device)
            cb.logits_out = logits.data_ptr()
        cb.loss_out = loss.data_ptr() if loss is not None else None
        # reference contract: fp32 logits (models.py:859) from the bf16 engine - written by the classifier kernel's own epilogue next to
        # the bf16 ones the loss reads (realise_batch.logits_f32_out, round 6: no cast pass over [B, S, V])
        want = self.logits_dtype if self.logits_dtype != "auto" else ("fp32" if not training else self.compute_dtype)
        wide = None
        if not no_logits and want != self.compute_dtype:
            wide = torch.empty((B, S, self.vocab_size), dtype=torch.float32, device=self.device)
            cb.logits_f32_out = wide.data_ptr()
        _capi.check(_capi.load().realise_engine_forward(self._engine, self._stream(), C.byref(cb)), "realise_engine_forward")
        self._last = keep
        if no_logits:
            return (_EngineLoss.apply(self._anchor, loss, self), None)
        if wide is not None:
            logits = wide
        if tgt is None:
            return (logits,)
        if need_grad:
            loss = _EngineLoss.apply(self._anchor, loss, self)
        return (loss, logits)

    def glyph_forward(self, src_idx, training=None):
        """BASELINE configs[3]: the glyph ResNet alone - ``resnet(char_images_multifonts[src_idx])`` (src/models.py:829-836,
        src/char_cnn.py:46-55) -> [B, S, 768] in the compute dtype, before ``resnet_layernorm``."""
        if self.model_type != "arch3":
            raise RuntimeError("glyph_forward needs the full model")
        src = self._dev(src_idx)
        B, S = src.shape
        self._ensure_engine(B, S, -1)
        training = bool(self.training if training is None else training)
        out = torch.empty((B, S, self.config.hidden_size), dtype=_DTYPES[self.compute_dtype][1], device=self.device)
        self._fwd_gen += 1
        _capi.check(_capi.load().realise_engine_glyph_forward(self._engine, self._stream(), src.data_ptr(), B, S, 1 if training else 0,
                                                              out.data_ptr()), "realise_engine_glyph_forward")
        self._glyph_fwd_gen = self._fwd_gen if training else None
        self._last = [src]
        return out

    def glyph_backward(self, d_res):
        """accumulates the conv / BatchNorm parameter gradients of the last training ``glyph_forward`` for d_res [B, S, 768]"""
        if getattr(self, "_glyph_fwd_gen", None) != self._fwd_gen:
            raise RuntimeError("glyph_backward needs the activations of a training glyph_forward; another forward ran since (or none did)")
        d = d_res.to(device=self.device, dtype=_DTYPES[self.compute_dtype][1]).contiguous()
        self._begin_gradient_pass()
        self._glyph_fwd_gen = None
        _capi.check(_capi.load().realise_engine_glyph_backward(self._engine, self._stream(), d.data_ptr()), "realise_engine_glyph_backward")
        self._attach_grads()

    def set_pinyin_table(self, table):
        """Enable the device-side ``build_batch``: ``table`` is a ``realise_amd.pinyin.PinyinTable`` of this vocabulary.
        A batch without ``pho_idx`` / ``pho_lens`` is then completed on the GPU from ``src_idx`` alone."""
        if table.table.shape[0] != self.vocab_size:
            raise ValueError("pinyin table has %d rows, vocabulary has %d" % (table.table.shape[0], self.vocab_size))
        self._pho_table = torch.from_numpy(table.table).to(self.device)
        self._pho_vlens = torch.from_numpy(table.lens).to(self.device)

    def build_batch_device(self, batch):
        """models.py:797-804 on the device: adds ``pho_idx`` [B*S, 7] and the length-sorted bookkeeping of the GRU
        (device tensors; there is no host ``pho_lens`` list on this path)."""
        if getattr(self, "_pho_table", None) is None:
            raise RuntimeError("set_pinyin_table() first")
        src = self._dev(batch["src_idx"])
        T_, Tw = src.numel(), int(self._pho_table.shape[1])
        pho_idx = torch.empty((T_, Tw), dtype=torch.int64, device=self.device)
        perm = torch.empty(T_, dtype=torch.int32, device=self.device)
        lens_sorted = torch.empty(T_, dtype=torch.int32, device=self.device)
        alive = torch.empty(Tw, dtype=torch.int32, device=self.device)
        _capi.check(_capi.load().realise_build_pho(self._stream(), src.data_ptr(), T_, self._pho_table.data_ptr(), self._pho_vlens.data_ptr(),
                                                   self.vocab_size, Tw, pho_idx.data_ptr(), perm.data_ptr(), lens_sorted.data_ptr(),
                                                   alive.data_ptr()), "realise_build_pho")
        batch = dict(batch)
        batch["src_idx"] = src
        batch["pho_idx"] = pho_idx
        batch["_pho_device"] = (perm, lens_sorted, alive)
        return batch

    @torch.no_grad()
    def decode(self, batch_or_logits):
        """Arg-max ids [B, S] (int64, on the device) of a batch or of logits already computed: the device-side form of
        ``np.argmax(logits.cpu().numpy(), -1)`` (run.py:262-263); first maximum wins, as in numpy."""
        self._raise_on_bad_ids()
        logits = batch_or_logits if torch.is_tensor(batch_or_logits) else self(batch_or_logits)[-1]
        if logits.dtype not in (torch.float32, torch.bfloat16) or not logits.is_cuda:
            raise TypeError("decode needs float32 / bfloat16 logits on the GPU")
        logits = logits.contiguous()
        V = logits.shape[-1]
        rows = logits.numel() // V
        ids = torch.empty(logits.shape[:-1], dtype=torch.int64, device=logits.device)
        _capi.check(_capi.load().realise_argmax(self._stream(), 0 if logits.dtype == torch.float32 else 1, logits.data_ptr(), V, rows, V,
                                                ids.data_ptr()), "realise_argmax")
        return ids

    def _attach_grads(self):
        views = getattr(self, "_grad_views", None)
        if views is None or views[0] is not self._grads:
            # the gradient views are built once per arena and re-attached by assignment (zero_grad() detaches them every step)
            vs = []
            for name, (arena, off, shape, param) in self._views.items():
                if arena == _AR_TRAIN and param is not None:
                    vs.append((param, self._grads[off:off + int(np.prod(shape))].view(shape)))
            views = self._grad_views = (self._grads, vs)
        for param, g in views[1]:
            if param.grad is None:
                param.grad = g

    def _begin_gradient_pass(self):
        """detached gradients (our zero_grad(), or an optimizer's zero_grad(set_to_none=True)) mean "start from zero": the engine
        does that itself - a one-launch partial fill + overwriting weight-gradient GEMMs (realise_engine_set_grads_fresh)"""
        sentinel = self._views["classifier.bias"][3]
        if sentinel.grad is None or getattr(self, "_zero_pending", False):
            _capi.load().realise_engine_set_grads_fresh(self._engine, 1)
            self._zero_pending = False

    def _run_backward(self, grad_out):
        lib = _capi.load()
        if self.strict_ids:
            self.check_ids()
        else:
            self._raise_on_bad_ids()
        self._begin_gradient_pass()
        # the incoming d loss stays on the device: the engine multiplies it into the three gradients that leave the classifier head
        # (no host read, no pass over the cross-entropy gradient rows); assume_unit_loss_grad skips even that (plain loss.backward())
        if self.assume_unit_loss_grad or grad_out is None:
            self._loss_grad_keep = None
            lib.realise_engine_set_loss_grad(self._engine, None)
        else:
            g = grad_out.detach().to(device=self.device, dtype=torch.float32).reshape(1).contiguous()
            self._loss_grad_keep = g              # (kept alive until the next backward replaces it: the kernels read it later)
            lib.realise_engine_set_loss_grad(self._engine, g.data_ptr())
        st = self._stream()
        n = len(self._buckets)
        if self.grad_sync is None:
            _capi.check(lib.realise_engine_backward(self._engine, st, 0, -1), "realise_engine_backward")
        else:
            events = self.grad_sync.bucket_events() if hasattr(self.grad_sync, "bucket_events") else None
            if events is not None and self.signalled_backward:
                # one engine call (three branch streams + deferred weight gradients stay overlapped); the engine records a "bucket
                # final" event per bucket and the communication stream all-reduces them in a fixed order as they complete
                import ctypes as _C
                handles = (_C.c_void_p * n)(*[_C.c_void_p(e.cuda_event) for e in events])
                _capi.check(lib.realise_engine_backward_signalled(self._engine, st, handles, n), "realise_engine_backward_signalled")
                self.grad_sync.buckets_signalled(self._bucket_comm_order(n))
            else:
                for i in range(n):
                    _capi.check(lib.realise_engine_backward(self._engine, st, i, i), "realise_engine_backward")
                    self.grad_sync.bucket_ready(i)
            self.grad_sync.finish()
        self._attach_grads()

    @staticmethod
    def _bucket_comm_order(n):
        """order in which the buckets are all-reduced (the same on every rank).  arch3: output_block first, then the bert groups
        interleaved with the shorter pinyin and glyph branches that run next to them, the embeddings (+ tied classifier) last"""
        if n < 5:
            return list(range(n))
        # buckets: 0 output_block | 1 gate + glyph | 2 pinyin | 3 .. n-2 bert groups (top layers first) | n-1 embeddings
        groups = list(range(3, n - 1))
        order = [0]
        for k, g in enumerate(groups):
            order.append(g)
            if k == 0:
                order.append(2)
            elif k == 1:
                order.append(1)
        for b in (2, 1):
            if b not in order:
                order.append(b)
        return order + [n - 1]

    def tap_dlogits(self):
        return self.tap("dlogits")

    def bucket_views(self):
        """gradient buckets (flat slices of the gradient arena) in backward completion order"""
        self._materialize_zero()
        return [self._grads[b0:b1] for b0, b1 in self._buckets]

    def flat_parameters(self):
        return self._arenas[0][:self._sizes[0]]

    def flat_gradients(self):
        self._materialize_zero()
        return self._grads[:self._sizes[0]]

    def flat_bn_buffers(self):
        return self._arenas[3][:self._sizes[3]]


_PINYIN_TABLES = {}      # id(tokenizer) -> (tokenizer, PinyinTable): the conversion is a pure function of the vocabulary


def pinyin_table_for(tokenizer):
    """The per-vocabulary pinyin table of a tokenizer, built once (Pinyin2.get_pinyin over every vocabulary entry,
    src/utils.py:74-90) and cached for the tokenizer's lifetime."""
    from .pinyin import PinyinTable
    hit = _PINYIN_TABLES.get(id(tokenizer))
    if hit is not None and hit[0] is tokenizer:
        return hit[1]
    n = getattr(tokenizer, "vocab_size", None) or len(tokenizer.vocab)
    table = PinyinTable.build(tokenizer.convert_ids_to_tokens(list(range(int(n)))))
    _PINYIN_TABLES[id(tokenizer)] = (tokenizer, table)
    return table


class SpellBertPho2ResArch3(RealiseModule):
    """src/models.py:652-870."""
    model_type = "arch3"

    @staticmethod
    def build_batch(batch, tokenizer):
        """src/models.py:797-804: adds ``pho_idx`` [B*S, max pinyin length in the batch] (long) and the host list ``pho_lens``.
        The reference runs pypinyin on every character of every batch; here the vocabulary's table is built once per tokenizer
        and a batch is a gather.  (``model.set_pinyin_table(pinyin_table_for(tokenizer))`` moves the gather to the device.)"""
        table = pinyin_table_for(tokenizer)
        src = batch["src_idx"]
        ids = src.detach().cpu().numpy() if torch.is_tensor(src) else np.asarray(src)
        pho_idx, pho_lens = table.convert(ids)
        batch["pho_idx"] = torch.from_numpy(np.ascontiguousarray(pho_idx))
        batch["pho_lens"] = pho_lens
        return batch


class SpellBert(RealiseModule):
    """src/models.py:32-73 (BASELINE config 1: BERT + tied classifier)."""
    model_type = "bert"


MODEL_CLASSES = {          # src/run.py:40-51
    "bert": SpellBert,
    "bert-pho2-res-arch3": SpellBertPho2ResArch3,
}
