"""Fused AdamW + linear warm-up schedule over the module's flat arenas.

Mirrors ``transformers/optimization.py:87-169`` (decoupled weight decay applied after the Adam
update, optional bias correction) and ``clip_grad_norm_`` (src/run.py:207) as two kernels per
step instead of ~4000 tiny ones: sum-of-squares of the gradient arena, then one AdamW sweep that
applies the clip coefficient on the fly (no host synchronisation anywhere).
"""
import ctypes as C

import torch

from . import _capi


class FusedAdamW(torch.optim.Optimizer):
    """Drop-in for the reference's ``AdamW(optimizer_grouped_parameters, lr=, eps=)`` (run.py:146-153)
    when the parameters belong to ONE RealiseModule.  Parameter groups keep their own lr /
    weight_decay (the trainer's decay / no-decay split); groups whose hyper-parameters agree are
    stepped with a single launch over the whole arena."""

    def __init__(self, module, params=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True,
                 max_grad_norm=None):
        self.module = getattr(module, "module", module)
        if params is None:
            params = [p for p in self.module.parameters() if p.requires_grad]
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        super().__init__(params, defaults)
        self.max_grad_norm = max_grad_norm
        flat = self.module.flat_parameters()
        self._m = torch.zeros_like(flat)
        self._v = torch.zeros_like(flat)
        self._norm_sq = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self._step = 0
        self._ranges = None
        # True: step through realise_engine_adamw once the module owns an engine (after its first forward): AdamW + the Linear weights'
        # operand copies in one pass.  False: the arena-level kernels, and the module re-derives every copy at its next forward.
        self.fused_operand_copies = True

    def _param_ranges(self):
        """(offset, numel) of every trainable parameter of each group inside the arena"""
        if self._ranges is None:
            by_id = {}
            for name, (arena, off, shape, p) in self.module._views.items():
                if arena == 0 and p is not None:
                    by_id[id(p)] = (off, p.numel())
            self._ranges = [[by_id[id(p)] for p in g["params"] if id(p) in by_id] for g in self.param_groups]
        return self._ranges

    def _group_map(self, flat_p):
        """uint8 per 64 arena elements: index of the parameter group that owns them, 255 = none (built once)"""
        if getattr(self, "_gmap", None) is None:
            import numpy as np
            n = flat_p.numel()
            gmap = np.full((n + 63) // 64, 255, np.uint8)
            for k, rng in enumerate(self._param_ranges()):
                for off, cnt in rng:
                    if off % 64:
                        raise RuntimeError("arena tensor not on a 64-element boundary")
                    gmap[off // 64:(off + cnt + 63) // 64] = k
            self._gmap = torch.from_numpy(gmap).to(flat_p.device)
        return self._gmap

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._ranges = None
        self._gmap = None

    @torch.no_grad()
    def step(self, closure=None):
        lib = _capi.load()
        mod = self.module
        st = C.c_void_p(torch.cuda.current_stream(mod.device).cuda_stream)
        self._step += 1
        flat_p, flat_g = mod.flat_parameters(), mod.flat_gradients()
        norm_ptr, max_norm = None, 0.0
        if self.max_grad_norm is not None:
            self._norm_sq.zero_()
            _capi.check(lib.realise_sumsq(st, flat_g.data_ptr(), flat_g.numel(), self._norm_sq.data_ptr()), "realise_sumsq")
            norm_ptr, max_norm = self._norm_sq.data_ptr(), float(self.max_grad_norm)
        g0 = self.param_groups[0]
        uniform = all(g["lr"] == g0["lr"] and g["weight_decay"] == g0["weight_decay"] and g["eps"] == g0["eps"]
                      and g["betas"] == g0["betas"] and g["correct_bias"] == g0["correct_bias"] for g in self.param_groups)
        # the whole-arena launch is only valid when the groups hold EVERY trainable tensor of the arena exactly once
        covered = sorted(r for rng in self._param_ranges() for r in rng)
        wanted = sorted({(o, p.numel()) for name, (a, o, s, p) in mod._views.items() if a == 0 and p is not None})
        fused = self.fused_operand_copies and len(self.param_groups) <= 8 and getattr(mod, "_engine", None) is not None \
            and getattr(mod, "_shadow", None) is not None and mod.trust_fused_optimizer
        if fused:
            # the engine's form of the grouped sweep: the Linear weights are stepped in the tiles of the operand-copy kernel, which stores
            # the new fp32 value and its bf16 W / W^T copies in one pass - the next forward skips that part of the refresh
            groups = (_capi.AdamwGroup * len(self.param_groups))()
            for k, g in enumerate(self.param_groups):
                groups[k] = _capi.AdamwGroup(g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], 1 if g["correct_bias"] else 0)
            # (pipeline_optimizer: the same sweep on the engine's side stream, the next forward waiting for it piece by piece)
            sweep = lib.realise_engine_adamw_pipelined if getattr(mod, "pipeline_optimizer", False) else lib.realise_engine_adamw
            _capi.check(sweep(mod._engine, st, self._m.data_ptr(), self._v.data_ptr(), self._group_map(flat_p).data_ptr(), groups,
                              len(self.param_groups), self._step, norm_ptr, max_norm), "realise_engine_adamw")
            mod.mark_parameters_updated(frozen=False, linear_copies_current=True)
            return
        if uniform and covered == wanted:
            _capi.check(lib.realise_adamw(st, flat_p.data_ptr(), flat_g.data_ptr(), self._m.data_ptr(), self._v.data_ptr(),
                                          flat_p.numel(), g0["lr"], g0["betas"][0], g0["betas"][1], g0["eps"], g0["weight_decay"],
                                          self._step, 1 if g0["correct_bias"] else 0, norm_ptr, max_norm), "realise_adamw")
        elif len(self.param_groups) <= 8:
            # groups with their own hyper-parameters (a real --weight_decay on the decay group, run.py:146-151): still ONE launch -
            # a byte per 64 parameters names the group (every tensor of the arena starts on a 64-element boundary)
            groups = (_capi.AdamwGroup * len(self.param_groups))()
            for k, g in enumerate(self.param_groups):
                groups[k] = _capi.AdamwGroup(g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], 1 if g["correct_bias"] else 0)
            _capi.check(lib.realise_adamw_grouped(st, flat_p.data_ptr(), flat_g.data_ptr(), self._m.data_ptr(), self._v.data_ptr(),
                                                  flat_p.numel(), self._group_map(flat_p).data_ptr(), groups, len(self.param_groups),
                                                  self._step, norm_ptr, max_norm), "realise_adamw_grouped")
        else:
            for g, rng in zip(self.param_groups, self._param_ranges()):
                for off, n in rng:
                    _capi.check(lib.realise_adamw(st, flat_p.data_ptr() + 4 * off, flat_g.data_ptr() + 4 * off,
                                                  self._m.data_ptr() + 4 * off, self._v.data_ptr() + 4 * off, n, g["lr"],
                                                  g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self._step,
                                                  1 if g["correct_bias"] else 0, norm_ptr, max_norm), "realise_adamw")
        mod.mark_parameters_updated(frozen=False)

    def state_dict(self):
        """torch's param_groups plus the flat first / second moments and the step count (they live outside ``self.state``)"""
        self.module.sync_optimizer()
        sd = super().state_dict()
        sd["realise_flat"] = {"m": self._m.detach().clone(), "v": self._v.detach().clone(), "step": int(self._step)}
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        flat = sd.pop("realise_flat", None)
        self.module.sync_optimizer()
        super().load_state_dict(sd)
        if flat is not None:
            if flat["m"].numel() != self._m.numel():
                raise ValueError("optimizer state belongs to a different model layout")
            self._m.copy_(flat["m"].to(self._m.device))
            self._v.copy_(flat["v"].to(self._v.device))
            self._step = int(flat["step"])

    def grad_norm(self):
        """global L2 norm measured by the last step() (device tensor; reading it synchronises)"""
        return self._norm_sq.sqrt()

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad: the default detaches (the module's lazy fresh-gradient pass), ``set_to_none=False``
        zero-fills the arena now and keeps every ``p.grad`` view attached"""
        self.module.zero_grad(set_to_none=set_to_none)


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    """transformers/optimization.py:45-54"""
    def lr_lambda(current_step):
        if current_step < num_warmup_steps:
            return float(current_step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - current_step) / float(max(1, num_training_steps - num_warmup_steps)))
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda, last_epoch)
