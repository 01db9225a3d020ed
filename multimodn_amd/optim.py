"""Adam on the HIP path: `optimizer.step()` of the reference's batch loop (multimodn.py:204) as ONE
launch of `k_adam` over the flat parameter / gradient buffers the engine keeps.

Drop-in for `torch.optim.Adam(params, lr, betas, eps, weight_decay)` as every reference pipeline
constructs it (pipelines/titanic/titanic_mlp_pipeline.py:74): same constructor arguments, same
`state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), same treatment of parameters
whose grad is None (left untouched).  The per-parameter state tensors are views into flat buffers.

There is no torch-op fallback: parameters off the GPU, or a missing libmmn_hip.so, raise.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch
from torch.optim import Optimizer

from . import hip


class _Run:
    """Maximal stretch of parameters that are adjacent in memory, with adjacent gradients."""

    def __init__(self, params: List[torch.nn.Parameter], grads: List[torch.Tensor], device: torch.device, lib):
        self.params = params
        self.p_ptr = params[0].data_ptr()
        self.g_ptr = grads[0].data_ptr()
        sizes = [p.numel() for p in params]
        self.n = sum(sizes)
        starts = [0]
        for k in sizes:
            starts.append(starts[-1] + k)
        self.starts = starts
        self.exp_avg = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(self.n, dtype=torch.float32, device=device)
        # one row of step counters per workgroup of k_adam (all rows equal; row 0 is state["step"])
        blocks = int(lib.mmn_adam_blocks(self.n))
        if blocks < 1:
            raise hip.MmnError(f"{self.n} parameters in one run: outside mmn_adam_step's range")
        self.steps = torch.zeros(blocks, len(params), dtype=torch.float32, device=device)
        self.seg_start = torch.tensor(starts, dtype=torch.int32, device=device)
        self.seg_skip = torch.zeros(len(params), dtype=torch.int32, device=device)
        self.skip_host: Tuple[int, ...] = tuple([0] * len(params))
        self.sig = tuple(p.data_ptr() for p in params)
        self.gsig = tuple(g.data_ptr() for g in grads)


class Adam(Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, amsgrad: bool = False, *, maximize: bool = False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not on the HIP path (no reference pipeline uses it)")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=maximize))
        self._lib = None
        self._runs: List[Optional[List[_Run]]] = [None] * len(self.param_groups)

    # ------------------------------------------------------------------ layout
    def _build_runs(self, gi: int, group) -> List[_Run]:
        """Group the parameters into memory-adjacent runs and move any existing per-parameter
        state (earlier steps, load_state_dict) into the runs' flat buffers."""
        params = [p for p in group["params"]]
        for p in params:
            if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                raise hip.MmnError("multimodn_amd.optim.Adam only runs on contiguous float32 parameters on an AMD "
                                   f"GPU (got {p.dtype} on {p.device}); there is no CPU fallback")
        last_g = getattr(self, "_last_grads", {})
        runs: List[_Run] = []
        cur_p: List[torch.nn.Parameter] = []
        cur_g: List[torch.Tensor] = []

        def flush():
            if cur_p:
                runs.append(_Run(list(cur_p), list(cur_g), cur_p[0].device, self._lib))
            cur_p.clear()
            cur_g.clear()

        for p in params:
            g = p.grad if p.grad is not None else last_g.get(id(p))
            if g is None:
                raise RuntimeError("first optimizer.step(): every parameter needs a gradient buffer "
                                   "(train_epoch assigns them)")
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device:
                raise hip.MmnError("gradients must be contiguous float32 tensors on the parameter's device")
            if cur_p:
                adjacent = (cur_p[-1].data_ptr() + 4 * cur_p[-1].numel() == p.data_ptr()
                            and cur_g[-1].data_ptr() + 4 * cur_g[-1].numel() == g.data_ptr()
                            and len(cur_p) < hip.ADAM_MAX_SEG)
                if not adjacent:
                    flush()
            if not cur_p and (p.data_ptr() % 16 or g.data_ptr() % 16):
                raise hip.MmnError("parameter / gradient storage must be 16-byte aligned")
            cur_p.append(p)
            cur_g.append(g)
        flush()
        with torch.no_grad():
            for r in runs:
                for i, p in enumerate(r.params):
                    lo, hi = r.starts[i], r.starts[i + 1]
                    st = self.state.get(p, {})
                    if "exp_avg" in st:
                        r.exp_avg[lo:hi].copy_(st["exp_avg"].reshape(-1))
                        r.exp_avg_sq[lo:hi].copy_(st["exp_avg_sq"].reshape(-1))
                        r.steps[:, i] = float(st["step"])
                    self.state[p] = {"step": r.steps[0, i], "exp_avg": r.exp_avg[lo:hi].view(p.shape),
                                     "exp_avg_sq": r.exp_avg_sq[lo:hi].view(p.shape)}
        return runs

    def _runs_valid(self, runs: List[_Run]) -> bool:
        for r in runs:
            if tuple(p.data_ptr() for p in r.params) != r.sig:
                return False
            for p, gp in zip(r.params, r.gsig):
                if p.grad is not None and p.grad.data_ptr() != gp:
                    return False
        return True

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._runs = [None] * len(self.param_groups)       # state tensors were replaced: re-flatten lazily

    def add_param_group(self, param_group) -> None:
        super().add_param_group(param_group)
        if hasattr(self, "_runs"):
            self._runs.append(None)

    @staticmethod
    def descriptor(r: _Run, group) -> "hip.AdamDesc":
        """The mmn_adam struct of one run (cached; hyper-parameters refreshed every step so LR
        schedulers work)."""
        d = getattr(r, "desc", None)
        if d is None:
            d = r.desc = hip.AdamDesc()
            d.params, d.grads = r.p_ptr, r.g_ptr
            d.exp_avg, d.exp_avg_sq = r.exp_avg.data_ptr(), r.exp_avg_sq.data_ptr()
            d.steps, d.seg_start = r.steps.data_ptr(), r.seg_start.data_ptr()
            d.n, d.n_seg = r.n, len(r.params)
        d.seg_skip = r.seg_skip.data_ptr() if any(r.skip_host) else None
        b1, b2 = group["betas"]
        d.lr, d.beta1, d.beta2 = float(group["lr"]), float(b1), float(b2)
        d.eps, d.weight_decay = float(group["eps"]), float(group["weight_decay"])
        d.maximize = 1 if group["maximize"] else 0
        return d

    # ------------------------------------------------------------------ fusion with the engine's last launch
    def fused_descriptor(self, engine) -> Optional["hip.AdamDesc"]:
        """The mmn_adam struct for engine.local_step(..., optimizer=self), or None when this
        optimizer cannot be fused with that engine (several groups / runs, other parameters,
        a refusal by the library earlier).  The library re-checks the layout itself."""
        if len(self.param_groups) != 1 or getattr(self, "_no_fuse_sig", None) == engine._sig:
            return None
        group = self.param_groups[0]
        params = group["params"]
        if len(params) != len(engine.params) or any(a is not b for a, b in zip(params, engine.params)):
            return None
        if self._lib is None:
            self._lib = hip.load()
        if not hasattr(self, "_last_grads"):
            self._last_grads = {}
        runs = self._runs[0]
        g_ptr = engine.flat_grads.data_ptr()
        # (the engine re-reads every parameter's address once per epoch - its _sig - and its gradient views never move
        #  while the plan lives: comparing against those spares this call two walks over all parameters)
        if runs is not None and len(runs) == 1 and runs[0].sig == engine._sig and runs[0].g_ptr == g_ptr \
                and getattr(self, "_lg_engine", None) == (id(engine), g_ptr) and not any(runs[0].skip_host):
            return self.descriptor(runs[0], group)
        for p, g in zip(engine.params, engine.grad_views):     # the gradients WILL live there
            self._last_grads[id(p)] = g
        self._lg_engine = (id(engine), g_ptr)
        if runs is None or not self._runs_valid(runs) or runs[0].g_ptr != g_ptr:
            saved = [p.grad for p in params]
            for p, g in zip(params, engine.grad_views):
                p.grad = g
            try:
                runs = self._runs[0] = self._build_runs(0, group)
            finally:
                for p, g in zip(params, saved):
                    p.grad = g
        if len(runs) != 1:
            return None
        r = runs[0]
        if any(r.skip_host):                                  # the kernel derives skips from the executed rows
            r.seg_skip.zero_()
            r.skip_host = tuple([0] * len(r.params))
        return self.descriptor(r, group)

    def mark_fused_step(self) -> None:
        self._fused_pending = True

    def fused_step_seen(self, n: int = 1) -> None:
        """What `n` calls of step() do after `n` fused steps.  Without step hooks that is bookkeeping only (the flag torch's
        LR schedulers look for included: they warn "lr_scheduler.step() before optimizer.step()" otherwise); with hooks
        registered (register_step_pre_hook / _post_hook, torch's global optimizer hooks) step() itself is called - it
        returns at once after a fused step - so that they fire once per step as they would in the reference's loop."""
        import torch.optim.optimizer as _to
        hooked = bool(getattr(self, "_optimizer_step_pre_hooks", None) or getattr(self, "_optimizer_step_post_hooks", None)
                      or getattr(_to, "_global_optimizer_pre_hooks", None) or getattr(_to, "_global_optimizer_post_hooks", None))
        if hooked:
            for _ in range(max(int(n), 1)):
                self._fused_pending = True
                self.step()
            return
        self._fused_pending = False
        self._opt_called = True

    def fusion_refused(self, engine) -> None:
        self._no_fuse_sig = engine._sig

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if getattr(self, "_fused_pending", False):           # engine.local_step(..., optimizer=self) already stepped
            self._fused_pending = False
            return loss
        if self._lib is None:
            self._lib = hip.load()
        if not hasattr(self, "_last_grads"):
            self._last_grads = {}
        for gi, group in enumerate(self.param_groups):
            if not group["params"]:
                continue
            if all(p.grad is None for p in group["params"]) and self._runs[gi] is None:
                continue
            runs = self._runs[gi]
            if runs is None or not self._runs_valid(runs):
                runs = self._runs[gi] = self._build_runs(gi, group)
            for p in group["params"]:
                if p.grad is not None:
                    self._last_grads[id(p)] = p.grad
            stream = torch.cuda.current_stream().cuda_stream
            for r in runs:
                skip = tuple(1 if p.grad is None else 0 for p in r.params)
                if all(skip):
                    continue
                if skip != r.skip_host:
                    r.seg_skip.copy_(torch.tensor(skip, dtype=torch.int32))
                    r.skip_host = skip
                d = self.descriptor(r, group)
                hip.check(self._lib.mmn_adam_step(C.byref(d), stream), "mmn_adam_step")
                hip.PARAM_WRITES[0] += 1                      # (raw-pointer update: torch's version counters do not see it)
        return loss
