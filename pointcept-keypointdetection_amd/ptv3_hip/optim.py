"""Fused multi-tensor AdamW on the HIP path (ptv3_adamw_step): one launch per optimizer step.

Drop-in for torch.optim.AdamW as the reference builds it (pointcept/utils/optimizer.py: param groups with
their own lr / weight_decay, e.g. the "block" keyword group of configs/my_dataset/offset_keypoint_ptv3.py);
schedulers that rewrite group["lr"] between steps (OneCycleLR, engines/train.py:207-213) are honoured because
lr / weight_decay travel as launch arguments, not in the device table.
"""
import ctypes

import torch

from .lib import lib
from .ops import _stream


def _shadow_of(p, dtype):
    """(natural, transposed) compute-dtype copies of a >= 2-d weight, created on first use.  Linear (out, in) ->
    (in, out); SubMConv3d (out, k, k, k, in) -> (in, taps mirrored, out).  Kept on the Parameter together with the
    torch version counter they were made at: any torch-side in-place change of the weight invalidates them, the
    fused optimizer step (which writes them itself) does not."""
    rows, cols = p.shape[0], p.shape[-1]
    kvol = p.numel() // (rows * cols)
    with torch.no_grad():
        w = p.detach().reshape(rows, kvol, cols)
        nat = w.reshape(rows, kvol * cols).to(dtype).contiguous()
        tr = w.flip(1).permute(2, 1, 0).reshape(cols, kvol * rows).to(dtype).contiguous()
    return {"version": p._version, "dtype": dtype, "nat": nat, "t": tr, "dims": (rows, cols, kvol)}


def weight_shadow(p, dtype):
    """The valid shadow of `p` in `dtype`, or None (no fused optimizer attached / stale)."""
    sh = getattr(p, "_ptv3_shadow", None)
    if sh is None or sh["dtype"] != dtype or sh["version"] != p._version:
        return None
    return sh


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, shadow_dtype=None):
        """shadow_dtype (torch.bfloat16 / torch.float32 / None): keep compute-dtype and transposed copies of every
        weight matrix up to date inside the step kernel (ptv3_adamw_fill_shadow); the training Functions of
        ptv3_hip.autograd pick them up instead of casting / transposing the fp32 masters every step."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > 8:
            raise ValueError("FusedAdamW: at most 8 parameter groups")
        self.shadow_dtype = shadow_dtype
        self._key = None
        self._table = None
        self._nt = self._nb = 0
        self._partial = None
        # Step accounting compatible with torch.optim.AdamW checkpoints: state[p]["step"] (CPU float tensor, as torch
        # keeps it) is authoritative while p is NOT in the device table; for the tensors in the table the count is
        # self._steps - lag (one launch argument + a per-entry constant instead of 495 host-side increments per step)
        # and is written back whenever the table is rebuilt or a state_dict is taken.
        self._steps = 0
        self._active = []   # (param, lag) of the current device table
        self._fast = None   # (parameters of the table, their addresses) for the per-step fast path of _build

    def _entries(self):
        out = []
        for gi, group in enumerate(self.param_groups):
            if group["betas"] != self.param_groups[0]["betas"] or group["eps"] != self.param_groups[0]["eps"]:
                raise ValueError("FusedAdamW: betas / eps must be shared by all groups")
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda:
                    raise TypeError("FusedAdamW: fp32 GPU parameters and gradients only (masters stay fp32)")
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedAdamW: parameters and gradients must be contiguous")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                out.append((p, st, gi))
        return out

    def _shadow(self, p):
        if self.shadow_dtype is None or p.dim() < 2 or p.shape[-1] % 8 or p.shape[0] % 8:
            return None
        sh = weight_shadow(p, self.shadow_dtype)
        if sh is None:
            sh = p._ptv3_shadow = _shadow_of(p, self.shadow_dtype)
        return sh

    def _build(self):
        # fast path (every step but the first): the same parameters, in the same places, all with a gradient - only the
        # gradient addresses are new.  The full walk with its dtype / layout checks runs when anything else changed.
        fast = self._fast
        if fast is not None and self._key is not None:
            params, ptrs = fast
            try:
                grads = [p.grad.data_ptr() for p in params]
            except AttributeError:      # a parameter lost its gradient: rebuild
                grads = None
            if grads is not None and [p.data_ptr() for p in params] == ptrs and \
                    sum(1 for g in self.param_groups for p in g["params"] if p.grad is not None) == len(params):
                self._grad_ptrs = (ctypes.c_void_p * max(1, len(grads)))(*grads)
                return
        ent = self._entries()
        shadows = [self._shadow(p) for p, _, _ in ent]
        # gradient addresses are NOT part of the key: they change every step after zero_grad(set_to_none=True) and
        # travel to the kernel as launch arguments (self._grad_ptrs)
        key = tuple((p.data_ptr(), gi, 0 if sh is None else sh["nat"].data_ptr())
                    for (p, _, gi), sh in zip(ent, shadows))
        self._grad_ptrs = (ctypes.c_void_p * max(1, len(ent)))(*[p.grad.data_ptr() for p, _, _ in ent])
        if key == self._key:
            return
        self._sync_steps()   # tensors leaving the table keep their own count in state["step"]
        esz, chunk = lib.ptv3_adamw_entry_bytes(), lib.ptv3_adamw_chunk()
        # the launch's step argument is the largest per-tensor count + 1; everybody else lags behind it
        self._steps = max([int(st["step"].item()) if "step" in st else 0 for _, st, _ in ent] or [0])
        active = []
        host = ctypes.create_string_buffer(max(1, esz * len(ent)))
        base = ctypes.addressof(host)
        blocks = tiles = 0
        first = []
        for i, (p, st, gi) in enumerate(ent):
            first.append(blocks)
            lib.check(lib.ptv3_adamw_fill_entry(base + i * esz, p.data_ptr(), p.grad.data_ptr(),
                                                st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), gi,
                                                blocks), "ptv3_adamw_fill_entry")
            own = int(st["step"].item()) if "step" in st else 0
            active.append((p, self._steps - own))
            if self._steps != own:
                lib.check(lib.ptv3_adamw_fill_step_lag(base + i * esz, self._steps - own), "ptv3_adamw_fill_step_lag")
            sh = shadows[i]
            lib.check(lib.ptv3_adamw_fill_first_tile(base + i * esz, tiles), "ptv3_adamw_fill_first_tile")
            if sh is not None:
                rows, cols, kvol = sh["dims"]
                lib.check(lib.ptv3_adamw_fill_shadow(base + i * esz, sh["nat"].data_ptr(), sh["t"].data_ptr(), rows,
                                                     cols, kvol, 0 if self.shadow_dtype == torch.float32 else 1),
                          "ptv3_adamw_fill_shadow")
                tiles += lib.ptv3_adamw_shadow_tiles(rows, cols, kvol)
            blocks += (p.numel() + chunk - 1) // chunk
        dev = ent[0][0].device if ent else torch.device("cuda")
        self._table = torch.frombuffer(host, dtype=torch.uint8).clone().to(dev)
        self._partial = torch.empty(max(blocks, 1), dtype=torch.float32, device=dev)
        self._first_blocks = (ctypes.c_int32 * max(1, len(first)))(*first)
        self._tiles = tiles
        self._fast = ([p for p, _, _ in ent], [p.data_ptr() for p, _, _ in ent])
        self._nt, self._nb, self._key, self._active = len(ent), blocks, key, active

    def _sync_steps(self):
        """write the step counts of the tensors in the device table back into state[p]["step"]"""
        for p, lag in self._active:
            st = self.state[p]
            st["step"] = torch.tensor(float(self._steps - lag), dtype=torch.float32)

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Accepts checkpoints of this class and of torch.optim.AdamW (exp_avg / exp_avg_sq / per-parameter step)."""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = st["step"].detach().to("cpu", torch.float32)
            elif "step" in st:
                st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
        self._key, self._active, self._fast = None, [], None   # moments are new tensors: rebuild the device table

    @torch.no_grad()
    def zero_grad(self, set_to_none=True):
        """Default drops the gradient tensors (torch's own default): the next backward then hands every parameter a
        fresh tensor - no zero fill and no accumulate-add launch per parameter (~450 + ~130 launches of the fork
        model's step) - and step() passes the new addresses to the kernel as launch arguments.  set_to_none=False keeps
        the tensors and zeroes them with multi-tensor fills."""
        if set_to_none:
            return super().zero_grad(set_to_none=True)
        grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
        if grads:
            torch._foreach_zero_(grads)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._build()
        self._steps += 1
        ng = len(self.param_groups)
        lr = (ctypes.c_float * ng)(*[float(g["lr"]) for g in self.param_groups])
        wd = (ctypes.c_float * ng)(*[float(g["weight_decay"]) for g in self.param_groups])
        b1, b2 = self.param_groups[0]["betas"]
        lib.check(lib.ptv3_adamw_step(self._table.data_ptr(), self._nt, self._nb, lr, wd, ng, float(b1), float(b2),
                                      float(self.param_groups[0]["eps"]), self._steps, float(grad_scale),
                                      self._first_blocks, self._grad_ptrs, self._tiles,
                                      0 if self.shadow_dtype in (None, torch.float32) else 1, _stream()),
                  "ptv3_adamw_step")
        # The kernel wrote the parameters through raw pointers: tell torch.  Every eval-side cache (folded BatchNorm,
        # cast / permuted weights, the executor's packed table) is keyed on (data_ptr, _version); without the bump an
        # evaluation between two training epochs would keep running on the weights of the first one.
        updated = [p for p, _ in self._active]
        torch._C._increment_version(updated)
        for p in updated:   # the shadows were written by the kernel itself: still exact at the new version
            sh = getattr(p, "_ptv3_shadow", None)
            if sh is not None:
                sh["version"] = p._version
        return loss

    @torch.no_grad()
    def grad_norm(self):
        """L2 norm of all gradients (0-d device tensor): the quantity clip_grad_norm_ computes."""
        self._build()
        out = torch.empty(1, dtype=torch.float32, device=self._table.device)
        lib.check(lib.ptv3_grad_sqnorm(self._table.data_ptr(), self._nt, self._nb, self._partial.data_ptr(),
                                       out.data_ptr(), self._first_blocks, self._grad_ptrs, _stream()),
                  "ptv3_grad_sqnorm")
        return out.sqrt_()[0]
