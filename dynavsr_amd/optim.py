"""Inner-loop optimisers on the native multi-tensor kernels (csrc/optim.hip: dvsr_adam_step / dvsr_sgd_step).

Drop-in for the two optimisers test_dynavsr.py:223-231 / train_dynavsr.py:313-321 build for the inner loop
(torch.optim.Adam(params, lr, betas) / torch.optim.SGD(params, lr)): same update rule and the same
zero_grad() / step() / param_groups / state_dict() surface the drivers touch.  The framework optimiser spends
most of its 0.6-1.3 ms per step grouping 158 tensors on the host; here a step is two ctypes calls.
"""
import ctypes

import torch

from . import _lib as L


class _Native(torch.optim.Optimizer):
    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._ps = [p for g in self.param_groups for p in g['params']]
        for p in self._ps:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise RuntimeError("dynavsr_amd.optim works on contiguous fp32 parameters on the MI355X; there is "
                                   "no CPU fallback (got %s on %s)" % (p.dtype, p.device))
        n = len(self._ps)
        self._n = n
        self._parr = (ctypes.c_void_p * n)(*[p.data_ptr() for p in self._ps])
        self._numel = (ctypes.c_longlong * n)(*[p.numel() for p in self._ps])
        self._garr = (ctypes.c_void_p * n)()

    def _grads(self):
        for i, p in enumerate(self._ps):
            g = p.grad
            if g is None:
                self._garr[i] = None
                continue
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = p.grad = g.float().contiguous()
            self._garr[i] = g.data_ptr()
        for i, p in enumerate(self._ps):  # parameters may have been re-pointed (param.data = ...)
            self._parr[i] = p.data_ptr()


class Adam(_Native):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        flat = torch.zeros(2 * sum(p.numel() for p in self._ps), dtype=torch.float32, device=self._ps[0].device)
        off, m, v = 0, [], []
        for p in self._ps:
            k = p.numel()
            m.append(flat[off:off + k]); v.append(flat[off + k:off + 2 * k]); off += 2 * k
            self.state[p] = {'step': 0, 'exp_avg': m[-1].view_as(p), 'exp_avg_sq': v[-1].view_as(p)}
        self._flat = flat
        self._marr = (ctypes.c_void_p * self._n)(*[t.data_ptr() for t in m])
        self._varr = (ctypes.c_void_p * self._n)(*[t.data_ptr() for t in v])
        self._step = 0

    def reset(self):
        """Back to the state of a freshly constructed optimiser (moments zero, step 0): the per-frame adaptation
        builds a new Adam for every frame (test_dynavsr.py:223-231); this is the same thing without 316 views."""
        self._flat.zero_()
        self._step = 0
        for p in self._ps:
            self.state[p]['step'] = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._grads()
        self._step += 1
        g = self.param_groups[0]
        if len(self.param_groups) != 1:
            raise RuntimeError("dynavsr_amd.optim.Adam: one parameter group (the DynaVSR inner loop has one)")
        L.check(L.lib().dvsr_adam_step(self._parr, self._garr, self._marr, self._varr, self._numel, self._n,
                                       g['lr'], g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'],
                                       self._step, L.stream()), "dvsr_adam_step")
        for p in self._ps:
            self.state[p]['step'] = self._step
        return loss


class SGD(_Native):
    def __init__(self, params, lr=1e-3, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay))

    def reset(self):
        pass  # stateless

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._grads()
        g = self.param_groups[0]
        L.check(L.lib().dvsr_sgd_step(self._parr, self._garr, self._numel, self._n, g['lr'], g['weight_decay'],
                                      L.stream()), "dvsr_sgd_step")
        return loss
