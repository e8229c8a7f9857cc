"""The per-frame test-time adaptation of DynaVSR as one reusable function.

This is the body of the hot loop of codes/test_dynavsr.py:197-283 (and of the validation loops
train_dynavsr.py:552-677) expressed over the wrapper API, so that drivers, benchmarks and tests
share one implementation:

    baseline SR  ->  deepcopy(netG, netE)  ->  inner optimizer over G u E params
    for adapt_iter steps:  SLR = netE(LR) (with grad);  loss = cri(netG(SLR), LR[:,center])
                           + 10 * L1(SLR, netE_fixed(LR));  backward;  step
    adapted SR = netG(LR)

One deviation that the wrapper contract allows and SURVEY.md §8f-1 calls out: the frozen
estimator's output does not change inside the step loop, so it is computed once per frame.
"""
from copy import deepcopy

import torch
import torch.nn.functional as F

from . import optim


def make_inner_optimizer(opt, netG, netE):
    m = opt['train']['maml']
    params = [p for p in netG.parameters() if p.requires_grad]
    if not opt['train']['use_real']:
        params += [p for p in netE.parameters() if p.requires_grad]
    # same update rules as the torch.optim objects test_dynavsr.py:223-231 builds; on the GPU the 158
    # parameter tensors are stepped by the native multi-tensor kernels (dynavsr_amd/optim.py)
    on_gpu = all(p.is_cuda for p in params)
    if m['optimizer'] == 'Adam':
        if on_gpu:
            return optim.Adam(params, lr=m['lr_alpha'], betas=(m['beta1'], m['beta2']))
        return torch.optim.Adam(params, lr=m['lr_alpha'], betas=(m['beta1'], m['beta2']))
    if m['optimizer'] == 'SGD':
        if on_gpu:
            return optim.SGD(params, lr=m['lr_alpha'])
        return torch.optim.SGD(params, lr=m['lr_alpha'])
    raise NotImplementedError()


def _fresh_copy(dst, src):
    """dst = deepcopy(src) (test_dynavsr.py:208).  When dst already is a copy from the previous frame (same class,
    same parameter names and shapes, same device) only the values are refreshed: one multi-tensor copy instead of
    re-creating ~150 modules and parameters (2.9 -> 0.3 ms per frame for netG + netE)."""
    if dst is not None and dst is not src and type(dst) is type(src):
        d, s = list(dst.named_parameters()), list(src.named_parameters())
        same = len(d) == len(s) and not list(src.buffers()) and all(
            a[0] == b[0] and a[1].shape == b[1].shape and a[1].device == b[1].device and a[1].dtype == b[1].dtype
            and a[1].requires_grad == b[1].requires_grad for a, b in zip(d, s))
        if same:
            with torch.no_grad():
                torch._foreach_copy_([a[1] for a in d], [b[1] for b in s])
            for _, p in d:
                p.grad = None
            dst.train(src.training)
            return dst
    return deepcopy(src)


def adapt_frame(opt, model, est_model, modelcp, est_modelcp, est_model_fixed, val_data, slr_weight=10.0):
    """Adapt copies of (model.netG, est_model.netE) on one LR clip and super-resolve it.

    val_data: {'LQs': [1,N,3,H,W] (+ 'SuperLQs' when train.use_real)}.  Returns a dict with the
    adapted output ``sr`` [1,3,sH,sW] (on device), the per-step ``losses`` and the updated wrappers
    (modelcp.netG / est_modelcp.netE hold the adapted weights)."""
    lqs = val_data['LQs']
    assert lqs.size(0) == 1
    center = lqs.size(1) // 2
    steps = opt['train']['maml']['adapt_iter']
    prev = (modelcp.netG, est_modelcp.netE)
    modelcp.netG, est_modelcp.netE = _fresh_copy(modelcp.netG, model.netG), _fresh_copy(est_modelcp.netE, est_model.netE)
    # a new inner optimiser per frame, like the reference; when the copies were refreshed in place the previous
    # frame's native optimiser holds the very same parameters and is reset instead of rebuilt
    m = opt['train']['maml']
    sig = (m['optimizer'], m['lr_alpha'], m.get('beta1'), m.get('beta2'), bool(opt['train']['use_real']))
    cached = getattr(modelcp, '_inner_opt', None)
    if cached is not None and cached[0] == sig and prev == (modelcp.netG, est_modelcp.netE) and hasattr(cached[1], 'reset'):
        inner = cached[1]
        inner.reset()
    else:
        inner = make_inner_optimizer(opt, modelcp.netG, est_modelcp.netE)
        modelcp._inner_opt = (sig, inner)
    est_model_fixed.feed_data(val_data)
    est_model_fixed.test()
    slr_fixed = est_model_fixed.fake_L
    target = lqs[:, center]
    losses = []
    for _ in range(steps):
        if not opt['train']['use_real']:
            est_modelcp.feed_data(val_data)
            est_modelcp.forward_without_optim()
            slr = est_modelcp.fake_L
        else:
            slr = val_data['SuperLQs']
        inner.zero_grad()
        modelcp.feed_data({'LQs': slr, 'GT': target})
        loss = modelcp.calculate_loss()
        loss = loss + slr_weight * F.l1_loss(slr.to(slr_fixed.device), slr_fixed)
        loss.backward()
        inner.step()
        losses.append(loss.detach())
    modelcp.feed_data({'LQs': lqs}, need_GT=False)
    modelcp.test()
    return {'sr': modelcp.fake_H, 'losses': losses, 'slr': slr.detach()}
