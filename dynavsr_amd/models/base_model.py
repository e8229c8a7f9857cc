"""Wrapper base class: device choice, LR helpers, checkpoint save/load/resume.

Same behaviour as codes/models/base_model.py:8-121 (method names, file naming
``{iter}_{label}.pth`` / ``{iter}_{type}.state``, 'module.' prefix stripped on load) so that the
DynaVSR drivers and checkpoints work unchanged.  Networks are held bare (no DataParallel shell):
with one process per GPU there is nothing for it to do, and state-dict keys are written without
the 'module.' prefix exactly like the reference writes them (base_model.py:77-79).
"""
import os
from collections import OrderedDict

import torch


def unwrap(network):
    return network.module if hasattr(network, 'module') and isinstance(network.module, torch.nn.Module) \
        else network


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device('cuda' if opt['gpu_ids'] is not None else 'cpu')
        self.is_train = opt['is_train']
        self.schedulers = []
        self.optimizers = []

    # -- hooks the concrete wrappers fill in
    def feed_data(self, data):
        pass

    def optimize_parameters(self):
        pass

    def get_current_visuals(self):
        pass

    def get_current_losses(self):
        pass

    def print_network(self):
        pass

    def save(self, label):
        pass

    def load(self):
        pass

    # -- learning-rate helpers (base_model.py:37-66)
    def _set_lr(self, lr_groups_l):
        for optimizer, lr_groups in zip(self.optimizers, lr_groups_l):
            for group, lr in zip(optimizer.param_groups, lr_groups):
                group['lr'] = lr

    def _get_init_lr(self):
        return [[g['initial_lr'] for g in o.param_groups] for o in self.optimizers]

    def update_learning_rate(self, cur_iter, warmup_iter=-1):
        for s in self.schedulers:
            s.step()
        if cur_iter < warmup_iter:
            self._set_lr([[v / warmup_iter * cur_iter for v in grp] for grp in self._get_init_lr()])

    def get_current_learning_rate(self):
        return [g['lr'] for g in self.optimizers[0].param_groups]

    def get_network_description(self, network):
        network = unwrap(network)
        return str(network), sum(p.numel() for p in network.parameters())

    # -- checkpoints (base_model.py:74-121)
    def save_network(self, network, network_label, iter_label):
        path = os.path.join(self.opt['path']['models'], '{}_{}.pth'.format(iter_label, network_label))
        torch.save(OrderedDict((k, v.cpu()) for k, v in unwrap(network).state_dict().items()), path)

    def load_network(self, load_path, network, strict=True):
        loaded = torch.load(load_path, map_location='cpu')
        clean = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in loaded.items())
        unwrap(network).load_state_dict(clean, strict=strict)

    def save_training_state(self, epoch, iter_step, model_type=None):
        state = {'epoch': epoch, 'iter': iter_step,
                 'schedulers': [s.state_dict() for s in self.schedulers],
                 'optimizers': [o.state_dict() for o in self.optimizers]}
        name = '{}_{}.state'.format(iter_step, model_type) if model_type is not None \
            else '{}.state'.format(iter_step)
        torch.save(state, os.path.join(self.opt['path']['training_state'], name))

    def resume_training(self, resume_state):
        opts, scheds = resume_state['optimizers'], resume_state['schedulers']
        assert len(opts) == len(self.optimizers), 'Wrong lengths of optimizers'
        assert len(scheds) == len(self.schedulers), 'Wrong lengths of schedulers'
        for mine, saved in zip(self.optimizers, opts):
            mine.load_state_dict(saved)
        for mine, saved in zip(self.schedulers, scheds):
            mine.load_state_dict(saved)
