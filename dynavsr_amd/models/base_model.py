"""What every model wrapper shares: the device, the optimiser / scheduler lists with their learning-rate helpers,
and checkpoint I/O.

Behaviour follows codes/models/base_model.py:8-121 where the DynaVSR drivers can observe it: the method names,
the checkpoint file names ``<iter>_<label>.pth`` / ``<iter>[_<type>].state``, tensors saved on the CPU, a leading
'module.' stripped from every key on load, linear learning-rate warm-up on top of the schedulers.  Networks are
held bare -- with one process per GPU a DataParallel shell has nothing to do -- and ``unwrap`` keeps code that
receives a wrapped network working.
"""
import os
from collections import OrderedDict

import torch


def unwrap(network):
    """The module itself, whether or not something DataParallel-like wraps it."""
    inner = getattr(network, 'module', None)
    return inner if isinstance(inner, torch.nn.Module) else network


def _state_file(opt, iter_step, model_type):
    stem = str(iter_step) if model_type is None else '{}_{}'.format(iter_step, model_type)
    return os.path.join(opt['path']['training_state'], stem + '.state')


class LogDict(OrderedDict):
    """``log_dict`` of the wrappers.  The reference stores ``loss.item()`` (Video_base_model.py:194): one host
    synchronisation per loss evaluation, in the middle of the inner step -- the GPU then idles while the host
    enqueues the backward (0.4-0.6 ms of a 9 ms step, and it keeps concurrent adaptations from overlapping).
    Here the device scalar is stored and turned into the float the drivers expect when it is READ, so the value
    a driver sees is the same and the synchronisation happens only where somebody looks."""

    @staticmethod
    def _value(v):
        return v.item() if isinstance(v, torch.Tensor) else v

    def __getitem__(self, key):
        v = OrderedDict.__getitem__(self, key)
        if isinstance(v, torch.Tensor):
            v = v.item()
            OrderedDict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def __repr__(self):
        return 'LogDict(%r)' % (self.items(),)


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.is_train = opt['is_train']
        self.device = torch.device('cpu' if opt['gpu_ids'] is None else 'cuda')
        self.optimizers, self.schedulers = [], []

    # ---- checkpoints -------------------------------------------------------------------------------------------
    def save_network(self, network, network_label, iter_label):
        weights = OrderedDict((name, t.cpu()) for name, t in unwrap(network).state_dict().items())
        torch.save(weights, os.path.join(self.opt['path']['models'], '{}_{}.pth'.format(iter_label, network_label)))

    def load_network(self, load_path, network, strict=True):
        prefix = 'module.'
        weights = OrderedDict((name[len(prefix):] if name.startswith(prefix) else name, t)
                              for name, t in torch.load(load_path, map_location='cpu').items())
        unwrap(network).load_state_dict(weights, strict=strict)

    def save_training_state(self, epoch, iter_step, model_type=None):
        torch.save({'epoch': epoch, 'iter': iter_step,
                    'optimizers': [o.state_dict() for o in self.optimizers],
                    'schedulers': [s.state_dict() for s in self.schedulers]},
                   _state_file(self.opt, iter_step, model_type))

    def resume_training(self, resume_state):
        for kind, mine in (('optimizers', self.optimizers), ('schedulers', self.schedulers)):
            saved = resume_state[kind]
            assert len(saved) == len(mine), 'Wrong lengths of ' + kind
            for obj, state in zip(mine, saved):
                obj.load_state_dict(state)

    # ---- learning rate -----------------------------------------------------------------------------------------
    def get_current_learning_rate(self):
        return [group['lr'] for group in self.optimizers[0].param_groups]

    def _get_init_lr(self):
        return [[group['initial_lr'] for group in o.param_groups] for o in self.optimizers]

    def _set_lr(self, lr_groups_l):
        for optimizer, lrs in zip(self.optimizers, lr_groups_l):
            for group, lr in zip(optimizer.param_groups, lrs):
                group['lr'] = lr

    def update_learning_rate(self, cur_iter, warmup_iter=-1):
        for scheduler in self.schedulers:
            scheduler.step()
        if cur_iter < warmup_iter:             # linear ramp towards the initial rates
            ramp = cur_iter / warmup_iter
            self._set_lr([[lr * ramp for lr in lrs] for lrs in self._get_init_lr()])

    # ---- introspection -----------------------------------------------------------------------------------------
    def get_network_description(self, network):
        net = unwrap(network)
        return str(net), sum(p.numel() for p in net.parameters())

    # ---- what a concrete wrapper provides ------------------------------------------------------------------------
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
