"""Estimator wrapper with the reference's surface (codes/models/LRestimator_model.py:29-174):
``feed_data / forward_without_optim / optimize_parameters / test`` and the attributes
``netE, real_H, real_L, var_H, fake_L, MyLoss`` used by the DynaVSR drivers
(test_dynavsr.py:238-241,267-269; train_dynavsr.py:360-362,417-426)."""
import logging
from collections import OrderedDict

import torch
import torch.nn as nn

from . import networks
from .base_model import BaseModel, unwrap

logger = logging.getLogger('base')


class LRimgestimator_Model(BaseModel):
    def name(self):
        return 'Estimator_Model'

    def __init__(self, opt):
        super().__init__(opt)
        self.rank = torch.distributed.get_rank() if opt['dist'] else -1
        t = self.train_opt = opt['train']
        ds = opt['datasets']['train']
        self.kernel_size, self.patch_size, self.batch_size = ds['kernel_size'], ds['patch_size'], ds['batch_size']
        self.scale = opt['scale']
        self.model_name = opt['network_E']['which_model_E']
        self.mode = opt['network_E']['mode']
        self.netE = networks.define_E(opt).to(self.device)
        self.load()
        self.MyLoss = {'l1': nn.L1Loss(reduction='mean'), 'l2': nn.MSELoss(reduction='mean')}.get(t['loss_ftn'])
        if self.MyLoss is not None:
            self.MyLoss = self.MyLoss.to(self.device)
        if self.is_train:
            self.netE.train()
            wd = t['weight_decay_R'] if t['weight_decay_R'] else 0
            self.optimizer_E = torch.optim.Adam([p for p in self.netE.parameters() if p.requires_grad],
                                                lr=t['lr_C'], weight_decay=wd)
            self.optimizers.append(self.optimizer_E)
            if t['lr_scheme'] != 'MultiStepLR':
                raise NotImplementedError('MultiStepLR learning rate scheme is enough.')
            self.schedulers.append(torch.optim.lr_scheduler.MultiStepLR(self.optimizer_E, list(t['lr_steps']),
                                                                        t['lr_gamma']))
            self.log_dict = OrderedDict()

    def feed_data(self, data):
        self.real_H = data['LQs'].to(self.device)
        self.real_L = data['SuperLQs'].to(self.device) if 'SuperLQs' in data.keys() else None
        b, t, c, h, w = self.real_H.shape
        # 'image' mode folds frames into the batch (SFDN); 'video' mode feeds B,C,T,H,W (MFDN)
        self.var_H = self.real_H.reshape(b * t, c, h, w) if self.mode == 'image' else self.real_H.transpose(1, 2)

    def _estimate(self):
        y = self.netE(self.var_H)
        if self.mode == 'image':
            b, t, c = self.real_H.shape[:3]
            return y.reshape(b, t, c, *y.shape[-2:])
        return y.transpose(1, 2)

    def forward_without_optim(self, step=None):
        self.fake_L = self._estimate()

    def optimize_parameters(self, step=None):
        self.optimizer_E.zero_grad()
        self.fake_L = self._estimate()
        loss = self.MyLoss(self.fake_L, self.real_L)
        self.log_dict['l_pix'] = loss.item()
        loss.backward()
        self.optimizer_E.step()

    def test(self):
        self.netE.eval()
        with torch.no_grad():
            self.fake_L = self._estimate()
        self.netE.train()

    def get_current_log(self):
        return self.log_dict

    def get_current_visuals(self, need_GT=True):
        out = OrderedDict()
        mid = self.fake_L.size(1) // 2
        out['LQ'] = self.real_L.detach()[0, mid].float().cpu()
        out['rlt'] = self.fake_L.detach()[0, mid].float().cpu()
        if need_GT:
            out['GT'] = self.real_H.detach()[0, mid].float().cpu()
        return out

    def print_network(self):
        s, n = self.get_network_description(self.netE)
        logger.info('Network R structure: {}, with parameters: {:,d}'.format(
            unwrap(self.netE).__class__.__name__, n))
        logger.info(s)

    def load(self):
        path = self.opt['path']['pretrain_model_E']
        if path is not None:
            logger.info('Loading pretrained model for E [{:s}] ...'.format(path))
            self.load_network(path, self.netE)

    def save(self, iter_step):
        self.save_network(self.netE, 'E', iter_step)
