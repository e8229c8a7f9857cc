"""VSR wrapper with the reference's driver-facing surface (codes/models/Video_base_model.py:16-251):
``feed_data / calculate_loss / optimize_parameters / optimize_by_loss / test /
get_current_visuals / get_current_log / load / save`` and the attributes ``netG, var_L, real_H,
fake_H, log_dict, optimizers, schedulers`` that test_dynavsr.py / train_dynavsr.py touch.

``netG`` is the engine-backed EDVR (one native call per forward/backward); it deep-copies,
exposes ordinary leaf nn.Parameters and can be re-assigned (``modelcp.netG = deepcopy(...)``,
test_dynavsr.py:208).  ``log_dict['l_pix']`` reads as the float the reference stores (Video_base_model.py:194); the host
synchronisation its eager ``.item()`` costs happens when the entry is read, not inside the step
(base_model.LogDict).
"""
import logging
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import networks
from .base_model import BaseModel, LogDict, unwrap
from .loss import CharbonnierLoss

logger = logging.getLogger('base')


def _pixel_criterion(kind):
    if kind == 'cb':
        return CharbonnierLoss()
    if kind == 'l1':
        return nn.L1Loss(reduction='mean')
    if kind == 'l2':
        return nn.MSELoss(reduction='mean')
    raise NotImplementedError('Loss type [{:s}] is not recognized.'.format(str(kind)))


class VideoBaseModel(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.rank = torch.distributed.get_rank() if opt['dist'] else -1
        train_opt = opt['train']
        self.netG = networks.define_G(opt).to(self.device)
        self.load()
        self.log_dict = LogDict()
        self.cri_pix = _pixel_criterion(train_opt['pixel_criterion']).to(self.device)
        self.l_pix_w = train_opt['pixel_weight']
        if self.is_train:
            self.netG.train()
            self._build_optimizer(train_opt)

    def _build_optimizer(self, t):
        """Plain variant of Video_base_model.py:52-154: one group over all trainable params, Adam or
        SGD, MultiStepLR.  (ft_tsa_only / freeze_front / restart schedulers belong to the
        pre-training pipelines that are out of this build's scope)"""
        for key in ('ft_tsa_only', 'freeze_front', 'small_offset_lr'):
            if t[key]:
                raise NotImplementedError('train.%s is a pre-training option not covered here' % key)
        params = [p for p in self.netG.parameters() if p.requires_grad]
        wd = t['weight_decay_G'] if t['weight_decay_G'] else 0
        if t['optim'] == 'SGD':
            self.optimizer_G = torch.optim.SGD(params, lr=t['lr_G'], weight_decay=wd)
        else:
            self.optimizer_G = torch.optim.Adam(params, lr=t['lr_G'], weight_decay=wd,
                                                betas=(t['beta1'], t['beta2']))
        self.optimizers.append(self.optimizer_G)
        if t['lr_scheme'] == 'MultiStepLR':
            self.schedulers.append(torch.optim.lr_scheduler.MultiStepLR(
                self.optimizer_G, list(t['lr_steps']), t['lr_gamma']))
        else:
            raise NotImplementedError('lr_scheme [{}]'.format(t['lr_scheme']))

    # ---- data / forward / loss -------------------------------------------------------------
    def feed_data(self, data, need_GT=True):
        self.var_L = data['LQs'].to(self.device)       # may carry an autograd graph (the SLR clip)
        if need_GT:
            self.real_H = data['GT'].to(self.device)

    def _pixel_loss(self):
        self.fake_H = self.netG(self.var_L)
        return self.l_pix_w * self.cri_pix(self.fake_H, self.real_H)

    def calculate_loss(self):
        l_pix = self._pixel_loss()
        self.log_dict['l_pix'] = l_pix.detach().clone()   # own storage: the drivers add to the returned loss IN PLACE (test_dynavsr.py:264-274)
        return l_pix

    def optimize_parameters(self, step):
        self.optimizer_G.zero_grad()
        l_pix = self._pixel_loss()
        l_pix.backward()
        self.optimizer_G.step()
        self.log_dict['l_pix'] = l_pix.detach().clone()   # own storage: the drivers add to the returned loss IN PLACE (test_dynavsr.py:264-274)

    def optimize_by_loss(self, loss):
        self.optimizer_G.zero_grad()
        loss.backward()
        self.optimizer_G.step()
        self.log_dict['l_pix'] = loss.detach().clone()

    def test(self):
        self.netG.eval()
        with torch.no_grad():
            self.fake_H = self.netG(self.var_L)
        self.netG.train()

    def get_current_log(self):
        return self.log_dict

    def get_current_visuals(self, need_GT=True):
        out = OrderedDict()
        out['LQ'] = self.var_L.detach()[0].float().cpu()
        out['rlt'] = self.fake_H.detach()[0].float().cpu()
        if need_GT:
            out['GT'] = self.real_H.detach()[0].float().cpu()
        return out

    def print_network(self):
        s, n = self.get_network_description(self.netG)
        if self.rank <= 0:
            logger.info('Network G structure: {}, with parameters: {:,d}'.format(
                unwrap(self.netG).__class__.__name__, n))
            logger.info(s)

    # ---- checkpoints -----------------------------------------------------------------------
    def load(self, verbose=True):
        path = self.opt['path']['pretrain_model_G']
        if path is not None:
            if verbose:
                logger.info('Loading model for G [{:s}] ...'.format(path))
            self.load_network(path, self.netG, self.opt['path']['strict_load'])

    def load_for_test(self):
        self.load_network(os.path.join(self.opt['path']['models'], 'latest_G.pth'), self.netG,
                          self.opt['path']['strict_load'])

    def save(self, iter_label):
        self.save_network(self.netG, 'G', iter_label)

    def save_for_test(self):
        sd = OrderedDict((k, v.cpu()) for k, v in unwrap(self.netG).state_dict().items())
        torch.save(sd, os.path.join(self.opt['path']['models'], 'latest_G.pth'))
