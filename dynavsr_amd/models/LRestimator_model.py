"""Wrapper around the down-scaling estimator netE (MFDN / SFDN): SLR = netE(LR).

The DynaVSR drivers touch this surface (codes/models/LRestimator_model.py:29-174; test_dynavsr.py:238-241,
267-269; train_dynavsr.py:360-362,417-426): ``feed_data`` -> ``forward_without_optim`` (with grad) or ``test`` (no
grad) -> ``fake_L``; ``real_H`` / ``real_L`` / ``var_H``; ``MyLoss``; ``optimize_parameters`` for the estimator's own
pre-training; ``load`` / ``save`` of ``*_E.pth``.  Clip layout: the data dict holds [B,T,C,H,W]; MFDN ('video' mode)
is fed [B,C,T,H,W], SFDN ('image' mode) gets the frames folded into the batch.
"""
import logging
from collections import OrderedDict

import torch

from . import networks
from .base_model import BaseModel, LogDict, unwrap

logger = logging.getLogger('base')
_LOSSES = {'l1': torch.nn.L1Loss, 'l2': torch.nn.MSELoss}


class LRimgestimator_Model(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        net_opt, train_set = opt['network_E'], opt['datasets']['train']
        self.train_opt = opt['train']
        self.rank = torch.distributed.get_rank() if opt['dist'] else -1
        self.scale, self.mode, self.model_name = opt['scale'], net_opt['mode'], net_opt['which_model_E']
        self.kernel_size = train_set['kernel_size']
        self.patch_size = train_set['patch_size']
        self.batch_size = train_set['batch_size']
        self.netE = networks.define_E(opt).to(self.device)
        self.load()
        loss_cls = _LOSSES.get(self.train_opt['loss_ftn'])
        self.MyLoss = loss_cls(reduction='mean').to(self.device) if loss_cls is not None else None
        if self.is_train:
            self._setup_training(self.train_opt)

    def name(self):
        return 'Estimator_Model'

    def _setup_training(self, t):
        if t['lr_scheme'] != 'MultiStepLR':
            raise NotImplementedError('MultiStepLR learning rate scheme is enough.')
        self.netE.train()
        trainable = [p for p in self.netE.parameters() if p.requires_grad]
        self.optimizer_E = torch.optim.Adam(trainable, lr=t['lr_C'], weight_decay=t['weight_decay_R'] or 0)
        self.optimizers.append(self.optimizer_E)
        self.schedulers.append(torch.optim.lr_scheduler.MultiStepLR(self.optimizer_E, list(t['lr_steps']), t['lr_gamma']))
        self.log_dict = LogDict()

    # ---- data in, estimate out -----------------------------------------------------------------------------------
    def feed_data(self, data):
        clip = data['LQs'].to(self.device)
        self.real_H = clip
        self.real_L = data['SuperLQs'].to(self.device) if 'SuperLQs' in data.keys() else None
        if self.mode == 'image':
            self.var_H = clip.reshape(-1, *clip.shape[2:])          # [B*T,C,H,W]
        else:
            self.var_H = clip.transpose(1, 2)                       # [B,C,T,H,W]

    def _estimate(self):
        y = self.netE(self.var_H)
        if self.mode != 'image':
            return y.transpose(1, 2)
        b, t, c = self.real_H.shape[:3]
        return y.reshape(b, t, c, y.shape[-2], y.shape[-1])

    def forward_without_optim(self, step=None):
        self.fake_L = self._estimate()

    def test(self):
        self.netE.eval()
        with torch.no_grad():
            self.fake_L = self._estimate()
        self.netE.train()

    def optimize_parameters(self, step=None):
        self.optimizer_E.zero_grad()
        self.forward_without_optim()
        lr_loss = self.MyLoss(self.fake_L, self.real_L)
        self.log_dict['l_pix'] = lr_loss.detach().clone()
        lr_loss.backward()
        self.optimizer_E.step()

    # ---- reporting -----------------------------------------------------------------------------------------------
    def get_current_log(self):
        return self.log_dict

    def get_current_visuals(self, need_GT=True):
        centre = self.fake_L.size(1) // 2

        def frame(clip):
            return clip.detach()[0, centre].float().cpu()
        visuals = OrderedDict(LQ=frame(self.real_L), rlt=frame(self.fake_L))
        if need_GT:
            visuals['GT'] = frame(self.real_H)
        return visuals

    def print_network(self):
        text, count = self.get_network_description(self.netE)
        logger.info('Network R structure: {}, with parameters: {:,d}'.format(type(unwrap(self.netE)).__name__, count))
        logger.info(text)

    # ---- checkpoints ---------------------------------------------------------------------------------------------
    def load(self):
        ckpt = self.opt['path']['pretrain_model_E']
        if ckpt is None:
            return
        logger.info('Loading pretrained model for E [{:s}] ...'.format(ckpt))
        self.load_network(ckpt, self.netE)

    def save(self, iter_step):
        self.save_network(self.netE, 'E', iter_step)
