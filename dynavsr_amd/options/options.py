"""Options-YAML surface of DynaVSR (codes/options/options.py:11-140): ``parse`` (YAML ->
OrderedDict with the derived ``is_train`` / dataset ``phase, scale, data_type`` / ``path.*`` /
``network_G.scale`` entries), ``dict2str``, ``NoneDict`` + ``dict_to_nonedict`` (missing key ->
None) and ``check_resume``.  Shipped YAMLs under options/{test,train} use the reference's key set.
"""
import logging
import os.path as osp
from collections import OrderedDict

import yaml


def _ordered_loader():
    class Loader(yaml.SafeLoader):
        pass

    Loader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG,
                           lambda loader, node: OrderedDict(loader.construct_pairs(node)))
    return Loader


def parse(opt_path, is_train=True, exp_name=None):
    with open(opt_path, mode='r') as f:
        opt = yaml.load(f, Loader=_ordered_loader())
    if exp_name is not None:
        opt['name'] = exp_name
    opt['is_train'] = is_train
    sr = opt['distortion'] == 'sr'
    for phase, ds in opt['datasets'].items():
        ds['phase'] = phase.split('_')[0]
        if sr:
            ds['scale'] = opt['scale']
        lmdb = False
        for key in ('dataroot_GT', 'dataroot_LQ'):
            if ds.get(key) is not None:
                ds[key] = osp.expanduser(ds[key])
                lmdb = lmdb or ds[key].endswith('lmdb')
        ds['data_type'] = 'lmdb' if lmdb else 'img'
        if ds['mode'].endswith('mc'):
            ds['data_type'] = 'mc'
            ds['mode'] = ds['mode'].replace('_mc', '')
    for key, path in opt['path'].items():
        if path and key != 'strict_load':
            if isinstance(path, OrderedDict):
                for sub, subpath in path.items():
                    if subpath:
                        path[sub] = osp.expanduser(subpath)
            else:
                opt['path'][key] = osp.expanduser(path)
    root = opt['path']['root'] = osp.abspath(osp.join(__file__, osp.pardir, osp.pardir, osp.pardir))
    if is_train:
        exp = osp.join(root, 'experiments', opt['name'])
        opt['path'].update(experiments_root=exp, models=osp.join(exp, 'models'),
                           training_state=osp.join(exp, 'training_state'), log=exp,
                           val_images=osp.join(exp, 'val_images'))
        if 'debug' in opt['name']:
            opt['train']['val_freq'] = 8
            opt['logger']['print_freq'] = 1
            opt['logger']['save_checkpoint_freq'] = 8
    else:
        res = osp.join(root, 'results', opt['name'])
        opt['path'].update(results_root=res, log=res)
    if sr:
        opt['network_G']['scale'] = opt['scale']
    return opt


def dict2str(opt, indent_l=1):
    msg = ''
    pad = ' ' * (indent_l * 2)
    for k, v in opt.items():
        if isinstance(v, dict):
            msg += pad + k + ':[\n' + dict2str(v, indent_l + 1) + pad + ']\n'
        else:
            msg += pad + k + ': ' + str(v) + '\n'
    return msg


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def check_resume(opt, resume_iter):
    logger = logging.getLogger('base')
    if opt['path']['resume_state']:
        if opt['path'].get('pretrain_model_G') is not None or opt['path'].get('pretrain_model_D') is not None:
            logger.warning('pretrain_model path will be ignored when resuming training.')
        opt['path']['pretrain_model_G'] = osp.join(opt['path']['models'], '{}_G.pth'.format(resume_iter))
        logger.info('Set [pretrain_model_G] to ' + opt['path']['pretrain_model_G'])
        if 'gan' in opt['model']:
            opt['path']['pretrain_model_D'] = osp.join(opt['path']['models'], '{}_D.pth'.format(resume_iter))
            logger.info('Set [pretrain_model_D] to ' + opt['path']['pretrain_model_D'])
