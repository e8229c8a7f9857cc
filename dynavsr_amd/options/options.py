"""The options-YAML surface the DynaVSR drivers and wrappers rely on (reference: codes/options/options.py):

* ``parse(path, is_train, exp_name)``   YAML -> ordered dict, plus the entries the reference derives:
  ``is_train``; per dataset ``phase`` (key up to the first '_'), ``scale`` (super-resolution runs), ``data_type``
  ('lmdb' | 'img' | 'mc'); user-expanded ``path.*``; ``path.root`` and the experiment / result directories;
  ``network_G.scale``; the shortened intervals of runs whose name contains 'debug'  (:11-88)
* ``dict2str``                         nested pretty-printer used for the log header            (:91-101)
* ``NoneDict`` / ``dict_to_nonedict``  a missing key reads as None instead of raising           (:104-123)
* ``check_resume``                     re-points pretrain paths at the checkpoint being resumed (:126-140)
"""
import logging
import os.path as osp
from collections import OrderedDict

import yaml

_REPO_ROOT = osp.abspath(osp.join(__file__, osp.pardir, osp.pardir, osp.pardir))


class _OrderedLoader(yaml.SafeLoader):
    """Mappings keep their file order (the drivers iterate over opt['datasets'])."""


_OrderedLoader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG,
                               lambda loader, node: OrderedDict(loader.construct_pairs(node)))


def _annotate_dataset(key, ds, scale):
    ds['phase'] = key.split('_')[0]
    if scale is not None:
        ds['scale'] = scale
    roots = [ds[k] for k in ('dataroot_GT', 'dataroot_LQ') if ds.get(k) is not None]
    for k in ('dataroot_GT', 'dataroot_LQ'):
        if ds.get(k) is not None:
            ds[k] = osp.expanduser(ds[k])
    kind = 'lmdb' if any(r.endswith('lmdb') for r in roots) else 'img'
    if ds['mode'].endswith('mc'):              # memcached variant: the suffix is a storage hint, not a mode
        kind, ds['mode'] = 'mc', ds['mode'].replace('_mc', '')
    ds['data_type'] = kind


def _expand(paths):
    for key, value in paths.items():
        if key == 'strict_load' or not value:
            continue
        if isinstance(value, OrderedDict):
            _expand(value)
        else:
            paths[key] = osp.expanduser(value)


def parse(opt_path, is_train=True, exp_name=None):
    with open(opt_path) as f:
        opt = yaml.load(f, Loader=_OrderedLoader)
    if exp_name is not None:
        opt['name'] = exp_name
    opt['is_train'] = is_train
    scale = opt['scale'] if opt['distortion'] == 'sr' else None
    for key, ds in opt['datasets'].items():
        _annotate_dataset(key, ds, scale)
    _expand(opt['path'])
    out = opt['path']
    out['root'] = _REPO_ROOT
    if is_train:
        base = osp.join(_REPO_ROOT, 'experiments', opt['name'])
        out['experiments_root'] = out['log'] = base
        for sub in ('models', 'training_state', 'val_images'):
            out[sub] = osp.join(base, sub)
        if 'debug' in opt['name']:             # quick smoke runs: validate / log / checkpoint almost immediately
            opt['train']['val_freq'] = 8
            opt['logger'].update(print_freq=1, save_checkpoint_freq=8)
    else:
        out['results_root'] = out['log'] = osp.join(_REPO_ROOT, 'results', opt['name'])
    if scale is not None:
        opt['network_G']['scale'] = scale
    return opt


def dict2str(opt, indent_l=1):
    pad, parts = ' ' * (2 * indent_l), []
    for key, value in opt.items():
        if isinstance(value, dict):
            parts += [pad, key, ':[\n', dict2str(value, indent_l + 1), pad, ']\n']
        else:
            parts += [pad, key, ': ', str(value), '\n']
    return ''.join(parts)


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, list):
        return [dict_to_nonedict(item) for item in opt]
    if isinstance(opt, dict):
        return NoneDict((key, dict_to_nonedict(value)) for key, value in opt.items())
    return opt


def check_resume(opt, resume_iter):
    """Resuming overrides any pretrain path with the checkpoint written at ``resume_iter``."""
    paths = opt['path']
    if not paths['resume_state']:
        return
    log = logging.getLogger('base')
    if paths.get('pretrain_model_G') is not None or paths.get('pretrain_model_D') is not None:
        log.warning('pretrain_model path will be ignored when resuming training.')
    for label in ('G', 'D') if 'gan' in opt['model'] else ('G',):
        key = 'pretrain_model_' + label
        paths[key] = osp.join(paths['models'], '{}_{}.pth'.format(resume_iter, label))
        log.info('Set [{}] to {}'.format(key, paths[key]))
