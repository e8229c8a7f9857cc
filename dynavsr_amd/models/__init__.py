"""Model factory with the reference's surface: ``create_model(opt)``.

Mirrors codes/models/__init__.py:5-37.  ``opt['model']`` is a '+'-joined list of wrapper names;
DynaVSR uses ``'video_base+lrimgestimator'`` and gets ``[VideoBaseModel, LRimgestimator_Model]``.
Only the two wrappers on the hot path exist here (SR/SRGAN/classifier wrappers are
BasicSR leftovers no DynaVSR YAML selects, two of them do not even exist in the reference tree).
"""
import logging

logger = logging.getLogger('base')

_WRAPPERS = {
    'video_base': ('Video_base_model', 'VideoBaseModel'),
    'lrimgestimator': ('LRestimator_model', 'LRimgestimator_Model'),
}


def _make(kind, opt):
    if kind not in _WRAPPERS:
        raise NotImplementedError('Model [{:s}] not recognized.'.format(kind))
    mod, cls = _WRAPPERS[kind]
    import importlib
    m = getattr(importlib.import_module('.' + mod, __name__), cls)(opt)
    logger.info('Model [{:s}] is created.'.format(m.__class__.__name__))
    return m


def create_model(opt):
    from .. import configure_runtime
    configure_runtime()     # before the wrappers touch the device (dynavsr_amd/_lib.py:configure_runtime)
    kinds = opt['model']
    if '+' in kinds:
        return [_make(k, opt) for k in kinds.split('+')]
    return _make(kinds, opt)
