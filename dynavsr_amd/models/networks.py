"""Network factories with the reference's signatures (codes/models/networks.py:8-72):
``define_G(opt)`` builds the EDVR, TOFlow or DUF backbone from ``opt['network_G']``, ``define_E(opt)`` the
down-scaling estimator (MFDN / SFDN) from ``opt['network_E']``."""
from .archs import EDVR_arch, LRimg_estimator


def define_G(opt):
    net = opt['network_G']
    which = net['which_model_G']
    if which == 'TOF':      # networks.py:37-39
        from .archs import TOF_arch
        return TOF_arch.TOFlow(adapt_official=True)
    if which == 'DUF':      # networks.py:29-36
        from .archs import DUF_arch
        cls = {16: DUF_arch.DUF_16L, 28: DUF_arch.DUF_28L}.get(net['layers'], DUF_arch.DUF_52L)
        return cls(scale=opt['scale'], adapt_official=True)
    if which != 'EDVR':
        raise NotImplementedError('Generator model [{:s}] not recognized (this build covers the video backbones EDVR, '
                                  'TOF and DUF)'.format(str(which)))
    return EDVR_arch.EDVR(nf=net['nf'], nframes=net['nframes'], groups=net['groups'],
                          front_RBs=net['front_RBs'], back_RBs=net['back_RBs'], center=net['center'],
                          predeblur=net['predeblur'], HR_in=net['HR_in'], w_TSA=net['w_TSA'],
                          scale=opt['scale'], bf16_mfma=int(net.get('bf16_mfma') or 0))


def define_E(opt):
    net = opt['network_E']
    which = net['which_model_E']
    if which == 'MFDN':
        return LRimg_estimator.DirectKernelEstimatorVideo(in_nc=net['in_nc'], nf=net['nf'],
                                                          scale=opt['scale'])
    if which == 'SFDN':
        return LRimg_estimator.DirectKernelEstimator_CMS(nf=net['nf'])
    raise NotImplementedError('Estimator model [{:s}] not recognized'.format(str(which)))
