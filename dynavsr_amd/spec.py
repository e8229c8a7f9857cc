"""Parameter inventory of the two networks on the hot path.

The names and shapes are the state-dict contract of the reference: a checkpoint
written by DynaVSR must load with ``strict=True``.  Order = registration order of
codes/models/archs/EDVR_arch.py:207-252 (EDVR), :65-93 (PCD_Align), :137-161 (TSA_Fusion),
codes/models/archs/dcn/deform_conv.py:258-272 (ModulatedDeformConvPack) and
codes/models/archs/LRimg_estimator.py:70-91 (MFDN = DirectKernelEstimatorVideo).
The native engine (csrc/edvr_engine.hip) indexes parameters by position in this list.
"""
from collections import OrderedDict


def _conv(spec, name, cout, cin, k):
    spec[name + ".weight"] = (cout, cin, k, k)
    spec[name + ".bias"] = (cout,)


def _dcnpack(spec, name, nf, groups):
    spec[name + ".weight"] = (nf, nf, 3, 3)
    spec[name + ".bias"] = (nf,)
    _conv(spec, name + ".conv_offset_mask", groups * 3 * 9, nf, 3)


def edvr_param_spec(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=4):
    """OrderedDict name -> shape for EDVR(predeblur=False, HR_in=False, w_TSA=True)."""
    s = OrderedDict()
    _conv(s, "conv_first", nf, 3, 3)
    for i in range(front_RBs):
        _conv(s, "feature_extraction.%d.conv1" % i, nf, nf, 3)
        _conv(s, "feature_extraction.%d.conv2" % i, nf, nf, 3)
    for n in ("fea_L2_conv1", "fea_L2_conv2", "fea_L3_conv1", "fea_L3_conv2"):
        _conv(s, n, nf, nf, 3)
    p = "pcd_align."
    _conv(s, p + "L3_offset_conv1", nf, 2 * nf, 3)
    _conv(s, p + "L3_offset_conv2", nf, nf, 3)
    _dcnpack(s, p + "L3_dcnpack", nf, groups)
    for lv in ("L2", "L1"):
        _conv(s, p + lv + "_offset_conv1", nf, 2 * nf, 3)
        _conv(s, p + lv + "_offset_conv2", nf, 2 * nf, 3)
        _conv(s, p + lv + "_offset_conv3", nf, nf, 3)
        _dcnpack(s, p + lv + "_dcnpack", nf, groups)
        _conv(s, p + lv + "_fea_conv", nf, 2 * nf, 3)
    _conv(s, p + "cas_offset_conv1", nf, 2 * nf, 3)
    _conv(s, p + "cas_offset_conv2", nf, nf, 3)
    _dcnpack(s, p + "cas_dcnpack", nf, groups)
    t = "tsa_fusion."
    _conv(s, t + "tAtt_1", nf, nf, 3)
    _conv(s, t + "tAtt_2", nf, nf, 3)
    _conv(s, t + "fea_fusion", nf, nframes * nf, 1)
    _conv(s, t + "sAtt_1", nf, nframes * nf, 1)
    _conv(s, t + "sAtt_2", nf, 2 * nf, 1)
    _conv(s, t + "sAtt_3", nf, nf, 3)
    _conv(s, t + "sAtt_4", nf, nf, 1)
    _conv(s, t + "sAtt_5", nf, nf, 3)
    _conv(s, t + "sAtt_L1", nf, nf, 1)
    _conv(s, t + "sAtt_L2", nf, 2 * nf, 3)
    _conv(s, t + "sAtt_L3", nf, nf, 3)
    _conv(s, t + "sAtt_add_1", nf, nf, 1)
    _conv(s, t + "sAtt_add_2", nf, nf, 1)
    for i in range(back_RBs):
        _conv(s, "recon_trunk.%d.conv1" % i, nf, nf, 3)
        _conv(s, "recon_trunk.%d.conv2" % i, nf, nf, 3)
    if scale == 4:
        _conv(s, "upconv1", nf * 4, nf, 3)
    _conv(s, "upconv2", 64 * 4, nf, 3)      # 64*4 regardless of nf (EDVR_arch.py:246)
    _conv(s, "HRconv", 64, 64, 3)
    _conv(s, "conv_last", 3, 64, 3)
    return s


def mfdn_param_spec(nf=64, in_nc=3, scale=4):
    """OrderedDict name -> shape for MFDN (LRimg_estimator.py:70-91); conv0/conv5 are Conv3d."""
    s = OrderedDict()
    s["conv0.weight"] = (nf, in_nc, 3, 3, 3)
    s["conv0.bias"] = (nf,)
    _conv(s, "conv1", nf, nf, 3)
    _conv(s, "conv2", nf * 2, nf, 4)
    _conv(s, "conv3", nf, nf * 2, 3 if scale == 2 else 4)
    _conv(s, "conv4", nf, nf, 3)
    s["conv5.weight"] = (nf, nf, 3, 3, 3)
    s["conv5.bias"] = (nf,)
    _conv(s, "conv6", in_nc, nf, 1)
    return s


def sfdn_param_spec(nf=64):
    """OrderedDict name -> shape for SFDN (DirectKernelEstimator_CMS, LRimg_estimator.py:38-53)."""
    s = OrderedDict()
    _conv(s, "conv0", nf, 3, 3)
    _conv(s, "conv1", nf, nf, 3)
    _conv(s, "conv2", nf, nf, 3)
    _conv(s, "conv3", nf * 2, nf, 4)
    _conv(s, "conv4", nf * 2, nf * 2, 3)
    _conv(s, "conv5", nf, nf * 2, 3)
    _conv(s, "conv6", 3, nf, 1)
    return s


def tof_param_spec():
    """OrderedDict name -> shape of the TOFlow state dict (TOF_arch.py:25-110): parameters AND BatchNorm buffers,
    in the module's state_dict() order."""
    s = OrderedDict()
    chans = [(8, 32), (32, 64), (64, 32), (32, 16), (16, 2)]
    for b in range(4):
        for i, (ci, co) in enumerate(chans):
            _conv(s, "SpyNet.blocks.%d.block.%d" % (b, 3 * i), co, ci, 7)
            if i < 4:
                pre = "SpyNet.blocks.%d.block.%d" % (b, 3 * i + 1)
                s[pre + ".weight"] = (co,)
                s[pre + ".bias"] = (co,)
                s[pre + ".running_mean"] = (co,)
                s[pre + ".running_var"] = (co,)
                s[pre + ".num_batches_tracked"] = ()
    _conv(s, "conv_3x7_64_9x9", 64, 21, 9)
    _conv(s, "conv_64_64_9x9", 64, 64, 9)
    _conv(s, "conv_64_64_1x1", 64, 64, 1)
    _conv(s, "conv_64_3_1x1", 3, 64, 1)
    return s
