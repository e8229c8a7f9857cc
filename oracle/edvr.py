"""Pure-torch functional restatement of the EDVR backbone (TEST INFRASTRUCTURE, CPU).

Each function cites the reference lines it follows.  The graph is written functionally over a
``{name: tensor}`` parameter dict that uses the reference's state-dict names, so the same
dict drives the reference modules (oracle/gen_golden.py), this oracle and the HIP engine.
Pinned against the imported reference by tests/golden/edvr_*.npz (see gen_golden.py).

``taps`` (optional dict) receives named intermediate tensors for layer-by-layer parity checks.
"""
import torch
import torch.nn.functional as F

from . import dcn as _dcn


def _c(P, name, x, stride=1, pad=None):
    w = P[name + ".weight"]
    pad = w.shape[-1] // 2 if pad is None else pad
    return F.conv2d(x, w, P[name + ".bias"], stride=stride, padding=pad)


def _lrelu(x):
    return F.leaky_relu(x, 0.1)   # EDVR_arch.py:93,161,252


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


def res_block(P, name, x):
    """arch_util.py:48-52  x + conv2(relu(conv1(x)))  (ReLU, not LeakyReLU)."""
    return x + _c(P, name + ".conv2", F.relu(_c(P, name + ".conv1", x)))


def dcn_pack(P, name, x, feat, groups):
    """deform_conv.py:274-291 with extra_offset_mask=True; offset = first 2/3 of the conv
    output (chunk+cat there is an identity re-slice), mask = sigmoid(last 1/3)."""
    om = _c(P, name + ".conv_offset_mask", feat)
    n_off = groups * 2 * 9
    offset, mask = om[:, :n_off], torch.sigmoid(om[:, n_off:])
    return _dcn.modulated_deform_conv(x, offset.contiguous(), mask.contiguous(), P[name + ".weight"],
                                      P[name + ".bias"], 1, 1, 1, 1, groups)


def pcd_align(P, nbr, ref, groups, taps=None, tag=""):
    """EDVR_arch.py:95-128.  nbr/ref = [L1, L2, L3] feature lists."""
    p = "pcd_align."
    t = (lambda k, v: taps.__setitem__(tag + k, v)) if taps is not None else (lambda k, v: None)
    # L3  (:100-103)
    o3 = _lrelu(_c(P, p + "L3_offset_conv1", torch.cat([nbr[2], ref[2]], 1)))
    o3 = _lrelu(_c(P, p + "L3_offset_conv2", o3))
    f3 = _lrelu(dcn_pack(P, p + "L3_dcnpack", nbr[2], o3, groups))
    t("L3_offset", o3); t("L3_fea", f3)
    # L2  (:105-112)
    o2 = _lrelu(_c(P, p + "L2_offset_conv1", torch.cat([nbr[1], ref[1]], 1)))
    o2 = _lrelu(_c(P, p + "L2_offset_conv2", torch.cat([o2, _up2(o3) * 2], 1)))
    o2 = _lrelu(_c(P, p + "L2_offset_conv3", o2))
    f2 = dcn_pack(P, p + "L2_dcnpack", nbr[1], o2, groups)          # no lrelu here (:110)
    f2 = _lrelu(_c(P, p + "L2_fea_conv", torch.cat([f2, _up2(f3)], 1)))
    t("L2_offset", o2); t("L2_fea", f2)
    # L1  (:114-121)
    o1 = _lrelu(_c(P, p + "L1_offset_conv1", torch.cat([nbr[0], ref[0]], 1)))
    o1 = _lrelu(_c(P, p + "L1_offset_conv2", torch.cat([o1, _up2(o2) * 2], 1)))
    o1 = _lrelu(_c(P, p + "L1_offset_conv3", o1))
    f1 = dcn_pack(P, p + "L1_dcnpack", nbr[0], o1, groups)
    f1 = _c(P, p + "L1_fea_conv", torch.cat([f1, _up2(f2)], 1))   # no lrelu (:121)
    t("L1_offset", o1); t("L1_fea", f1)
    # cascade  (:123-126)
    oc = _lrelu(_c(P, p + "cas_offset_conv1", torch.cat([f1, ref[0]], 1)))
    oc = _lrelu(_c(P, p + "cas_offset_conv2", oc))
    out = _lrelu(dcn_pack(P, p + "cas_dcnpack", f1, oc, groups))
    t("out", out)
    return out


def tsa_fusion(P, aligned, center, taps=None):
    """EDVR_arch.py:163-203.  aligned: [B, N, C, H, W]."""
    q = "tsa_fusion."
    b, n, c, h, w = aligned.shape
    emb_ref = _c(P, q + "tAtt_2", aligned[:, center])
    emb = _c(P, q + "tAtt_1", aligned.reshape(-1, c, h, w)).view(b, n, c, h, w)
    cor = torch.sigmoid((emb * emb_ref.unsqueeze(1)).sum(2))          # B, N, H, W  (:169-174)
    gated = (aligned * cor.unsqueeze(2)).reshape(b, n * c, h, w)       # (:175-176)
    fea = _lrelu(_c(P, q + "fea_fusion", gated))
    att = _lrelu(_c(P, q + "sAtt_1", gated))
    pooled = torch.cat([F.max_pool2d(att, 3, 2, 1), F.avg_pool2d(att, 3, 2, 1)], 1)
    att = _lrelu(_c(P, q + "sAtt_2", pooled))
    att_l = _lrelu(_c(P, q + "sAtt_L1", att))
    pooled = torch.cat([F.max_pool2d(att_l, 3, 2, 1), F.avg_pool2d(att_l, 3, 2, 1)], 1)
    att_l = _lrelu(_c(P, q + "sAtt_L2", pooled))
    att_l = _up2(_lrelu(_c(P, q + "sAtt_L3", att_l)))
    att = _lrelu(_c(P, q + "sAtt_3", att)) + att_l
    att = _up2(_lrelu(_c(P, q + "sAtt_4", att)))
    att = _c(P, q + "sAtt_5", att)
    att_add = _c(P, q + "sAtt_add_2", _lrelu(_c(P, q + "sAtt_add_1", att)))
    if taps is not None:
        taps["tsa_cor"] = cor
        taps["tsa_gated"] = gated
        taps["tsa_att"] = att
    return fea * torch.sigmoid(att) * 2 + att_add                     # (:200-202)


def edvr_forward(P, x, nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=4, center=None,
                 taps=None):
    """EDVR_arch.py:254-313 with predeblur=False, HR_in=False, w_TSA=True.  x: [B,N,3,H,W]."""
    b, n, c, h, w = x.shape
    center = n // 2 if center is None else center
    f1 = _lrelu(_c(P, "conv_first", x.reshape(-1, c, h, w)))
    for i in range(front_RBs):
        f1 = res_block(P, "feature_extraction.%d" % i, f1)
    f2 = _lrelu(_c(P, "fea_L2_conv2", _lrelu(_c(P, "fea_L2_conv1", f1, stride=2))))
    f3 = _lrelu(_c(P, "fea_L3_conv2", _lrelu(_c(P, "fea_L3_conv1", f2, stride=2))))
    f1 = f1.view(b, n, -1, h, w)
    f2 = f2.view(b, n, -1, h // 2, w // 2)
    f3 = f3.view(b, n, -1, h // 4, w // 4)
    if taps is not None:
        taps["L1_fea"], taps["L2_fea"], taps["L3_fea"] = f1, f2, f3
    ref = [f1[:, center], f2[:, center], f3[:, center]]
    aligned = torch.stack([pcd_align(P, [f1[:, i], f2[:, i], f3[:, i]], ref, groups, taps,
                                     "pcd%d_" % i) for i in range(n)], 1)
    if taps is not None:
        taps["aligned"] = aligned
    fea = tsa_fusion(P, aligned, center, taps)
    if taps is not None:
        taps["tsa_out"] = fea
    out = fea
    for i in range(back_RBs):
        out = res_block(P, "recon_trunk.%d" % i, out)
    if taps is not None:
        taps["recon"] = out
    if scale == 4:
        out = _lrelu(F.pixel_shuffle(_c(P, "upconv1", out), 2))
    out = _lrelu(F.pixel_shuffle(_c(P, "upconv2", out), 2))
    out = _lrelu(_c(P, "HRconv", out))
    out = _c(P, "conv_last", out)
    base = F.interpolate(x[:, center], scale_factor=scale, mode="bilinear", align_corners=False)
    return out + base


def charbonnier(x, y, eps=1e-6):
    """loss.py:26-30."""
    d = x - y
    return torch.mean(torch.sqrt(d * d + eps))
