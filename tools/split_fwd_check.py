"""Forward accuracy of the bf16 modes on the residual branch (output minus the bilinear base)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dynavsr_amd import synth
from dynavsr_amd.models.archs.EDVR_arch import EDVR

def relerr(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))

P = synth.edvr_state_dict(4)
x = synth.clip(11, 1, 5, 32, 48)
xg = x.cuda()
base = F.interpolate(xg[:, 2], scale_factor=4, mode="bilinear", align_corners=False)
ys = {}
for mode in (0, 1, 2):
    net = EDVR(bf16_mfma=mode)
    net.load_state_dict(P, strict=True)
    net = net.cuda()
    with torch.no_grad():
        ys[mode] = net(xg)
from oracle import edvr as oedvr
with torch.no_grad():
    yo = oedvr.edvr_forward(P, x).cuda()
print("|residual| / |y| = %.3e" % float((ys[0] - base).norm() / ys[0].norm()))
for mode in (0, 1, 2):
    print("mode %d: y rel vs oracle %.2e | residual branch rel vs oracle %.2e" % (
        mode, relerr(ys[mode], yo), relerr(ys[mode] - base, yo - base)))
