import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from wino5_debug import conv, report  # noqa
import torch.nn.functional as F
n, c, cout, h, w = 1, 16, 64, 8, 64
for chn in (1, 2, 3):
    wt = torch.zeros(cout, c, 3, 3); wt[:, chn, 1, 1] = 1.0
    for kind in ("ones", "xramp", "yramp"):
        x = torch.zeros(n, c, h, w)
        if kind == "ones": x[:, chn] = 1.0
        if kind == "xramp": x[:, chn] = torch.arange(w).float().view(1, 1, w)
        if kind == "yramp": x[:, chn] = torch.arange(h).float().view(1, h, 1)
        y, geo = conv(x, wt)
        print("ch", chn, kind, "\n", y[0, 0, :8, :12].numpy().round(2))
