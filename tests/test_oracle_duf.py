"""CPU: oracle/duf.py against the golden produced by the reference's own DUF modules (oracle/gen_golden.py duf)."""
from collections import OrderedDict

import torch

from conftest import load_golden, relerr
from dynavsr_amd import synth
from oracle import duf as oduf, edvr as oedvr


def test_duf_oracle_matches_reference_golden():
    g = load_golden("duf_16x24")
    P = synth.duf_state_dict(int(g["wseed"]), 16, 4)
    x = synth.clip(int(g["xseed"]), 1, 7, int(g["h"]), int(g["w"]))
    tgt = synth.clip(int(g["tseed"]), 1, 1, 4 * int(g["h"]), 4 * int(g["w"]))[:, 0]
    with torch.no_grad():
        y = oduf.duf_forward(OrderedDict((k, v.clone()) for k, v in P.items()), x, 16, 4, True, False)
    assert relerr(y, g["out_eval"]) < 1e-6
    names = [str(n) for n in g["grad_names"]]
    PO = OrderedDict((k, (v.clone().requires_grad_(True) if k in names else v.clone())) for k, v in P.items())
    y = oduf.duf_forward(PO, x, 16, 4, True, True)
    assert relerr(y, g["out_train"]) < 1e-6
    loss = oedvr.charbonnier(y, tgt)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6 * float(g["loss"])
    grads = torch.autograd.grad(loss, [PO[k] for k in names])
    for k, a, b in zip(names, grads, g["grad_norms"]):
        assert abs(float(a.norm()) - b) < 1e-4 * b + 1e-7, k
    assert relerr(PO["bn3d_2.running_var"], g["running_var_bn3d_2"]) < 1e-6
    for layers, scale in ((16, 2), (28, 4), (52, 3)):
        Pv = synth.duf_state_dict(4, layers, scale)
        xv = synth.clip(int(g["xseed"]) + layers, 1, 7, 8, 12)
        with torch.no_grad():
            yv = oduf.duf_forward(Pv, xv, layers, scale, True, False)
        assert relerr(yv, g["out_eval_%dL_x%d" % (layers, scale)]) < 1e-6


def test_duf_factory_and_cpu_refusal():
    import pytest
    from dynavsr_amd.models import networks
    from dynavsr_amd.options.options import dict_to_nonedict
    for layers, name in ((16, "DUF_16L"), (28, "DUF_28L"), (52, "DUF_52L")):
        net = networks.define_G(dict_to_nonedict({"network_G": {"which_model_G": "DUF", "layers": layers}, "scale": 4}))
        assert type(net).__name__ == name and net.adapt_official is True and net.scale == 4
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 7, 3, 8, 8))
