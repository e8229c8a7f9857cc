"""Run-to-run spread of the EDVR weight gradients (fp32 MFMA vs itself, 3-way bf16 split vs fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
import torch
from dynavsr_amd import synth
from dynavsr_amd.models.archs.EDVR_arch import EDVR

def relerr(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))

def grads(mode, P, xg, go):
    net = EDVR(bf16_mfma=mode)
    net.load_state_dict(P, strict=True)
    net = net.cuda()
    y = net(xg)
    y.backward(go)
    torch.cuda.synchronize()
    return y.detach().clone(), [p.grad.clone() for p in net.parameters()], [n for n, _ in net.named_parameters()]

P = synth.edvr_state_dict(4)
xg = synth.clip(11, 1, 5, 32, 48).cuda()
torch.manual_seed(0)
go = torch.randn(1, 3, 128, 192, device="cuda")
runs = {}
for tag, mode in (("f32a", 0), ("f32b", 0), ("splita", 2), ("splitb", 2)):
    runs[tag] = grads(mode, P, xg, go)
names = runs["f32a"][2]
for a, b in (("f32a", "f32b"), ("splita", "splitb"), ("splita", "f32a")):
    ya, ga, _ = runs[a]; yb, gb, _ = runs[b]
    errs = [(relerr(u, v), n) for u, v, n in zip(ga, gb, names)]
    errs.sort(reverse=True)
    print("%s vs %s: y rel %.2e | worst grads %s" % (a, b, relerr(ya, yb), ", ".join("%s %.1e" % (n, e) for e, n in errs[:4])))

ya, ga, _ = runs["splita"]; yb, gb, _ = runs["f32a"]
for u, v, n in zip(ga, gb, names):
    if n.endswith("weight"):
        print("  %-40s %.1e" % (n, relerr(u, v)))
