"""oracle/edvr_tape.py (EDVR in the launch order of the native tape, the vehicle of the kink-free gradient test) against
oracle/edvr.py, the restatement pinned by the reference's goldens: same output, loss and gradients in fp64."""
import torch

from dynavsr_amd import synth
from oracle import edvr as oedvr
from oracle import edvr_tape


def _grads(fn, P, x, tgt):
    Pd = {k: v.double().clone().requires_grad_(True) for k, v in P.items()}
    xd = x.double().clone().requires_grad_(True)
    y = fn(Pd, xd)
    loss = oedvr.charbonnier(y, tgt.double())
    loss.backward()
    return y.detach(), float(loss.detach()), xd.grad, {k: v.grad for k, v in Pd.items()}


def test_tape_order_restatement_equals_the_reference_pinned_oracle():
    P = synth.edvr_state_dict(4)
    x = synth.clip(11, 2, 5, 16, 24)
    tgt = synth.clip(12, 2, 1, 64, 96)[:, 0]
    names = []

    def tape(Pd, xd):
        y, nm = edvr_tape.edvr_forward_tape(Pd, xd)
        names.extend(nm)
        return y
    y0, l0, gx0, g0 = _grads(lambda Pd, xd: oedvr.edvr_forward(Pd, xd), P, x, tgt)
    y1, l1, gx1, g1 = _grads(tape, P, x, tgt)
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
    assert rel(y1, y0) < 1e-12 and abs(l1 - l0) < 1e-12 * abs(l0)
    assert rel(gx1, gx0) < 1e-10
    assert max(rel(g1[k], g0[k]) for k in g0) < 1e-10
    # 86 launches for B = 2 (84 + one base_up per clip), under the names the engine gives them
    assert len(names) == 85 and names[0] == "conv_first" and names[-1] == "conv_last" and names.count("fe_rb_a") == 5


def test_forcing_with_the_oracles_own_values_changes_nothing():
    """Teacher forcing with the unforced run's own launch outputs (fp64): identical values and gradients -- the forced
    activation and the straight-through substitution are consistent with the plain ops."""
    P = synth.edvr_state_dict(5)
    x = synth.clip(13, 1, 5, 16, 16)
    tgt = synth.clip(14, 1, 1, 64, 64)[:, 0]
    rec = {}

    def record(i, name, which, v):
        return None
    Pd = {k: v.double() for k, v in P.items()}
    outs = []

    class Rec:
        def __call__(self, i, name, which, v):
            return None
    # first pass: record every launch output
    t_out = []
    orig_out = edvr_tape.Tape.out

    def spy(self, name, y, y2=None):
        r = orig_out(self, name, y, y2)
        t_out.append((name, r if y2 is None else r[0], None if y2 is None else r[1]))
        return r
    edvr_tape.Tape.out = spy
    try:
        with torch.no_grad():
            edvr_tape.edvr_forward_tape(Pd, x.double())
    finally:
        edvr_tape.Tape.out = orig_out
    vals = [(a.clone(), None if b is None else b.clone()) for (_n, a, b) in t_out]

    def force(i, name, which, v):
        return vals[i][which]
    y0, l0, gx0, g0 = _grads(lambda Pq, xq: edvr_tape.edvr_forward_tape(Pq, xq)[0], P, x, tgt)
    y1, l1, gx1, g1 = _grads(lambda Pq, xq: edvr_tape.edvr_forward_tape(Pq, xq, force=force)[0], P, x, tgt)
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
    assert rel(y1, y0) < 1e-12 and rel(gx1, gx0) < 1e-10 and max(rel(g1[k], g0[k]) for k in g0) < 1e-10
