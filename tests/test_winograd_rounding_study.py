"""CPU: the end-to-end rounding figures DESIGN.md / include/se3tracknet.h quote for the Winograd tiles are reproducible -- the oracle
network with the four 256 / 512-channel stride-1 convs replaced by a float32 emulation of the device algorithm
(scripts/study_winograd_rounding.py: U = G g G^T in float64 rounded once, transforms / products in float32) against a float64 forward
of the same weights and inputs.  Also pins the exact-rational Toom-Cook construction against the tables compiled into wino_mfma.hip."""
import importlib.util
import os

import numpy as np
import torch

from oracle import fixtures as Fx
from oracle import se3_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _study():
    spec = importlib.util.spec_from_file_location("study_winograd_rounding", os.path.join(ROOT, "scripts", "study_winograd_rounding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_generated_tables_equal_the_compiled_ones():
    st = _study()
    from tests.test_winograd_matrices import SRC, _matrix
    src = open(SRC).read()
    for m, name in ((2, "t2"), (4, "t4"), (6, "t6")):
        AT, G, BT = st.toom_cook(m, st.POINTS[m])
        assert st.check_identity(AT, G, BT, m) < 1e-9
        if m == 2:
            continue   # t2 uses the textbook scaling of G (1/2) with unit B^T / A^T: the same algorithm, another normalisation
        assert np.allclose(BT, _matrix(src, "wino_bt", name), rtol=0, atol=1e-12), m
        assert np.allclose(AT, _matrix(src, "wino_at", name), rtol=0, atol=1e-12), m
        assert np.allclose(G, _matrix(src, "wino_g", name), rtol=0, atol=1e-15), m


def test_end_to_end_rounding_of_the_tiles():
    st = _study()
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    sd = O.make_state_dict(1)
    A, B = Fx.net_inputs(1, 2)
    ref = O.forward({k: v.double() for k, v in sd.items()}, A.double(), B.double())
    want = torch.cat([ref["trans_logit"], ref["rot_logit"]], 1)
    err = {}
    for key in (None, 4, 6):
        out = st.forward_with(sd, A, B, st.WinoConv(key) if key else None)
        err[key] = float((torch.cat([out["trans_logit"], out["rot_logit"]], 1).double() - want).abs().max())
    print("max |d logit| vs float64: direct f32 %.2e, F(4x4) %.2e, F(6x6) %.2e" % (err[None], err[4], err[6]))
    assert err[None] < 2e-6 and err[4] < 2e-6 and err[6] < 4e-6      # profiles/EXPERIMENTS.md items 25: 0.5 / 0.6 / 1.4e-6
    assert err[6] < 6 * max(err[4], 3e-7)                             # not the 10-20 x a single layer's figures suggest
