"""The reference's integer filter taps, from the committed fixture
tests/golden/ref_tables.npz (tests/golden/make_ref_tables.py reads them out of the
reference HDL text): what impulse responses of the oracle must replay."""
import os

import numpy as np

_T = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tables.npz"))


def _interleave(p1, p2):
    h = []
    for k in range(len(p1)):
        h.append(int(p1[k]))
        if k < len(p1) - 1:
            h.append(int(p2[k]))
    return h


RX_CICCOMP_H = _interleave(_T["rx_ciccomp_phase1"], _T["rx_ciccomp_phase2"])  # rx_ciccomp.vhd:81-146
RX_HILB_C = [int(v) for v in _T["rx_hilb_coeff"]]
TX_CICCOMP_P1 = [int(v) for v in _T["tx_ciccomp_phase1"]]
TX_CICCOMP_P2 = [int(v) for v in _T["tx_ciccomp_phase2"]]
