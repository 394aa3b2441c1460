"""G12: the reference's CME order table and constructor term snapping (config.py:278-418, w_nl.py:86-88).

    python tests/golden/make_golden_cme.py        # needs /root/reference (build container only)

Stores data only: the array ``config.CME_reconstruction_terms()`` returns and, for requested term counts 4..1100, the
snapped count the reference constructor's expression ``terms[np.argmin(terms < s) - 2]`` yields.
"""
import os
import sys

import numpy as np

sys.argv = [sys.argv[0]]
sys.path.insert(0, "/root/reference")
import config  # noqa: E402

terms = config.CME_reconstruction_terms()
req = np.arange(4, 1101)
snapped = np.array([terms[np.argmin(terms < s) - 2] for s in req])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g12_cme_terms.npz")
np.savez_compressed(out, terms=terms, requested=req, snapped=snapped)
print("wrote", out, len(terms), "orders")
