"""Loader for the committed golden fixtures (outputs of the reference, see tools/gen_goldens.py)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)

    def rec(self, key):
        """Re-assemble a ``summarize`` record stored as ``key/field`` arrays."""
        pre = key + "/"
        out = {k[len(pre):]: self.z[k] for k in self.z.files if k.startswith(pre)}
        assert out, f"golden key {key!r} missing"
        return out

    def arr(self, key):
        return self.z[key]

    def keys(self, prefix=""):
        seen = []
        for k in self.z.files:
            base = k.split("/")[0]
            if base.startswith(prefix) and base not in seen:
                seen.append(base)
        return seen
