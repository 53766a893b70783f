import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fused as t
make = t._bicycle(500)
opts = dict(iterations_max=40, use_backtracking=True)
runs = {
  "seq_nospec": {"ALTRO_HIP_NO_FUSED": "1", "ALTRO_HIP_NO_SPECULATION": "1"},
  "seq_spec": {"ALTRO_HIP_NO_FUSED": "1"},
  "fused_all": {"ALTRO_HIP_FUSED_SWEEPS": "1000"},
  "hand2": {"ALTRO_HIP_FUSED_SWEEPS": "2"},
  "hand2_nospec": {"ALTRO_HIP_FUSED_SWEEPS": "2", "ALTRO_HIP_NO_SPECULATION": "1"},
}
out = {k: t._solve(make, env, **opts) for k, env in runs.items()}
ref = out["seq_nospec"]
for name, r in out.items():
    if name == "seq_nospec": continue
    print("==", name, "sweeps", r[0]["sweeps"], "merit_launches", r[0]["merit_launches"])
    for key in t.KEYS:
        a, b = np.asarray(ref[0][key]), np.asarray(r[0][key])
        bad = np.flatnonzero(a != b)
        if len(bad): print("  ", key, len(bad), bad[:8], a[bad[:4]], b[bad[:4]])
    for i, nm in ((1, "x_nom"), (2, "u_nom"), (3, "x_cand"), (4, "K")):
        d = np.abs(ref[i] - r[i])
        if d.max() > 0: print("  ", nm, d.max(), np.argwhere(d > 0)[:3].tolist())
