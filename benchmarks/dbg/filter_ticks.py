"""Where a filter wave's cycles go (needs a cycle-stamped build of match_l2_filter.hip that exports ph_dbg_zero / ph_dbg_read:
POSEHOT_LIB=...): barrier wait, MFMA section (fragment reads + chains), wait for the next tile's copy, filter VALU."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pose_estimation_amd import matching, synthetic, _lib as L
dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scs = [synthetic.two_view_pair(n=8192, seed=1002 + p) for p in range(min(P, 8))]
a = torch.from_numpy(np.stack([scs[p % len(scs)]["desc1"] for p in range(P)])).to(dev)
b = torch.from_numpy(np.stack([scs[p % len(scs)]["desc2"] for p in range(P)])).to(dev)
lib = L.load()
for _ in range(2):
    matching.match_l2_ratio(a, b, 0.75)
torch.cuda.synchronize()
lib.ph_dbg_zero()
matching.match_l2_ratio(a, b, 0.75)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
lib.ph_dbg_read(out)
bar, mfma, vm, flt, waves = [int(out[i]) for i in range(5)]
tot = bar + mfma + vm + flt
print(f"waves {waves}; cycles per wave: barrier {bar / waves:.0f} ({bar / tot:.1%}), MFMA section {mfma / waves:.0f} ({mfma / tot:.1%}), "
      f"copy wait {vm / waves:.0f} ({vm / tot:.1%}), filter VALU {flt / waves:.0f} ({flt / tot:.1%}); total {tot / waves:.0f}")
