r"""Physical constants: same values AND same dtype (0-dim ``torch.double``) as the reference's
``mrphy/__init__.py:58-65`` -- the dtype decides how the host-side constants γ2πdt, E1, E2 get
rounded when they meet fp32 data.
"""
import torch
from torch import tensor

γH = tensor(4257.6, dtype=torch.double)    # Hz/Gauss, water proton gyromagnetic ratio
T1G = tensor(1.47, dtype=torch.double)     # s, grey-matter T1
T2G = tensor(0.07, dtype=torch.double)     # s, grey-matter T2
dt0 = tensor(4e-6, dtype=torch.double)     # s, default dwell time
gmax0 = tensor(5, dtype=torch.double)      # Gauss/cm
smax0 = tensor(12e3, dtype=torch.double)   # Gauss/cm/s
rfmax0 = tensor(0.25, dtype=torch.double)  # Gauss
