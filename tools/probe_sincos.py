"""Accuracy scan of the two device sincos forms (debug entry points)."""
import numpy as np
import torch
from xrt_amd import hipcalls
rng = np.random.default_rng(2)
for table in (False, True):
    for top in (10., 1e6, 1e12, 4e12, 1e13, 1e14):
        phi = rng.uniform(-top, top, 1000000)
        s, c = hipcalls.debug_sincos(torch.as_tensor(phi, device="cuda"), table=table)
        print(table, top, np.abs(s.cpu().numpy() - np.sin(phi)).max(),
              np.abs(c.cpu().numpy() - np.cos(phi)).max())
    phi = np.concatenate([np.arange(-4096, 4097) * (np.pi / 1024),
                          (np.arange(-4096, 4097) + 0.5) * (np.pi / 1024)])
    s, c = hipcalls.debug_sincos(torch.as_tensor(phi, device="cuda"), table=table)
    print(table, 'nodes', np.abs(s.cpu().numpy() - np.sin(phi)).max(),
          np.abs(c.cpu().numpy() - np.cos(phi)).max())
