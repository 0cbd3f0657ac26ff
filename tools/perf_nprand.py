"""np.random.permutation(n)[:k] three ways at PseudoLR's shape: numpy, the host-only library form, the device walk."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seesaw_amd.nprand import permutation_prefix  # noqa: E402

n, k = 1559883, 10000
np.random.seed(0)
for name, fn in (("numpy", lambda: np.random.permutation(n)[:k]), ("library, host walk", lambda: permutation_prefix(n, k)),
                 ("library, device walk", lambda: permutation_prefix(n, k, device=0))):
    fn()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    print(f"{name}: {1e2 * (time.perf_counter() - t0):.3f} ms per draw of {k} of {n}")
