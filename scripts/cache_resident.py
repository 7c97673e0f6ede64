"""Stage times of one config-2 pair with whichever library SFFT_AMD_LIB names (scripts/cache_resident.sh).  With the cache-resident build the
results are wrong by construction: a failed solve is expected and ignored, only the stage timers are read."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sfft_amd import _lib            # noqa: E402
from sfft_amd.plan import Plan       # noqa: E402

dev = torch.device("cuda", 0)
N = 4096
g = torch.Generator(device="cpu").manual_seed(3)
imgs = [(torch.rand((N, N), generator=g, dtype=torch.float64) + 0.1).to(dev) for _ in range(4)]
plan = Plan(N, N, 8, 2, 2, True, device=0)
plan.set_timing(True)
acc, calls = {}, 7
for it in range(calls + 2):
    try:
        plan.subtract(imgs[0], imgs[1], imgs[2], imgs[3])
    except (np.linalg.LinAlgError, _lib.SfftError):
        pass
    torch.cuda.synchronize(dev)
    if it >= 2:
        for k, v in plan.stage_ms().items():
            acc.setdefault(k, []).append(v)
print(json.dumps({"which": sys.argv[1] if len(sys.argv) > 1 else "", "calls": calls, "ms": {k: float(np.median(v)) for k, v in acc.items()},
                  "kernels": plan.stage_kernels()}))
