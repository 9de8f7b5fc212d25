"""The attention-pool stage alone at the north_star point (64 bags x 4096 x 512 fp32 = 512 MiB of x, beyond the Infinity
Cache), for rocprofv3 passes whose per-kernel averages must not mix shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import ops
from mil_amd.bags import BagLayout
dev = torch.device("cuda")
N, L, B = 4096, 512, 64
x = torch.randn(B * N, L, device=dev)
scores = torch.randn(B * N, device=dev)
lay = BagLayout.uniform(B, N, dev)
for _ in range(int(os.environ.get("STEPS", "20"))):
    ops.attn_pool_fwd(x, scores, lay)
torch.cuda.synchronize()
print("done")
