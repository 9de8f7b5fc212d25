"""Config 5 (32 x 4096 x 1024, bf16) step time and in-step launch groups, eval and train mode; MIL_FUSE_POOL=0/1 compares the
stand-alone pool pass with the pass in the forward's epilogue."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer

dev = torch.device("cuda")
B, N, L = 32, 4096, 1024
p = syn.image_only_params(1234, L=L)
x = syn.make_bags(4321, B, N, L).reshape(B * N, L).to(dev).to(torch.bfloat16)
y = syn.make_labels(99, B).to(dev)
lay = BagLayout.uniform(B, N, dev)
out = {"MIL_FUSE_POOL": os.environ.get("MIL_FUSE_POOL", "1")}
for mode in (False, True):
    tr = ImageOnlyTrainer(p, dev, train_mode=mode)
    for _ in range(40):
        tr.train_step(x, lay, y)
    torch.cuda.synchronize()
    runs = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(60):
            tr.train_step(x, lay, y)
        torch.cuda.synchronize()
        runs.append((time.perf_counter() - t0) / 60 * 1e3)
    kb, ev = tr.time_step_groups(x, lay, y, 30)
    out["train" if mode else "eval"] = {"ms_per_step": round(sorted(runs)[2], 4), "groups": {k: round(v, 4) for k, v in kb.items()}}
    del tr
print(json.dumps(out))
