"""cfg5 (32 x 4096 x 1024, bf16 storage) in model.train() mode: in-step launch groups next to the eval-mode ones."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
for mode in (False, True):
    tr = ImageOnlyTrainer(p, dev, train_mode=mode)
    for _ in range(30):
        tr.train_step(x, lay, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        tr.train_step(x, lay, y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    kb, _ = tr.time_step_groups(x, lay, y, 30)
    print("train" if mode else "eval ", round(ms, 4), {k: round(v, 4) for k, v in kb.items()})
    del tr
