import sys, torch
sys.path.insert(0, "/root/repo")
import mil_amd
from mil_amd import synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer
dev = torch.device("cuda")
B, N, L = 32, 4096, 1024
p = syn.image_only_params(1234, L=L)
x = syn.make_bags(4321, B, N, L).reshape(B * N, L).to(dev).to(torch.bfloat16)
y = syn.make_labels(99, B).to(dev)
lay = BagLayout.uniform(B, N, dev)
for tm in (False, True):
    tr = ImageOnlyTrainer(p, dev, train_mode=tm)
    for _ in range(40):
        tr.train_step(x, lay, y)
    kb, ev = tr.time_step_groups(x, lay, y, 30)
    print("train" if tm else "eval", {k: round(v * 1e3, 1) for k, v in kb.items()}, round(ev * 1e3, 1))
