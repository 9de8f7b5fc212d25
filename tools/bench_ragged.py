"""One ragged bag per step (the authors' regime), image-only fused step: bags with N ~ U[2000, 15592] patches, lengths on
the device, one hipGraph per capacity bucket.  Prints ms/step for the bucketed-graph path and for the exact-length eager path."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer, RaggedImageOnlyStepper

dev = torch.device("cuda")
L, steps = 512, 300
p = syn.image_only_params(1, L=L)
rng = np.random.default_rng(0)
lens = [int(v) for v in rng.integers(2000, 15593, size=steps)]
xs = torch.randn((16384, L), device=dev)
y = syn.make_labels(3, 1).to(dev)
out = {}
for mode in ("bucket_graph", "exact_eager"):
    tr = ImageOnlyTrainer(p, dev, train_mode=True, counted=True)
    st = RaggedImageOnlyStepper(tr, B=1)
    def run(n):
        if mode == "bucket_graph":
            slot = st.slot(n)
            slot.x[:n].copy_(xs[:n], non_blocking=True)      # stands for the loader's H2D copy into the bucket's buffer
            slot.y.copy_(y)
            st.step(slot, [n])
        else:
            tr.train_step(xs[:n], BagLayout.make([n], dev), y)
    for n in lens[:40]:
        run(n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for n in lens:
        run(n)
    torch.cuda.synchronize()
    out[mode] = round((time.perf_counter() - t0) / steps * 1e3, 4)
    if mode == "bucket_graph":
        out["graphs"] = sum(s.graph is not None for s in st.slots.values())
out["mean_patches"] = float(np.mean(lens))
print(json.dumps({"workload": "1 ragged bag/step, N~U[2000,15592] x 512, image-only train-mode step", **out}))
