"""Which torch ops still launch kernels in the fusion step (config 3)?  Runs the eager step of tools/bench_fusion.py under
torch.profiler and lists every aten op that owns a device kernel / copy (i.e. everything that is NOT one of the library's
own launches), with its shapes, its chain of parent ops (autograd node names) and the package's Python frames."""
import collections, os, sys
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd
from mil_amd import synthetic as syn
from mil_amd.model.utils import get_model
from mil_amd.optim import FlatAdam
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda")
prompts = int(os.environ.get("PROMPTS", "1"))
args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL", num_classes=2,
                       learnablePrompt=0, n_ctx=8, clinical_features=["f"] * 9, clip_gemm_pieces=0, alignment_base="CI",
                       model_CT="resnetMC3_18", clip_layers=12, cache_text=1)
torch.manual_seed(1234)
model = get_model(args).to(dev).eval()
x = syn.make_bags(1, 32, 1024, 768).to(dev)
ids = syn.make_token_ids(2, 32, prompts).to(dev)
y = syn.make_labels(3, 32).to(dev)
opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-5, weight_decay=1e-7, counted=True)
with torch.no_grad():
    tfeat = model.clinic_extractor(ids)


def step():
    model([x], ids, labels=y, text_features=tfeat)
    loss = model.last_loss
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

agg = collections.OrderedDict()
for e in prof.events():
    ks = getattr(e, "kernels", None) or []
    if not ks:
        continue
    # only leaf ops (a parent's .kernels repeats its children's)
    if any((getattr(c, "kernels", None) or []) for c in (e.cpu_children or [])):
        continue
    chain, p = [], e.cpu_parent
    while p is not None and len(chain) < 4:
        chain.append(p.name[:60])
        p = p.cpu_parent
    frames = [f.split("/")[-1] for f in (e.stack or []) if "mil_amd" in f or "llm-guided" in f or "bench_fusion" in f or "prof_glue" in f][:3]
    key = (e.name, str(e.input_shapes)[:80], " < ".join(chain), " | ".join(frames))
    d = agg.setdefault(key, [0, 0.0, set()])
    d[0] += 1
    d[1] += sum(k.duration for k in ks)
    d[2].update(k.name[:50] for k in ks)
tot = 0.0
for (name, shapes, chain, frames), (n, us, kn) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += us
    print(f"{us:8.1f} us x{n:<3d} {name:22s} {shapes}\n           parents: {chain}\n           frames : {frames}\n           kernels: {sorted(kn)}")
print("total device time of torch-owned launches: %.1f us" % tot)
