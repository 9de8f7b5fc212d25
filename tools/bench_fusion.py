"""BASELINE config 3 (pathology + CLIP-text fusion branch) fwd+bwd+Adam through aggregator(args) on synthetic bags."""
import argparse, json, os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd
from mil_amd import synthetic as syn, ops
from mil_amd.model.utils import get_model

ap = argparse.ArgumentParser()
ap.add_argument("--bags", type=int, default=32)
ap.add_argument("--patches", type=int, default=1024)
ap.add_argument("--prompts", type=int, default=1)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--clip_layers", type=int, default=12)
ap.add_argument("--cache_text", action="store_true")
ap.add_argument("--coop", action="store_true", help="learnable prompts (upstream default --learnablePrompt 1): 10 prompts "
                "per bag, ctx trained through the frozen text tower every step; SGD lr 1e-3 as train_ddp.py:104-109")
ap.add_argument("--clip_gemm_pieces", type=int, default=0, help="2 / 3: split-bf16 products for the frozen text tower's GEMMs")
ap.add_argument("--torch_adam", action="store_true", help="torch.optim.Adam / SGD instead of the flat one-launch optimizers")
ap.add_argument("--op_tail", action="store_true", help="criterion outside the module (op-by-op pool/head/loss tail) instead of forward(labels=y)")
ap.add_argument("--tower_in_graph", action="store_true", help="frozen prompts: run the text tower inside the captured step "
                "(fixed-shape form) instead of caching its output per note")
ap.add_argument("--graph", action="store_true", help="capture fwd+bwd+Adam of the trainable part in one hipGraph")
a = ap.parse_args()
dev = torch.device("cuda")
if a.coop:
    a.prompts = 10
args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL", num_classes=2,
                       learnablePrompt=int(a.coop), n_ctx=8, clinical_features=["f"] * 9, clip_gemm_pieces=a.clip_gemm_pieces, alignment_base="CI", model_CT="resnetMC3_18", clip_layers=a.clip_layers, cache_text=int(a.cache_text))
torch.manual_seed(1234)
model = get_model(args).to(dev).eval()      # eval: parity mode (dropout off), gradients still flow
x = syn.make_bags(1, a.bags, a.patches, 768).to(dev)
ids = syn.make_token_ids(2, a.bags, a.prompts).to(dev)
y = syn.make_labels(3, a.bags).to(dev)
if a.coop and a.graph:
    model.clinic_extractor.model.static_rows = True      # tower inside the graph: fixed-shape form (no host sync)
if a.coop and a.torch_adam:
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-7)
elif a.coop:
    from mil_amd.optim import FlatSGD
    opt = FlatSGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-7)
elif a.torch_adam:
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-5, weight_decay=1e-7,
                           capturable=a.graph)
else:
    from mil_amd.optim import FlatAdam
    opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-5, weight_decay=1e-7, counted=a.graph)
crit = torch.nn.BCELoss()

def fwd_loss(**kw):
    if a.op_tail:
        prob, _ = model([x], ids, **kw)
        return crit(prob, y)
    model([x], ids, labels=y, **kw)          # pool + head + BCE as one fused node; the loss is left in model.last_loss
    return model.last_loss

def step():
    loss = fwd_loss()
    opt.zero_grad(set_to_none=True)
    ops.backward(loss)
    opt.step()
    return loss

if a.graph:
    tfeat = None
    if a.tower_in_graph:
        model.clinic_extractor.model.static_rows = True
    elif not a.coop:
        with torch.no_grad():
            tfeat = model.clinic_extractor(ids)      # frozen tower: outside the graph (cached per note in training)
    def gstep():
        loss = fwd_loss(text_features=tfeat)
        ops.backward(loss)
        opt.step()
        return loss
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            gstep()
    torch.cuda.current_stream().wait_stream(side)
    # grads are None at capture time: backward() allocates them from the graph's pool and every replay
    # overwrites them in place (no per-parameter fill / accumulate kernels)
    opt.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = gstep()
    def step():
        g.replay()
        return static_loss
for _ in range(a.warmup): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): loss = step()
torch.cuda.synchronize(); el = time.perf_counter() - t0
print(json.dumps({"workload": f"fusion {a.bags} bags x {a.patches} x 768, {a.prompts} prompt(s), CLIP {a.clip_layers} layers",
                  "ms_per_step": round(el / a.steps * 1e3, 3), "bags_per_s": round(a.bags * a.steps / el, 1), "loss": float(loss.detach())}))
