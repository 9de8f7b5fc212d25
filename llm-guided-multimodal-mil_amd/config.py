"""Flag system of the entry points.  Flag names and defaults follow the reference's config.py:10-142 for
everything the hot path reads (modality / model_* / aggregator / num_classes / learnablePrompt / n_ctx /
clinical_features / lr / b1 / b2 / batch_size / seed / dist_* ...), list flags are parsed with
ast.literal_eval as upstream (config.py:4-8).  Hospital-data flags are dropped; synthetic-data flags are new."""
import argparse
import ast


def arg_as_list(s):
    v = ast.literal_eval(s)
    if not isinstance(v, list):
        raise argparse.ArgumentTypeError(f'Argument "{s}" is not a list')
    return v


def create_arg_parser(argv=None):
    p = argparse.ArgumentParser(description="MI355X-native LLM-guided multi-modal MIL (hot path)")
    # ---- model (reference names; defaults chosen so the built path runs: upstream defaults select CT + TransMIL)
    p.add_argument("--modality", default=["pathology"], type=arg_as_list, help="subset of ['CT', 'pathology', 'CI'] ('CT': the encoder's feature map is a synthetic input)")
    p.add_argument("--alignment_base", default="CI", type=str)
    p.add_argument("--model_CT", default="resnetMC3_18", type=str)
    p.add_argument("--model_pathology", default="ABMIL", type=str)
    p.add_argument("--model_CI", default="CLIP", type=str)
    p.add_argument("--aggregator", default="ABMIL", type=str)
    p.add_argument("--CI_prompt_version", default="single", type=str, help="single (1 note) | devided (10 prompts)")
    p.add_argument("--clinical_features", type=arg_as_list,
                   default=["sex", "age", "sm", "locationcancer", "cancerimaging", "cancerimagingT", "cancerimagingN",
                            "cancerimagingM", "classification_cancer"])
    p.add_argument("--learnablePrompt", default=0, type=int)
    p.add_argument("--n_ctx", default=8, type=int)
    p.add_argument("--prompt_len", default=0, type=int)
    p.add_argument("--num_classes", type=int, default=2)
    p.add_argument("--variant", default="fusion", choices=["fusion", "image_only"],
                   help="fusion: model/aggregator.py path; image_only: model/aggregator_clip.py path")
    # ---- optimisation (train_ddp.py:104-118 forces lr 1e-5 for Adam with 2 classes)
    p.add_argument("--start_epoch", type=int, default=0)
    p.add_argument("--n_epochs", type=int, default=2)
    p.add_argument("--resume", default="", type=str)
    p.add_argument("--lr", type=float, default=1e-5)
    p.add_argument("--loss", type=str, default="BCE", help="'BCE'; a name containing 'textCosSim' adds "
                   "CosineEmbeddingLoss(x_CT2CI, x_Pth2CI, 1) when both tokens exist (train_ddp.py:102,325-329)")
    p.add_argument("--cache_text", type=int, default=0, help="1: frozen text tower (learnablePrompt 0) - keep every note's "
                   "embedding after its first encode_text instead of recomputing it each step as model/dim1/CLIP.py:71-75 does "
                   "(same values: the tower is frozen)")
    p.add_argument("--train_contract", type=int, default=0, help="1: the module returns the tuple the reference's training "
                   "loop unpacks, ([out, out, out], [CT2CI, Pth2CI], None) (train_ddp.py:300), instead of the shipped "
                   "module's (aggregator.py:202-209)")
    p.add_argument("--loss_point", type=str, default="Last")
    p.add_argument("--schedule", default=[500], nargs="*", type=int)
    p.add_argument("--cos", action="store_true")
    p.add_argument("--b1", type=float, default=0.9)
    p.add_argument("--b2", type=float, default=0.999)
    p.add_argument("--seed", default=1234, type=int)
    p.add_argument("--batch_size", default=8, type=int, help="global mini-batch (divided by the number of GPUs)")
    p.add_argument("--iter_per_epoch", type=int, default=20)
    p.add_argument("--save_dir", type=str, default="")
    p.add_argument("--test_pth", type=str, default=None)
    p.add_argument("--best_thres", type=float, default=0.5)
    # ---- distributed (one process per GPU; torchrun env or mp.spawn as upstream train_ddp.py:622-624)
    p.add_argument("--gpu", default="0", type=str, help="comma-separated GPU ids when spawning")
    p.add_argument("--multiprocessing_distributed", action="store_true")
    p.add_argument("--dist_url", type=str, default="tcp://127.0.0.1:4444")
    p.add_argument("--dist_backend", type=str, default="nccl")
    p.add_argument("--world_size", type=int, default=1)
    p.add_argument("--rank", type=int, default=0)
    # ---- on-disk cohort (reference: --path_data_pathology, config.py; dataset.py:366-393).  A directory of
    # <patientid>.npy patch-feature bags [n, F] fp32 plus a JSON index {"id": {"label", "kind", "ids"}} in place of the
    # private Excel sheets; unset = synthetic bags.
    p.add_argument("--path_data_pathology", type=str, default="", help="directory of <patientid>.npy bags (dataset.py:367)")
    p.add_argument("--index_json", type=str, default="", help="cohort index; default <path_data_pathology>/index.json")
    p.add_argument("--augmentation", type=int, default=1, help="train-time patch drop (dataset.py:374-381)")
    p.add_argument("--resident_cohort", type=int, default=1,
                   help="with --hip_graph 1: load every bag ONCE into HBM (cohort.DeviceCohort), draw the per-epoch patch drop "
                        "on the device and feed each step by one gather launch; 0 = the host pipeline per step (np.load, "
                        "random.sample drop, zero-pad, pageable copy).  Falls back to 0 by itself when the cohort does not fit")
    p.add_argument("--patch_keep", type=float, default=1.0, help="synthetic cohorts: keep fraction of the per-epoch patch drop "
                   "(on-disk cohorts use the reference's 0.9 / 0.8 per bag kind)")
    # ---- synthetic data (no hospital data offline)
    p.add_argument("--synthetic", default=[1024, 768, 64], type=arg_as_list,
                   help="[patches per bag, patch feature dim, bags in the synthetic cohort]")
    p.add_argument("--ragged", action="store_true", help="draw a different patch count per bag")
    p.add_argument("--clip_layers", type=int, default=12)
    p.add_argument("--fused_step", action="store_true", help="image_only: fused trainer instead of autograd + DDP")
    p.add_argument("--no_dropout", action="store_true", help="--fused_step: eval-mode arithmetic (no dropout) while training; "
                   "the default is a true model.train() step with in-kernel dropout masks")
    p.add_argument("--clip_gemm_pieces", type=int, default=0, choices=[0, 2, 3],
                   help="frozen CLIP text tower: 0 = fp32 MFMA GEMMs (parity path); 2 / 3 = split-bf16 products "
                        "(3 / 6 cross terms: ~3e-6 / fp32-level relative error, 2.3x / 1.45x the fp32 GEMM rate)")
    p.add_argument("--hip_graph", type=int, default=0, help="autograd path: capture forward + backward of a repeating "
                   "batch shape in a hipGraph and replay it (graph_step.py); meant for the one-bag-per-GPU regime")
    p.add_argument("--flat_adam", type=int, default=1, help="autograd path: parameters in one flat buffer, one gradient "
                   "all-reduce and one Adam launch per step (optim.FlatAdam); 0 = torch DDP + torch.optim.Adam")
    return p.parse_args(argv)
