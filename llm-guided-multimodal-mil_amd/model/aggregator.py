"""`aggregator(args)`: the drop-in boundary of the path (reference: model/aggregator.py:9-209).

Same constructor (`args.modality`, `args.model_pathology`, `args.model_CI`, `args.aggregator`,
`args.num_classes`), same `forward(x_list, x_CI)` and return tuples, same state_dict keys - but every
tensor op on the path runs through the HIP library:

    fc_pathology (GEMM + tanh) -> clinic_extractor (frozen CLIP text tower, cached per note)
    -> fc_CI2Pth (GEMM + tanh) -> TwoWayTransformer (text <-> patches) -> multi-modal bag [text | patches]
    -> aggregator (gated-attention MIL pool) -> fc + sigmoid.

Differences, all deliberate: a batch is B independent bags (the reference's B>1 ABMIL
degenerates to a sum); CT encoders / TransMIL / tabular-CI MLPs are out of scope and raise; the positional
table lives on the device (the reference re-uploads `pe[:, :N]` every forward, aggregator.py:160-190);
`last_logits` exposes the pre-sigmoid scores for the parity checks."""
import math
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import lifetime, ops
from ..bags import BagLayout
from .sam.transformer import TwoWayTransformer

EMBED = 512


def _arg(args, name, default):
    return getattr(args, name, default)


class aggregator(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        modality = list(args.modality)
        # 'CT' in modality: the CT ENCODERS (torchvision / MONAI 3-D backbones, model/dim3/*) stay outside the hot path, but
        # their output - the feature map [B, 512, 160, h, w] of aggregator.py:139-140 - is accepted as x_list[0], precomputed
        # (extractor_CT is the identity here), and everything downstream of it is built: map -> tokens, fc_CI2CT,
        # TwoWayTransformer_CT / _Both, the 4-segment multi-modal bag (:155-184).
        mk_twoway = lambda: TwoWayTransformer(args=args, depth=2, embedding_dim=EMBED, num_heads=8, mlp_dim=2048)  # noqa: E731
        self.fc_CI2CT = nn.Sequential(nn.Linear(EMBED, EMBED), nn.Tanh())                 # aggregator.py:44
        if "pathology" in modality:
            self.fc_pathology = nn.Sequential(nn.Linear(768, EMBED), nn.Tanh())           # :47
            if _arg(args, "model_pathology", "ABMIL") == "ABMIL":
                from .dim1 import ABMIL
                self.extractor_pathology = ABMIL(args, L=EMBED)                           # :50 (built, never called)
            elif args.model_pathology not in ("-", None):
                raise NotImplementedError(f"model_pathology={args.model_pathology}: only ABMIL is on the built path")
            self.TwoWayTransformer_Pth = mk_twoway()                                      # :58
        if "CT" in modality:
            self.TwoWayTransformer_CT = mk_twoway()                                       # :36
        self.fc_CI2Pth = nn.Sequential(nn.Linear(EMBED, EMBED), nn.Tanh())                # :66
        self.fc_CI = nn.Sequential(nn.Linear(EMBED, EMBED), nn.Tanh())                    # :68
        self.TwoWayTransformer_Both = mk_twoway()                                         # :70
        agg = _arg(args, "aggregator", "ABMIL")
        if agg == "ABMIL":
            from .dim1 import ABMIL
            self.aggregator = ABMIL(args, L=EMBED)                                        # :81
        elif agg != "-":
            raise NotImplementedError(f"aggregator={agg}: TransMIL needs the absent nystrom_attention package "
                                      "(parity unpinned, SURVEY.md section 8f); ABMIL_v2 is not on the built path")
        if _arg(args, "model_CI", "CLIP") == "CLIP":
            from .dim1 import CLIP
            self.clinic_extractor = CLIP(args)                                            # :122
        else:
            raise NotImplementedError(f"model_CI={args.model_CI}: only the CLIP text extractor is on the built path")
        self.prompt_embedding = nn.Parameter(torch.randn(1, EMBED))                       # :124 (unused upstream too)
        self.fc = nn.Sequential(nn.Dropout(0.25), nn.Linear(EMBED, args.num_classes))     # :128-131
        self._pe: Optional[torch.Tensor] = None
        self.last_logits: Optional[torch.Tensor] = None
        self.last_loss: Optional[torch.Tensor] = None
        self._labels: Optional[torch.Tensor] = None
        self._loss_scale: Optional[float] = None

    # ------------------------------------------------------------------ positional table (aggregator.py:99-106)
    def pe_rows(self, n: int, device) -> torch.Tensor:
        """First n rows of the sinusoidal table, device-resident, grown on demand.  Computed on the host with
        the reference's own expressions (bit-identical to `self.pe[:, :n]`) and uploaded once."""
        if self._pe is None or self._pe.shape[0] < n or self._pe.device != device:
            rows = max(n, 2048, 0 if self._pe is None else 2 * self._pe.shape[0])
            pe = torch.zeros((rows, EMBED))
            position = torch.arange(0, rows).unsqueeze(1)
            div_term = torch.exp(torch.arange(0, EMBED, 2, dtype=torch.float) * -(math.log(10000.0) / EMBED))
            pe[:, 0::2] = torch.sin(position.float() * div_term)
            pe[:, 1::2] = torch.cos(position.float() * div_term)
            self._pe = pe.to(device)
        return lifetime.note(self._pe)            # a captured graph keeps the table it saw when a larger one replaces it

    def _lin_tanh(self, seq: nn.Sequential, x, rows_dev=None):
        return ops.linear_act(x, seq[0].weight, seq[0].bias, "tanh", rows_dev=rows_dev)

    def _head(self, M):
        if self.training:
            M = F.dropout(M, 0.25, True)
        p, z = ops.head_sigmoid(M, self.fc[1].weight, self.fc[1].bias)
        self.last_logits = z
        return p

    def _pool_head(self, x0, layout):
        """Multi-modal bag rows -> prob.  With labels handed to forward() (the training loop's `criterion(prob, y)` of
        train_ddp.py:323-324 moved inside) and an ABMIL aggregator, pool + Dropout(.25) + fc + sigmoid + BCE / CE run as one
        fused node (ops.gated_pool_head_loss) and the loss is left in `self.last_loss`; otherwise op by op."""
        y = self._labels
        C = self.args.num_classes
        if y is not None and hasattr(getattr(self, "aggregator", None), "flat_head_loss") and \
                self.aggregator.can_fuse_head(x0, C):
            ce = C > 2                                     # train_ddp.py:95-98: CrossEntropyLoss on the sigmoid outputs
            scale = self._loss_scale if self._loss_scale is not None else 1.0 / (layout.B * (1 if ce else C))
            loss, p, z = self.aggregator.flat_head_loss(x0, layout, self.fc[1].weight, self.fc[1].bias, y, scale,
                                                        1 if ce else 0, head_train=self.training)
            self.last_logits, self.last_loss = z, loss
            return p
        M = self.aggregator.flat(x0, layout) if hasattr(self, "aggregator") else x0
        p = self._head(M)
        if y is not None:
            crit = torch.nn.CrossEntropyLoss() if C > 2 else torch.nn.BCELoss()
            self.last_loss = crit(p, y) * (1.0 if self._loss_scale is None else
                                           self._loss_scale * layout.B * (1 if C > 2 else C))
        return p

    # ------------------------------------------------------------------ CT branches (aggregator.py:155-184)
    def _forward_ct(self, x_list, t, lengths):
        """x_list[0] = the CT encoder's feature map [B, 512, 160, h, w] (precomputed; or tokens [B, D, 512]);
        with 'pathology' also x_list[1] = [B, N, 768].  Returns (prob, x_CT2CI[, x_Pth2CI]) as :202-205."""
        B, P, _ = t.shape
        dev = t.device
        ct = x_list[0]
        if ct.dim() == 5:
            ct_rows, D = ops.ct_map_tokens(ct, _arg(self.args, "model_CT", "resnetMC3_18"))     # sam/transformer.py:86-98
        else:
            D = ct.shape[1]
            ct_rows = ct.reshape(B * D, EMBED).contiguous()
        tflat = t.reshape(B * P, EMBED)
        point_ct = self._lin_tanh(self.fc_CI2CT, tflat)                                       # :160 / :179
        if "pathology" not in self.args.modality:
            q, k = self.TwoWayTransformer_CT.flat(ct_rows, point_ct, self.pe_rows(D, dev), [D] * B, [P] * B,
                                                  keys_tail_rows=B * P)                       # :179
            x0 = ops.append_rows(k, q, tail_reserved=True)                                    # :184
            layout = BagLayout.two_segment([D] * B, [P] * B, dev)
            return self._pool_head(x0, layout), q.view(B, P, EMBED)                           # :204-205
        x = x_list[1]
        if x.dim() == 2:
            x = x.unsqueeze(0)
        N = x.shape[1]
        if lengths is None:
            n_len = [N] * B
            flat = x.reshape(B * N, x.shape[2])
        else:
            n_len = [int(v) for v in lengths]
            flat = torch.cat([x[b, :n] for b, n in enumerate(n_len)], 0)
        tw = self.TwoWayTransformer_Both                                                      # :160,168: one module, twice
        q_ct, k_ct = tw.flat(ct_rows, point_ct, self.pe_rows(D, dev), [D] * B, [P] * B)
        xi = self._lin_tanh(self.fc_pathology, flat)                                          # :141
        point_p = self._lin_tanh(self.fc_CI2Pth, tflat)
        extra = B * (2 * P + D)
        q_p, k_p = tw.flat(xi, point_p, self.pe_rows(max(n_len), dev), n_len, [P] * B, keys_tail_rows=extra)
        # multi-modal bag (:173) = [x_CT2CI | x_CI2CT | x_Pth2CI | x_CI2Pth] per patient.  Rows are kept as
        # [all patch tokens | CT-side text tokens | CT tokens | pathology-side text tokens]: the big patch block stays where
        # the last LayerNorm wrote it, the small blocks are appended, the tile map says which rows form a bag.
        x0 = ops.append_rows(k_p, torch.cat([q_ct, k_ct, q_p], 0), tail_reserved=True)
        layout = BagLayout.multi_segment([n_len, [P] * B, [D] * B, [P] * B], dev)
        return self._pool_head(x0, layout), q_ct.view(B, P, EMBED), q_p.view(B, P, EMBED)     # :198-200,202-203

    # ------------------------------------------------------------------ capacity-bucket form of the pathology branch
    def _forward_bucket(self, x, t, bucket):
        """The pathology + text branch (aggregator.py:186-192,207) on a segments.FusionBucket: x [cap, 768] holds the
        patches of the B bags packed from row 0 (zero rows behind), the true lengths are on the device (bucket.len_dev),
        and every launch below depends on (cap, B) only - so ONE captured graph per bucket serves every bag length the
        authors' loader produces (dataset.py:366-393; fusion_step.RaggedFusionStepper).  One text token per bag."""
        B, P, _ = t.shape
        if x.dim() != 2 or x.shape[0] != bucket.cap or B != bucket.B or P != bucket.P:
            raise ValueError(f"bucket of {bucket.B} bags x {bucket.cap} rows x {bucket.P} token(s): got x {tuple(x.shape)}, "
                             f"text {tuple(t.shape)}")
        bucket.refresh()                                                                   # maps from len_dev, on the device
        xi = self._lin_tanh(self.fc_pathology, x, rows_dev=bucket.rows_dev)                # :149 (tiles behind the true rows: zeros)
        point = self._lin_tanh(self.fc_CI2Pth, t.reshape(B * P, EMBED))                    # :190
        q, k = self.TwoWayTransformer_Pth.flat(xi, point, self.pe_rows(bucket.cap, xi.device), None, None,
                                               keys_tail_rows=B * P, segs=(bucket.s_tt, bucket.s_ti, bucket.s_it))
        x0 = ops.append_rows(k, q, tail_reserved=True)                                     # :192 (no concat copy)
        return self._pool_head(x0, bucket.layout), q.view(B, P, EMBED)                     # :198-200,207

    def _forward_ct_bucket(self, x_list, t, bucket):
        """The CT + pathology branch (aggregator.py:155-173,202-203) on a segments.FusionBucket whose tail is [P, D, P]:
        x_list = [CT feature map [B, 512, 160, h, w] (or tokens [B, D, 512]), patches [cap, 768] packed from row 0].  The CT
        side is static (D tokens per bag) and runs on host-built maps; the pathology side and the 4-segment multi-modal bag
        follow the device-side bag lengths.  This is the authors' own run (run_train.sh:81: `--modality ['CT','pathology']
        --CI_prompt_version single --learnablePrompt 0 --loss_point CT-Pth-Last`, one bag per GPU)."""
        B, P, _ = t.shape
        dev = t.device
        ct, x = x_list[0], x_list[1]
        if ct.dim() == 5:
            ct_rows, D = ops.ct_map_tokens(ct, _arg(self.args, "model_CT", "resnetMC3_18"))     # sam/transformer.py:86-98
        else:
            D = ct.shape[1]
            ct_rows = ct.reshape(B * D, EMBED).contiguous()
        if x.dim() != 2 or x.shape[0] != bucket.cap or B != bucket.B or list(bucket.tail) != [P, D, P]:
            raise ValueError(f"bucket of {bucket.B} bags x {bucket.cap} rows with tail {bucket.tail}: got x {tuple(x.shape)}, "
                             f"text {tuple(t.shape)}, {D} CT tokens per bag")
        bucket.refresh()
        tflat = t.reshape(B * P, EMBED)
        tw = self.TwoWayTransformer_Both                                                      # :160,168: one module, twice
        q_ct, k_ct = tw.flat(ct_rows, self._lin_tanh(self.fc_CI2CT, tflat), self.pe_rows(D, dev), [D] * B, [P] * B)
        xi = self._lin_tanh(self.fc_pathology, x, rows_dev=bucket.rows_dev)                   # :141
        q_p, k_p = tw.flat(xi, self._lin_tanh(self.fc_CI2Pth, tflat), self.pe_rows(bucket.cap, dev), None, None,
                           keys_tail_rows=bucket.tail_rows, segs=(bucket.s_tt, bucket.s_ti, bucket.s_it))
        x0 = ops.append_rows(k_p, torch.cat([q_ct, k_ct, q_p], 0), tail_reserved=True)        # :173 (no concat of the patches)
        return self._pool_head(x0, bucket.layout), q_ct.view(B, P, EMBED), q_p.view(B, P, EMBED)

    # ------------------------------------------------------------------ forward (aggregator.py:134-209)
    def forward(self, x_list: List[torch.Tensor], x_CI: torch.Tensor, lengths: Optional[List[int]] = None,
                text_features: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                loss_scale: Optional[float] = None, bucket=None):
        """The shipped module's return tuples (aggregator.py:202-209), which test_ddp.py:220 consumes.  With
        `args.train_contract` set, the tuple the reference's TRAINING loop unpacks instead (train_ddp.py:300,305-316; no
        shipped module provides it): CT + pathology -> ([out, out, out], [x_CT2CI, x_Pth2CI], None) - one head, so the three
        outputs of loss_point 'CT-Pth-Last' are the same tensor -, a single modality -> (out, token, None), text only ->
        (out, None)."""
        out = self._forward(x_list, x_CI, lengths, text_features, labels, loss_scale, bucket)
        if not _arg(self.args, "train_contract", 0):
            return out
        if not isinstance(out, tuple):
            return out, None
        if len(out) == 3:
            return [out[0], out[0], out[0]], [out[1], out[2]], None
        return out[0], out[1], None

    def _forward(self, x_list: List[torch.Tensor], x_CI: torch.Tensor, lengths: Optional[List[int]] = None,
                 text_features: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                 loss_scale: Optional[float] = None, bucket=None):
        """x_list = [x_pathology [B, N, 768]] (or [] for CI only); x_CI int64 [B, P, ctx] token ids.
        `lengths` (optional) gives the true patch count of each zero-padded bag (dataset.py:386-391 pads to a
        fixed length when batch > 1); padded rows are then dropped instead of being attended to."""
        # labels [B, C] float one-hot (optional, an extension of the reference signature): the criterion of
        # train_ddp.py:323-324 evaluated inside, result in `self.last_loss` (mean loss unless loss_scale is given);
        # the return tuple is unchanged
        self._labels, self._loss_scale, self.last_loss = labels, loss_scale, None
        modality = self.args.modality
        # text_features [B, P, 512] (optional): embeddings of the frozen text tower computed earlier by
        # `self.clinic_extractor(x_CI)`; lets a captured hipGraph replay the trainable part only
        t = text_features if text_features is not None else self.clinic_extractor(x_CI)   # :151  [B, P, 512]
        B, P, _ = t.shape
        if "CT" in modality and "pathology" in modality and bucket is not None:
            return self._forward_ct_bucket(x_list, t, bucket)
        if "CT" in modality:
            return self._forward_ct(x_list, t, lengths)
        if "pathology" in modality and bucket is not None:
            return self._forward_bucket(x_list[0], t, bucket)
        if "pathology" in modality:
            x = x_list[0]
            if x.dim() == 2:
                x = x.unsqueeze(0)
            N = x.shape[1]
            if lengths is None:
                n_len = [N] * B
                flat = x.reshape(B * N, x.shape[2])
            else:
                n_len = [int(v) for v in lengths]
                flat = torch.cat([x[b, :n] for b, n in enumerate(n_len)], 0)
            xi = self._lin_tanh(self.fc_pathology, flat)                                  # :149
            point = self._lin_tanh(self.fc_CI2Pth, t.reshape(B * P, EMBED))               # :190
            q, k = self.TwoWayTransformer_Pth.flat(xi, point, self.pe_rows(max(n_len), xi.device), n_len, [P] * B,
                                                   keys_tail_rows=B * P)
            # multi-modal bag per patient = its text tokens + its patch tokens (:192).  Rows are kept as
            # [all patches | all tokens] (the patch tokens stay where the last LayerNorm wrote them, only the few
            # token rows are appended) and the tile map tells the pool kernels which rows belong to which bag;
            # attention pooling does not depend on the order of a bag's rows.
            x0 = ops.append_rows(k, q, tail_reserved=True)     # k comes from the last block's norm4 with keys_tail_rows
            layout = BagLayout.two_segment(n_len, [P] * B, x0.device)
            return self._pool_head(x0, layout), q.view(B, P, EMBED)                       # :198-200,207
        if "CI" in modality:
            x0 = self._lin_tanh(self.fc_CI, t.reshape(B * P, EMBED))                      # :195
            return self._pool_head(x0, BagLayout.uniform(B, P, x0.device))                # :198-200,209
        raise NotImplementedError(f"modality {modality}")
