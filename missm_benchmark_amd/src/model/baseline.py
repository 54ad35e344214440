"""Drop-in for the reference's ``src.model.baseline`` (reference: src/model/baseline.py:8,27-61,421-453).

``finetune_model(args, output_dims, encoder_model)`` keeps the reference constructor / attributes / forward:
``forward(data, missing_index) = fusion(encoder(data), missing_index)``; ``args`` needs ``modality_types, feature_dims,
fusion_dim, dropout_prob, fusion_type``.  ``'concat'`` (zero / mean / median imputation, reference :64-90), ``'retrieval'``
(:164-180), ``'intra_attention'`` (:183-205), ``'dedicated_dnn'`` (:333-353), ``'regression'`` (:93-161) and the distillation network (``'Distill_tea'`` / ``'MTD_stu'`` /
``'KL_stu'``, :356-380; ``'self_distill'``, :383-418) are available too.  The primary head (``fusion_type == 'sum'``: per-modality projection, zeroing of
the missing modality's rows, sum, LayerNorm, MLP head) runs on the HIP kernels; its parameters keep the reference's
state-dict keys (``fusion.modal_proj.<m>.{weight,bias}``, ``fusion.norm.*``, ``fusion.head.head.{0,3}.*``).

``missing_type_index`` extends the reference's 4 codes with depth/thermal (codes 0-4 unchanged) - the reference's towers
for those modalities exist but its fusion table does not list them.
"""
from __future__ import annotations

import torch
from torch import nn

from ... import nn as hnn
from ...languagebind import LanguageBind, LanguageBindImageTokenizer, to_device, transform_dict  # noqa: F401  (reference re-exports)

missing_type_index = {"language": 1, "video": 2, "audio": 3, "image": 4, "depth": 5, "thermal": 6}


class Head(nn.Module):
    """Linear -> ReLU -> Dropout -> Linear kept under ``head.{0,3}`` like the reference's nn.Sequential."""

    def __init__(self, args, input_dims, output_dims):
        super().__init__()
        stages = nn.Module()
        stages.add_module("0", hnn.HipLinear(input_dims, args.fusion_dim, relu=True))   # ReLU fused into the launch
        stages.add_module("2", hnn.HipDropout(args.dropout_prob))
        stages.add_module("3", hnn.HipLinear(args.fusion_dim, output_dims))
        self.head = stages

    def forward(self, inputs):
        m = self.head._modules
        return m["3"](m["2"](m["0"](inputs)))


class _FusionBase(nn.Module):
    def __init__(self, args, output_dims, head_in):
        super().__init__()
        self.modality_types = list(args.modality_types)
        unknown = [m for m in self.modality_types if m not in missing_type_index]
        if unknown:
            raise KeyError(f"no missing-type code for {unknown}")
        self.modal_proj = nn.ModuleDict({m: hnn.HipLinear(args.feature_dims, args.fusion_dim) for m in self.modality_types})
        self.norm = hnn.HipLayerNorm(head_in)
        self.head = Head(args, head_in, output_dims)

    def _codes(self):
        return [missing_type_index[m] for m in self.modality_types]


class modal_sum(_FusionBase):
    """Arithmetic composition: sum of the present modalities' projections (reference :43-61)."""

    def __init__(self, args, output_dims):
        super().__init__(args, output_dims, args.fusion_dim)

    def forward(self, batch, missing_index):
        z = hnn.fused_modal_sum(missing_index, self._codes(), [batch[m] for m in self.modality_types],
                                [self.modal_proj[m] for m in self.modality_types])
        return self.head(self.norm(z))


class modal_concat(_FusionBase):
    """Zero / mean / median composition (reference :64-90): a missing modality's embedding rows are replaced by the
    ``statistics_<modal>`` buffer (zeros until ``set_statistics``), every modality is projected and the projections are
    concatenated."""

    def __init__(self, args, output_dims):
        super().__init__(args, output_dims, args.fusion_dim * len(args.modality_types))
        for m in self.modality_types:
            self.register_buffer(f"statistics_{m}", torch.zeros(args.feature_dims, dtype=torch.float))

    def forward(self, batch, missing_index):
        return self.head(self.norm(hnn.fused_modal_concat(
            missing_index, self._codes(), [batch[m] for m in self.modality_types], [self.modal_proj[m] for m in self.modality_types],
            [self.get_buffer(f"statistics_{m}") for m in self.modality_types])))

    def set_statistics(self, statistics, modality_types):
        """reference :88-90 (called from test.py:115 with per-modality mean / median embeddings of the training set)"""
        for m in modality_types:
            dev = self.modal_proj[m].weight.device
            self.register_buffer(f"statistics_{m}", torch.as_tensor(statistics[m], dtype=torch.float, device=dev).contiguous())


class modal_concat_full(_FusionBase):
    """Retrieval-based composition (reference :164-180): missing embeddings were already filled upstream by retrieval, so
    the head is projection + concatenation with no masking."""

    def __init__(self, args, output_dims):
        super().__init__(args, output_dims, args.fusion_dim * len(args.modality_types))

    def forward(self, batch, missing_index):
        return self.head(self.norm(hnn.fused_modal_concat(
            missing_index, self._codes(), [batch[m] for m in self.modality_types], [self.modal_proj[m] for m in self.modality_types])))


class modal_intra_channel_attention(_FusionBase):
    """Intra-modality attention (reference :183-205): every projected modality is re-weighted channel-wise by a gate computed
    from it and a learned fusion representation; missing rows are zeroed after the gate; the modalities are summed."""

    def __init__(self, args, output_dims):
        super().__init__(args, output_dims, args.fusion_dim)
        F = args.fusion_dim
        if F % 16:
            raise ValueError("fusion_dim must be a multiple of 16 (the gate MLP is fusion_dim // 4 wide)")
        self.fusion_representation = nn.Parameter(torch.randn(1, F))
        ca = nn.Module()                                     # keeps the reference's nn.Sequential keys: channel_attention.{0,2}.*
        ca.add_module("0", hnn.HipLinear(2 * F, F // 4, relu=True))
        ca.add_module("2", hnn.HipLinear(F // 4, F))
        self.channel_attention = ca

    def forward(self, batch, missing_index):
        m = self.channel_attention._modules
        z = hnn.fused_intra_attention(missing_index, self._codes(), [batch[k] for k in self.modality_types],
                                      [self.modal_proj[k] for k in self.modality_types], self.fusion_representation, m["0"], m["2"])
        return self.head(self.norm(z))


class _MultiheadAttentionParams(nn.Module):
    """the parameters of ``nn.MultiheadAttention(embed_dim, num_heads)`` under its state-dict keys (``in_proj_weight``,
    ``in_proj_bias``, ``out_proj.{weight,bias}``) and its initialisation (xavier-uniform packed in-projection, zero biases)"""

    def __init__(self, embed_dim: int, num_heads: int):
        super().__init__()
        if embed_dim % num_heads or (embed_dim // num_heads) % 8:
            raise ValueError("fusion_dim / num_heads must be a multiple of 8")
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = hnn.HipLinear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        with torch.no_grad():
            self.out_proj.bias.zero_()


class modal_inter_attention(_FusionBase):
    """Inter-modality attention (reference :207-236): the projected modalities are tokens, a learned query token attends over them
    (4 heads), a missing modality is removed from the keys; LayerNorm; Head."""

    def __init__(self, args, output_dims):
        super().__init__(args, output_dims, args.fusion_dim)
        self.query_token = nn.Parameter(torch.randn(1, 1, args.fusion_dim))
        self.attn = _MultiheadAttentionParams(args.fusion_dim, 4)

    def forward(self, batch, missing_index):
        z = hnn.fused_inter_attention(missing_index, self._codes(), [batch[m] for m in self.modality_types],
                                      [self.modal_proj[m] for m in self.modality_types], self.query_token, self.attn.in_proj_weight,
                                      self.attn.in_proj_bias, self.attn.out_proj, self.attn.num_heads)
        return self.head(self.norm(z))


class fusion_gcn(hnn.HipFusionGCN):
    """reference :11-24 (two SuperGAT layers with a GELU between them) on the per-sample modality graphs; see nn.HipSuperGATConv"""


def _node_ok(missing_index, modality_types):
    """bool [B, M]: modality i of sample b is present (the 'missing_modal_index' ones of the reference, :254-258)"""
    return torch.stack([missing_index != missing_type_index[m] for m in modality_types], dim=1).contiguous()


class modal_graph_fusion(_FusionBase):
    """Graph fusion (reference :240-277): every projected modality is a node, edges join the modalities that are present (plus self
    loops), two SuperGAT layers, mean over the nodes, LayerNorm, Head.  PARITY UNPINNED: torch_geometric's SuperGATConv is
    restated (nn.HipSuperGATConv), no reference fixture can be produced here."""

    def __init__(self, args, output_dims):
        super().__init__(args, output_dims, args.fusion_dim)
        if args.fusion_dim != 256:
            raise ValueError("graph_fusion: the reference's fusion_gcn() is built for fusion_dim == 256 (src/model/baseline.py:251)")
        self.gcn = fusion_gcn()

    def forward(self, batch, missing_index):
        mt = self.modality_types
        x = hnn.fused_modal_concat(missing_index, [-1] * len(mt), [batch[m] for m in mt], [self.modal_proj[m] for m in mt])   # [B, M * D], no masking
        B, M = x.shape[0], len(mt)
        out = self.gcn(x.view(B * M, -1), _node_ok(missing_index, mt))
        return self.head(self.norm(hnn.node_mean(out, B, M)))


class modal_unified_graph(nn.Module):
    """Unified GNN (reference :280-331): a completion network over the raw embeddings predicts the missing modality's embedding from
    the present ones, the filled-in embeddings go through the fusion network on the complete graph; mean over nodes, LayerNorm, Head.
    PARITY UNPINNED (SuperGATConv restated)."""

    def __init__(self, args, output_dims):
        super().__init__()
        self.modality_types = list(args.modality_types)
        if args.feature_dims != 768 or args.fusion_dim != 256:
            raise ValueError("unified_graph: the reference's networks are built for feature_dims == 768 and fusion_dim == 256 (:288-289)")
        self.norm = hnn.HipLayerNorm(args.fusion_dim)
        self.head = Head(args, args.fusion_dim, output_dims)
        self.complete_gcn = fusion_gcn(in_channels=768, hidden_dim=384, output_dim=768)
        self.fusion_gcn = fusion_gcn(in_channels=768)

    def forward(self, batch, missing_index):
        mt = self.modality_types
        B, M, D = batch[mt[0]].shape[0], len(mt), batch[mt[0]].shape[1]
        ok = _node_ok(missing_index, mt)
        feats = hnn.stack_blocks([batch[m] for m in mt])                                   # [B, M * D]
        completed = self.complete_gcn(feats.view(B * M, D), ok).view(B, M * D)
        filled = [hnn.fill_missing(batch[m], completed[:, i * D:(i + 1) * D], missing_index, missing_type_index[m]) for i, m in enumerate(mt)]
        allf = hnn.stack_blocks(filled)
        out = self.fusion_gcn(allf.view(B * M, D), torch.ones_like(ok))
        return self.head(self.norm(hnn.node_mean(out, B, M)))


class modal_dedicated_dnn(nn.Module):
    """Dedicated training (reference :333-353): one network over all modalities and one per missing-modality case over the
    remaining ones; every sample is routed to the network of its missing code."""

    def __init__(self, args, output_dims):
        super().__init__()
        self.modality_types = list(args.modality_types)
        M = len(self.modality_types)
        if M < 2:
            raise ValueError("dedicated_dnn needs at least two modalities")
        nets = {m: hnn.HipLinear(args.feature_dims * (M - 1), args.fusion_dim) for m in self.modality_types}
        nets["full"] = hnn.HipLinear(args.feature_dims * M, args.fusion_dim)
        self.dedicated_dnn = nn.ModuleDict(nets)
        self.norm = hnn.HipLayerNorm(args.fusion_dim)
        self.head = Head(args, args.fusion_dim, output_dims)

    def forward(self, batch, missing_index):
        # routing table (index bookkeeping on int64 codes, like the reference's boolean masks): 0 = full, m + 1 = modality m missing
        sel = torch.zeros_like(missing_index)
        for i, m in enumerate(self.modality_types):
            sel = torch.where(missing_index == missing_type_index[m], torch.full_like(sel, i + 1), sel)
        z = hnn.fused_dedicated_dnn(sel, [batch[m] for m in self.modality_types], self.dedicated_dnn["full"],
                                    [self.dedicated_dnn[m] for m in self.modality_types])
        return self.head(self.norm(z))


class modal_regression(_FusionBase):
    """Direct-to-task representation generation (reference :93-161): a missing modality's projection is predicted from the
    other modalities by the ``cross_modal_regressors`` (``<source>_to_<target>``) and averaged."""

    def __init__(self, args, output_dims):
        super().__init__(args, output_dims, args.fusion_dim * len(args.modality_types))
        if len(self.modality_types) < 2:
            raise ValueError("regression needs at least two modalities")
        self.cross_modal_regressors = nn.ModuleDict({f"{s}_to_{t}": hnn.HipLinear(args.feature_dims, args.fusion_dim)
                                                     for s in self.modality_types for t in self.modality_types if s != t})

    def forward(self, batch, missing_index):
        mt = self.modality_types
        regs = [[None if s == t else self.cross_modal_regressors[f"{s}_to_{t}"] for s in mt] for t in mt]
        z = hnn.fused_regression_concat(missing_index, self._codes(), [batch[m] for m in mt], [self.modal_proj[m] for m in mt], regs)
        return self.head(self.norm(z))


class modal_distillation(nn.Module):
    """Teacher / student network of the distillation baselines (reference :356-380, fusion types Distill_tea, MTD_stu, KL_stu):
    missing modalities zeroed, embeddings concatenated, Linear-ReLU-Linear, LayerNorm, Head; returns ``(features, logits)``."""

    def __init__(self, args, output_dims):
        super().__init__()
        self.modality_types = list(args.modality_types)
        seq = nn.Module()                                    # nn.Sequential keys of the reference: modal_proj.{0,2}.*
        seq.add_module("0", hnn.HipLinear(args.feature_dims * len(self.modality_types), args.fusion_dim, relu=True))
        seq.add_module("2", hnn.HipLinear(args.fusion_dim, args.fusion_dim))
        self.modal_proj = seq
        self.norm = hnn.HipLayerNorm(args.fusion_dim)
        self.head = Head(args, args.fusion_dim, output_dims)

    def forward(self, batch, missing_index):
        features = hnn.masked_concat(missing_index, [missing_type_index[m] for m in self.modality_types],
                                     [batch[m] for m in self.modality_types])
        m = self.modal_proj._modules
        inputs = m["2"](m["0"](features))
        return features, self.head(self.norm(inputs))


class modal_self_distillation(modal_distillation):
    """Self distillation (reference :383-418): the same network; in training it also returns, per modality, the student
    features computed from that modality alone (the other blocks of the concatenated row are zero) next to the teacher features
    of the full row: ``(missing_mask, stu_features, tea_features, logits)``.  In eval: logits only."""

    def _project(self, features):
        m = self.modal_proj._modules
        return m["2"](m["0"](features))

    def forward(self, batch, missing_index):
        mt = self.modality_types
        codes = [missing_type_index[m] for m in mt]
        xs = [batch[m] for m in mt]
        tea = self._project(hnn.masked_concat(missing_index, codes, xs))
        if not self.training:
            return self.head(self.norm(tea))
        never = -1                                  # a code no sample carries: blocks of the other modalities are all-zero inputs
        stu, masks = [], []
        for i, m in enumerate(mt):
            zeros = [torch.zeros_like(x) for x in xs]
            only_i = hnn.masked_concat(missing_index, [codes[j] if j == i else never for j in range(len(mt))],
                                       [xs[j] if j == i else zeros[j] for j in range(len(mt))])
            stu.append(self._project(only_i))
            masks.append(missing_index != codes[i])
        return masks, stu, tea, self.head(self.norm(tea))


_NOT_YET = ()


class finetune_model(nn.Module):
    def __init__(self, args, output_dims, encoder_model):
        super().__init__()
        self.encoder = encoder_model
        self.fusion_type = args.fusion_type
        if args.fusion_type == "sum":
            self.fusion = modal_sum(args, output_dims)
        elif args.fusion_type == "concat":
            self.fusion = modal_concat(args, output_dims)
        elif args.fusion_type == "retrieval":
            self.fusion = modal_concat_full(args, output_dims)
        elif args.fusion_type == "intra_attention":
            self.fusion = modal_intra_channel_attention(args, output_dims)
        elif args.fusion_type == "inter_attention":
            self.fusion = modal_inter_attention(args, output_dims)
        elif args.fusion_type == "graph_fusion":
            self.fusion = modal_graph_fusion(args, output_dims)
        elif args.fusion_type == "unified_graph":
            self.fusion = modal_unified_graph(args, output_dims)
        elif args.fusion_type == "dedicated_dnn":
            self.fusion = modal_dedicated_dnn(args, output_dims)
        elif args.fusion_type == "regression":
            self.fusion = modal_regression(args, output_dims)
        elif args.fusion_type in ("Distill_tea", "MTD_stu", "KL_stu"):
            self.fusion = modal_distillation(args, output_dims)
        elif args.fusion_type == "self_distill":
            self.fusion = modal_self_distillation(args, output_dims)
        elif args.fusion_type in _NOT_YET:
            raise NotImplementedError(f"fusion_type {args.fusion_type!r} is queued behind the 'sum' hot path (SURVEY.md 8f rank 2)")
        else:
            raise ValueError(f"unknown fusion_type {args.fusion_type!r}")

    def forward(self, data, missing_index):
        embedding = self.encoder(data)
        return self.fusion(embedding, missing_index)
