"""Temporal deformable transformer -- host-side mirror of pdvc/deformable_transformer.py on the HIP MSDA op.

The public surface is the reference's: ``DeformableTransformer`` with ``prepare_encoder_inputs`` (:85),
``forward_encoder`` (:117), ``prepare_decoder_input_query`` (:128), ``prepare_decoder_input_proposal`` (:137),
``forward_decoder`` (:154), and ``build_deforamble_transformer(args)`` (:353; the reference's spelling is kept).
Sub-module / parameter names match the reference so that its checkpoints load with ``strict=True``:
``encoder.layers.N.{self_attn,norm1,linear1,linear2,norm2}``, ``decoder.layers.N.{cross_attn,norm1,self_attn,
norm2,linear1,linear2,norm3}``, ``decoder.bbox_head`` (set by PDVC), ``level_embed, pos_trans, pos_trans_norm,
reference_points``.

MI355X notes: level lengths are Python ints here, so the (L,)-shaped ``temporal_shapes`` / ``level_start_index``
tensors carry a host copy (``_gvl_host_lengths``) and nothing on the path reads the device back
(the reference's ``assert input_spatial_shapes.sum() == Len_in`` synchronises every layer).
"""
import copy
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import layers as _layers
from . import train_layers as _tl
from . import train_mha as _mha
from .ops.modules import MSDeformAttn
from .linear import Linear


def inverse_sigmoid(x, eps=1e-5):
    """misc/detr_utils/misc.py:582-586"""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def _clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def _activation(name):
    if name == "relu":
        return F.relu
    if name == "gelu":
        return F.gelu
    if name == "glu":
        return F.glu
    raise RuntimeError(F"activation should be relu/gelu, not {name}.")


_LEVEL_CACHE = {}


def make_level_tensors(lengths, device):
    """(temporal_shapes (L,), level_start_index (L,)) int64 on `device`, each carrying host copies.  Cached per
    (lengths, device): the tensors are constants, so no host->device copy is issued per forward."""
    lengths = [int(x) for x in lengths]
    key = (tuple(lengths), str(device))
    hit = _LEVEL_CACHE.get(key)
    if hit is not None:
        return hit
    starts = [0]
    for x in lengths[:-1]:
        starts.append(starts[-1] + x)
    ts = torch.tensor(lengths, dtype=torch.long, device=device)
    ls = torch.tensor(starts, dtype=torch.long, device=device)
    ts._gvl_host_lengths = (lengths, starts)
    ls._gvl_host_lengths = (lengths, starts)
    _LEVEL_CACHE[key] = (ts, ls)
    return ts, ls


class DeformableTransformerEncoderLayer(nn.Module):
    """:159-199 -- x = LN(x + drop(MSDA(x + pos, ref, x)));  x = LN(x + drop(W2 drop(act(W1 x))))"""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_ffn)
        self.activation = _activation(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, src, pos=None, resid=None, fan=1):
        # (norm(x + dropout(sub)): one hand-written forward / backward kernel in training, gvl_amd/train_layers.py; `pos`: the
        #  result also carries the row maxima of result + pos for the next layer's attention query.  resid: the handle of `src` the
        #  residual reads, fan: the number of handles of the result -- see residual_dropout_norm(fan=...))
        return _tl.residual_dropout_norm(src if resid is None else resid,
                                         self.linear2(_tl.relu_dropout(self.linear1(src), self.activation, self.dropout2)),
                                         self.dropout3, self.norm2, pos, fan=fan)

    def forward(self, src, pos, reference_points, temporal_shapes, level_start_index, padding_mask=None, out_fan=1):
        """src: a tensor, or the three handles (attention query, attention input, residual) the previous layer returned for
        out_fan=3; out_fan > 1: the result as that many handles (one per consumer: residual_dropout_norm(fan=...))"""
        sq, sv, sr = src if isinstance(src, tuple) else (src, src, src)
        attn = self.self_attn(_tl.add_pos(sq, pos), reference_points, sv, temporal_shapes, level_start_index, padding_mask)
        # (two handles of norm1's result, for the FFN and for the residual around it: their gradients meet inside norm1's backward)
        a, b = _tl.residual_dropout_norm(sr, attn, self.dropout1, self.norm1, fan=2)
        return self.forward_ffn(a, pos, resid=b, fan=out_fan)


class DeformableTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(temporal_shapes, valid_ratios, device):
        """:209-218 -- centre of every frame of every level, normalised by that level's valid length and re-scaled
        to every target level: (B, S, L, 1)."""
        host = getattr(temporal_shapes, "_gvl_host_lengths", None)
        lengths = host[0] if host is not None else [int(x) for x in temporal_shapes.tolist()]
        per_level = []
        for lvl, T in enumerate(lengths):
            centres = torch.linspace(0.5, T - 0.5, T, dtype=torch.float32, device=device)
            per_level.append(centres[None] / (valid_ratios[:, None, lvl] * T))
        ref = torch.cat(per_level, 1)
        return (ref[:, :, None] * valid_ratios[:, None])[..., None]

    def forward(self, src, temporal_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None):
        if _layers.encoder_eligible(self, src, temporal_shapes):
            # inference: the layers' Linear / LayerNorm chain on the hand-written kernels of gvl_amd/layers.py
            return _layers.encoder_forward(self, src, temporal_shapes, level_start_index, valid_ratios, pos,
                                           padding_mask)
        ref = _layers.enc_ref_of(valid_ratios)          # left by prepare_encoder_inputs (one launch, same bits)
        if ref is None or ref.shape[:2] != src.shape[:2]:
            ref = self.get_reference_points(temporal_shapes, valid_ratios, device=src.device)
        out = src
        for i, layer in enumerate(self.layers):
            # (a layer's result has three consumers in the next layer: one handle each)
            out = layer(out, pos, ref, temporal_shapes, level_start_index, padding_mask,
                        out_fan=3 if i + 1 < len(self.layers) else 1)
        return out



class DeformableTransformerDecoderLayer(nn.Module):
    """:229-280 -- MHA over the queries, MSDA cross-attention into the memory, FFN; post-norm."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_ffn)
        self.activation = _activation(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt, pos=None, resid=None, fan=1):
        return _tl.residual_dropout_norm(tgt if resid is None else resid,
                                         self.linear2(_tl.relu_dropout(self.linear1(tgt), self.activation, self.dropout3)),
                                         self.dropout4, self.norm3, pos, fan=fan)

    def forward(self, tgt, query_pos, reference_points, src, src_temporal_shapes, level_start_index,
                src_padding_mask=None, query_mask=None, out_fan=1):
        """tgt: a tensor, or the three handles (attention values, attention query, residual) the previous layer returned;
        out_fan > 1: the result as that many handles (one per consumer: residual_dropout_norm(fan=...))"""
        tgt, tgt_q, tgt_r = tgt if isinstance(tgt, tuple) else (tgt, tgt, tgt)
        if _mha.eligible(self.self_attn, tgt, query_pos):
            # training: in-projection (one launch, the positional addend applied in its load path), attention core and
            # out-projection on the hand-written kernels (gvl_amd/train_mha.py); same parameters, same result
            sa = _mha.self_attention(self.self_attn, tgt, query_pos, query_mask, tgt_q=tgt_q)
        else:
            qk = _tl.add_pos(tgt_q, query_pos).transpose(0, 1)
            # The reference discards the averaged attention map ([0] at pdvc/deformable_transformer.py:267-268).  Not asking for
            # it lets nn.MultiheadAttention take its fused attention path: measured 0.4 % of the eval step; in training the
            # fused forward + backward kernels are slower than bmm / softmax / bmm at this size (300 queries): +0.9 % of the step
            # (average_attn_weights=False: the unfused path without the mean over the heads of a (B, 8, Q, Q) map nobody reads)
            sa = self.self_attn(qk, qk, tgt.transpose(0, 1), key_padding_mask=~query_mask, need_weights=self.training,
                                average_attn_weights=False)[0].transpose(0, 1)
        # (sa: a transposed view, read in place.  Two handles of each norm's result -- one for the sublayer that follows, one for
        #  the residual around it: residual_dropout_norm(fan=2))
        tq, tr = _tl.residual_dropout_norm(tgt_r, sa, self.dropout2, self.norm2, query_pos, fan=2)
        ca = self.cross_attn(_tl.add_pos(tq, query_pos), reference_points, src, src_temporal_shapes,
                             level_start_index, src_padding_mask)
        tf, tr = _tl.residual_dropout_norm(tr, ca, self.dropout1, self.norm1, fan=2)
        return self.forward_ffn(tf, query_pos, resid=tr, fan=out_fan)


class DeformableTransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, return_intermediate=False):
        super().__init__()
        self.layers = _clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.bbox_head = None          # set by PDVC when with_box_refine (pdvc.py:134-140)

    def forward(self, tgt, reference_points, src, src_temporal_shapes, src_level_start_index, src_valid_ratios,
                query_pos=None, src_padding_mask=None, query_padding_mask=None, disable_iterative_refine=False):
        for k in ("_gvl_deltas", "_gvl_coords", "_gvl_cls"):
            self.__dict__[k] = None
        if _layers.decoder_eligible(self, tgt, src, src_temporal_shapes):
            return _layers.decoder_forward(self, tgt, reference_points, src, src_temporal_shapes,
                                           src_level_start_index, src_valid_ratios, query_pos, src_padding_mask,
                                           query_padding_mask, disable_iterative_refine)
        out = tgt
        hs, refs, coords = [], [], []
        next_in = None                                 # the next layer's scaled reference points, when the refinement kernel left them
        for lid, layer in enumerate(self.layers):
            if next_in is not None:
                ref_in, next_in = next_in, None
            elif reference_points.shape[-1] == 2:                                     # (centre, length): :302-304
                ref_in = reference_points[:, :, None] * torch.stack([src_valid_ratios] * 2, -1)[:, None]
            else:
                assert reference_points.shape[-1] == 1
                ref_in = reference_points[:, :, None] * src_valid_ratios[:, None, :, None]
            # one handle of the layer's result per consumer (three in the next layer, the box head, the returned stack): their
            # gradients meet inside norm3's backward kernel instead of in a chain of autograd adds
            refine = not disable_iterative_refine and self.bbox_head is not None
            n_next = 3 if lid + 1 < len(self.layers) else 0
            k = n_next + int(refine) + int(self.return_intermediate)
            outs = layer(out, query_pos, ref_in, src, src_temporal_shapes, src_level_start_index, src_padding_mask,
                         query_padding_mask, out_fan=max(k, 1))
            outs = list(outs) if isinstance(outs, tuple) else [outs] * max(k, 1)
            out = tuple(outs[:3]) if n_next else outs[0]
            out_box = outs[n_next] if refine else None
            out_hs = outs[-1]
            if refine:                                                                # :314-324
                delta = self.bbox_head[lid](out_box)
                if _layers.box_refine_train_eligible(delta, reference_points):
                    # sigmoid(delta + inverse_sigmoid(reference)) and the next layer's reference points in one launch, one
                    # more for the gradient (gvl_amd/layers.py: _BoxRefineTrain)
                    new_ref, next_in = _layers.box_refine_train(delta, reference_points, src_valid_ratios,
                                                                lid + 1 < len(self.layers))
                else:
                    prior = inverse_sigmoid(reference_points)
                    if reference_points.shape[-1] == 2:
                        new_ref = (delta + prior).sigmoid()
                    else:
                        new_ref = torch.cat([delta[..., :1] + prior, delta[..., 1:]], -1).sigmoid()
                reference_points = new_ref.detach()
                coords.append(new_ref)
            if self.return_intermediate:
                hs.append(out_hs)
                refs.append(reference_points)
        # pdvc.py:452-474 applies the SAME box MLP to the same rows and adds the same inverse_sigmoid(reference): PDVC's
        # heads take these (attached) results instead of repeating ~23 launches per layer (as the inference path does)
        self.__dict__["_gvl_coords"] = coords if len(coords) == len(self.layers) else None
        if self.return_intermediate:
            return torch.stack(hs), torch.stack(refs)
        return out_hs, reference_points


class DeformableTransformer(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_feature_levels=4, dec_n_points=4,
                 enc_n_points=4):
        super().__init__()
        self.d_model = d_model
        self.nhead = nhead
        self.no_encoder = (num_encoder_layers == 0)
        self.num_feature_levels = num_feature_levels
        enc_layer = DeformableTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                      num_feature_levels, nhead, enc_n_points)
        self.encoder = DeformableTransformerEncoder(enc_layer, num_encoder_layers)
        dec_layer = DeformableTransformerDecoderLayer(d_model, dim_feedforward, dropout, activation,
                                                      num_feature_levels, nhead, dec_n_points)
        self.decoder = DeformableTransformerDecoder(dec_layer, num_decoder_layers, return_intermediate_dec)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self.pos_trans = nn.Linear(d_model, d_model * 2)
        self.pos_trans_norm = nn.LayerNorm(d_model * 2)
        self.reference_points = nn.Linear(d_model, 1)
        self._reset_parameters()
        # the layers' Linears own their weights alone (a layer runs once per step, nothing is tied into it): their weight
        # gradients may wait for the backward pass's grouped launches (gvl_amd/linear.py: _WgradQueue)
        for layer in list(self.encoder.layers) + list(self.decoder.layers):
            for m in layer.modules():
                if isinstance(m, (Linear, MSDeformAttn)):
                    m.defer_wgrad = True
                elif isinstance(m, nn.MultiheadAttention):
                    m.__dict__["_gvl_defer_wgrad"] = True

    def _reset_parameters(self):
        """:54-63 (same order of RNG consumption): xavier every matrix, then the MSDeformAttn-specific init."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        nn.init.xavier_uniform_(self.reference_points.weight.data, gain=1.0)
        nn.init.constant_(self.reference_points.bias.data, 0.)
        nn.init.normal_(self.level_embed)

    # -- proposal -> query embedding (:66-79, :137-151) --------------------------------------------------------
    def get_proposal_pos_embed(self, proposals):
        num_pos_feats, temperature, scale = 256, 10000, 2 * math.pi
        dim_t = torch.arange(num_pos_feats, dtype=torch.float32, device=proposals.device)
        dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
        pos = (proposals.sigmoid() * scale)[:, :, :, None] / dim_t
        return torch.stack((pos[:, :, :, 0::2].sin(), pos[:, :, :, 1::2].cos()), dim=4).flatten(2)

    def get_valid_ratio(self, mask):
        return torch.sum(~mask, 1).float() / mask.shape[1]

    def prepare_encoder_inputs(self, srcs, masks, pos_embeds, src_flatten=None):
        """:85-115.  srcs[l] (B,C,T_l), masks[l] (B,T_l) True=pad, pos_embeds[l] (B,C,T_l).  src_flatten given (the training pyramid
        of BaseEncoder.forward_flat_train): the levels arrive flattened already, srcs is None."""
        lengths = [int(m.shape[-1]) for m in masks]
        if src_flatten is None:
            src_flatten = torch.cat([s.transpose(1, 2) for s in srcs], 1)
        mask_flatten = torch.cat(masks, 1)
        if _layers.level_pos_embed_eligible(self.level_embed, pos_embeds):
            lvl_pos = _layers.level_pos_embed(self.level_embed, pos_embeds)       # (training: the embedding's gradient in two launches)
        else:
            lvl_embed = self.level_embed.unbind(0)    # (one UnbindBackward instead of a zero-filled SelectBackward per level)
            lvl_pos = torch.cat([p.transpose(1, 2) + lvl_embed[l].view(1, 1, -1) for l, p in enumerate(pos_embeds)], 1)
        temporal_shapes, level_start_index = make_level_tensors(lengths, src_flatten.device)
        if (_layers.enabled() and mask_flatten.is_cuda and not self.no_encoder and len(lengths) <= 8
                and mask_flatten.dtype == torch.bool):
            # valid ratios and the encoder's reference points (:209-218) from the flat mask in ONE launch (masks carry no
            # gradient: the training forward takes the same kernel)
            starts = temporal_shapes._gvl_host_lengths[1]
            valid_ratios, ref = _layers.encoder_geometry(mask_flatten, lengths, starts)
            valid_ratios._gvl_enc_ref = (ref, valid_ratios._version)
        else:
            valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)
        return src_flatten, temporal_shapes, level_start_index, valid_ratios, lvl_pos, mask_flatten

    def flat_geometry(self, mask_flatten, lengths):
        """level tensors + valid ratios (+ the encoder's reference points, stashed on them) for inputs that arrive already
        flattened (BaseEncoder.forward_flat): the tail of prepare_encoder_inputs (:100-115)"""
        temporal_shapes, level_start_index = make_level_tensors(lengths, mask_flatten.device)
        starts = temporal_shapes._gvl_host_lengths[1]
        valid_ratios, ref = _layers.encoder_geometry(mask_flatten, lengths, starts)
        valid_ratios._gvl_enc_ref = (ref, valid_ratios._version)
        return temporal_shapes, level_start_index, valid_ratios

    def forward_encoder(self, src_flatten, temporal_shapes, level_start_index, valid_ratios, lvl_pos_embed_flatten,
                        mask_flatten):
        if self.training and torch.is_grad_enabled():
            _tl.advance(src_flatten.device)           # new dropout masks for this forward's residual chains (train_layers.py)
        if self.no_encoder:
            return src_flatten
        return self.encoder(src_flatten, temporal_shapes, level_start_index, valid_ratios, lvl_pos_embed_flatten,
                            mask_flatten)

    def prepare_decoder_input_query(self, memory, query_embed):
        """:128-135 -- query_embed (Q, 2C) is chunked into (query_pos, tgt) in that order."""
        bs = memory.shape[0]
        pos_rows, tgt = torch.chunk(query_embed, 2, dim=1)
        if _layers.expand_parts_eligible(query_embed, 2):
            # (training: the two gradients batch-summed in one launch; pos_rows: the node's own handle of the first block)
            query_pos, tgt, pos_rows, tgt_rows = _layers.expand_parts(query_embed, bs, 2, rows=True)
            # (the first decoder layer's in-projection runs on these Q rows: gvl_amd/train_mha.py _InProjShared)
            query_pos._gvl_rows, tgt._gvl_rows = pos_rows, tgt_rows
        else:
            query_pos = pos_rows.unsqueeze(0).expand(bs, -1, -1)
            tgt = tgt.unsqueeze(0).expand(bs, -1, -1)
        if not torch.is_grad_enabled() and query_embed.is_cuda and isinstance(self.reference_points, nn.Linear):
            # inference: sigmoid(reference_points(query_pos)) depends on parameters only -- kept until one of them changes (the
            # reference evaluates the Linear on the batch-expanded rows: the same arithmetic per row)
            lin = self.reference_points
            ps = [query_embed, lin.weight] + ([lin.bias] if lin.bias is not None else [])
            key = tuple((p_.data_ptr(), p_._version) for p_ in ps) + (
                bs, torch.is_autocast_enabled(), torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None)
            hit = self.__dict__.get("_gvl_query_ref")
            if hit is None or hit[0] != key:
                hit = self.__dict__["_gvl_query_ref"] = (key, lin(query_pos).sigmoid())
            reference_points = hit[1]
        else:
            # (training: the Linear on the Q rows of the embedding, THEN the batch expansion -- the same arithmetic per row as the
            #  reference's Linear over the batch-expanded rows, without materialising them and their gradient)
            reference_points = self.reference_points(pos_rows).sigmoid().unsqueeze(0).expand(bs, -1, -1)
        return reference_points, tgt, reference_points, query_pos

    def prepare_decoder_input_proposal(self, gt_reference_points, inversed_input=False):
        if inversed_input:
            unact = gt_reference_points
            gt_reference_points = torch.sigmoid(gt_reference_points)
        else:
            unact = inverse_sigmoid(gt_reference_points)
        emb = self.pos_trans_norm(self.pos_trans(self.get_proposal_pos_embed(unact)))
        query_pos, tgt = torch.chunk(emb, 2, dim=2)
        return gt_reference_points, tgt, gt_reference_points, query_pos

    def convert_proposal_to_query(self, gt_reference_points):
        return self.pos_trans_norm(self.pos_trans(self.get_proposal_pos_embed(inverse_sigmoid(gt_reference_points))))

    def forward_decoder(self, *kargs):
        return self.decoder(*kargs)


def build_deforamble_transformer(args):
    return DeformableTransformer(
        d_model=args.hidden_dim, nhead=args.nheads, num_encoder_layers=args.enc_layers,
        num_decoder_layers=args.dec_layers, dim_feedforward=args.transformer_ff_dim,
        dropout=args.transformer_dropout_prob, activation="relu", return_intermediate_dec=True,
        num_feature_levels=args.num_feature_levels, dec_n_points=args.dec_n_points, enc_n_points=args.enc_n_points)
