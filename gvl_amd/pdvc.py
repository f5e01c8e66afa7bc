"""PDVC orchestrator -- the ``pdvc.pdvc`` contract (``build(args)``, ``PDVC.forward``) on the MI355X hot path.

Mirrors pdvc/pdvc.py:41-314 (model + forward), :434-519 (eval: parallel_prediction_full), :540-660 (train:
parallel_prediction_matched), :699-884 (caption_prediction / caption_prediction_eval) and :1181-1238 (build), for the
configuration space of BASELINE.json: ``transformer_input_type='queries'``, LSTM-DSA captioner
(``caption_decoder_type: standard``), ``enable_contrastive=False`` (the contrastive branch needs the frozen RoBERTa
text encoder, which SURVEY.md section 2 row 15 places outside the accelerated path -- requesting it raises).
Sub-module and parameter names equal the reference's (``base_encoder, transformer, caption_head.N, query_embed,
class_head.N, count_head.N, bbox_head.N.layers.K``) so reference checkpoints load with ``strict=True``;
``PDVC.forward(dt, criterion, contrastive_criterion, transformer_input_type, eval_mode)`` returns the same
``(out, loss)`` dictionaries.
"""
import copy
import math

import os

import torch
import torch.nn.functional as F
from torch import nn

from .base_encoder import build_base_encoder
from .CaptioningHead import build_captioner
from .criterion import SetCriterion, unbind_tagged
from .deformable_transformer import build_deforamble_transformer, inverse_sigmoid
from .linear import Linear
from .matcher import build_matcher
from .postprocess import PostProcess
from . import layers as _layers_mod
from . import train_layers as _train_layers

_NO_DROP = nn.Dropout(0.0)


def _clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class MLP(nn.Module):
    """pdvc.py:1166-1178"""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i < self.num_layers - 1:
                # training: ReLU as the row kernel that leaves max |row| for the next Linear's operand split (and, in backward,
                # for the previous one's) -- two passes over the activation fewer per ReLU
                x = _train_layers.relu_dropout(x, F.relu, _NO_DROP) if torch.is_grad_enabled() and x.requires_grad else F.relu(x)
        return x


def autocast_inference_policy():
    """What an INFERENCE forward under torch.autocast runs on (GVL_AUTOCAST_INFERENCE):
    "f16"  (default) the hand-written fp32-storage path with ONE fp16 matrix-core product per fp32 product instead of three
           (MSDA.f16_products(1)): operands rounded to fp16 at their row scale -- 11 significant bits where autocast's
           bfloat16 keeps 8 -- fp32 accumulation, fp32 activations, every fused epilogue of the fp32 path;
    "fp32" the same path at full fp32 accuracy (three products): autocast may lower precision, it need not;
    "bf16" the bf16-storage path (library bf16 GEMMs + the _bf16 kernel twins)."""
    import os
    v = os.environ.get("GVL_AUTOCAST_INFERENCE", "")
    return v if v in ("bf16", "fp32") else "f16"


def autocast_training_policy():
    """What a TRAINING step under torch.autocast runs on (GVL_AUTOCAST_TRAINING; BASELINE config 5 names bf16, the reference has
    only the flag, pdvc.py:214-215):
    "f16"  (default) the hand-written fp32-storage training path -- fused layer kernels, split-fp16 Linear products forward,
           input gradient and weight gradient, the owned attention core -- with ONE fp16 matrix-core product per fp32 product
           (MSDA.f16_products(1), recorded per autograd node so that the backward takes the same): operands rounded to fp16 at
           their row / tensor scale -- 11 significant bits where autocast's bf16 carries 8 --, fp32 accumulation, fp32 master
           weights, moments and optimizer.  Through round 5 autocast sent every training Linear / attention / LayerNorm chain
           to the bf16 library GEMMs + ATen casts and the step was SLOWER than the fp32 step (cfg A 10.0 against 7.4 ms, yc2
           T = 512 11.7 against 9.7: profiles/r05_other_configs.json).
    "fp32" the same path with the exact three-product split: autocast may lower precision, it need not;
    "bf16" torch's own autocast formulation (bf16 library GEMMs, bf16-storage deformable attention), as in rounds 1-5."""
    import os
    pol = os.environ.get("GVL_AUTOCAST_TRAINING", "f16")
    if pol not in ("f16", "fp32", "bf16"):
        raise ValueError(f"GVL_AUTOCAST_TRAINING={pol!r}: expected f16, fp32 or bf16")
    return pol


def autocast_products():
    """fp16 products per fp32 product of the split-fp16 kernels under the current autocast inference policy"""
    return 1 if autocast_inference_policy() == "f16" else 3


class PDVC(nn.Module):
    def __init__(self, base_encoder, text_encoder, transformer, captioner, num_classes, num_queries,
                 num_feature_levels, aux_loss=True, with_box_refine=False, opt=None, translator=None):
        super().__init__()
        if opt.enable_contrastive or text_encoder is not None:
            raise NotImplementedError("gvl_amd.PDVC: the contrastive text branch (frozen RoBERTa) is outside the "
                                      "accelerated path; build with enable_contrastive=False")
        self.opt = opt
        self.enable_contrastive = False
        self.base_encoder = base_encoder
        self.transformer = transformer
        self.caption_head = captioner
        hidden_dim = transformer.d_model
        self.query_embed = nn.Embedding(num_queries, hidden_dim * 2)
        if vars(opt).get('support_mlp_class_head', False):
            self.class_head = MLP(hidden_dim, hidden_dim, num_classes, 3)
        else:
            self.class_head = nn.Linear(hidden_dim, num_classes)
        self.count_head = nn.Linear(hidden_dim, opt.max_eseq_length + 1)
        self.bbox_head = MLP(hidden_dim, hidden_dim, 2, 3)
        self.num_feature_levels = num_feature_levels
        self.aux_loss = aux_loss
        self.with_box_refine = with_box_refine
        self.share_caption_head = opt.share_caption_head

        # initialisation, pdvc.py:115-146
        prior_prob = 0.01
        if not vars(opt).get('support_mlp_class_head', False):
            self.class_head.bias.data = torch.ones(num_classes) * (-math.log((1 - prior_prob) / prior_prob))
        nn.init.constant_(self.bbox_head.layers[-1].weight.data, 0)
        nn.init.constant_(self.bbox_head.layers[-1].bias.data, 0)
        num_pred = transformer.decoder.num_layers
        if self.share_caption_head:
            self.caption_head = nn.ModuleList([self.caption_head for _ in range(num_pred)])
        else:
            self.caption_head = _clones(self.caption_head, num_pred)
        init_bias = vars(opt).get('box_head_init_bias', -2.0)
        if with_box_refine:
            self.class_head = _clones(self.class_head, num_pred)
            self.count_head = _clones(self.count_head, num_pred)
            self.bbox_head = _clones(self.bbox_head, num_pred)
            nn.init.constant_(self.bbox_head[0].layers[-1].bias.data[1:], init_bias)
            self.transformer.decoder.bbox_head = self.bbox_head          # iterative refinement shares these heads
            # (a plain attribute, not a registered sub-module: the inference decoder computes the class logits beside the
            #  box MLP, whose input rows they share)
            self.transformer.decoder.__dict__["_gvl_class_head"] = self.class_head
        else:
            nn.init.constant_(self.bbox_head.layers[-1].bias.data[1:], init_bias)
            self.class_head = nn.ModuleList([self.class_head for _ in range(num_pred)])
            self.count_head = nn.ModuleList([self.count_head for _ in range(num_pred)])
            self.bbox_head = nn.ModuleList([self.bbox_head for _ in range(num_pred)])
            self.transformer.decoder.bbox_head = None
        self.translator = translator
        self.disable_mid_caption_heads = opt.disable_mid_caption_heads
        self.background_embed = None

    # -- parameter groups used by train.py (pdvc.py:170-209) ---------------------------------------------------
    def get_filter_rule_for_encoder(self):
        return lambda x: ('input_proj' in x or 'transformer.encoder' in x or 'transformer.level_embed' in x
                          or 'base_encoder' in x)

    def _named(self, key):
        return [v for n, v in self.named_parameters() if key in n and v.requires_grad]

    def class_head_paramenters(self):
        return self._named('class_head')

    def captioner_parameters(self):
        return self._named('caption_head')

    def bbox_head_parameters(self):
        return self._named('bbox_head')

    def encoder_decoder_parameters(self):
        rule = self.get_filter_rule_for_encoder()
        enc, dec = [], []
        for name, p in self.named_parameters():
            (enc if rule(name) else dec).append(p)
        return enc, dec

    # -- the hot path ---------------------------------------------------------------------------------------------
    def encode(self, dt):
        """base encoder -> deformable encoder -> decoder; returns everything the heads need (pdvc.py:250-300)."""
        vf = dt['video_tensor']
        mask = ~dt['video_mask']
        duration = dt['video_length'][:, 1]
        if self.base_encoder.flat_eligible(vf, mask) and not self.transformer.no_encoder:
            # inference: pyramid, masks, position / level embeddings produced flattened by the hand-written kernels
            src, mask_flat, lvl_pos, lengths = self.base_encoder.forward_flat(vf, mask, duration,
                                                                              self.transformer.level_embed)
            tshapes, lsi, valid_ratios = self.transformer.flat_geometry(mask_flat, lengths)
        elif self.base_encoder.flat_train_eligible(vf, mask) and not self.transformer.no_encoder:
            # training: the pyramid (conv + GroupNorm of every level) as one autograd node on the hand-written kernels
            src_flat = self.base_encoder.forward_flat_train(vf)
            masks, pos = self.base_encoder.train_geometry(vf, mask, duration)
            src, tshapes, lsi, valid_ratios, lvl_pos, mask_flat = self.transformer.prepare_encoder_inputs(
                None, masks, pos, src_flatten=src_flat)
        else:
            srcs, masks, pos = self.base_encoder(vf, mask, duration)
            src, tshapes, lsi, valid_ratios, lvl_pos, mask_flat = self.transformer.prepare_encoder_inputs(srcs, masks, pos)
        memory = self.transformer.forward_encoder(src, tshapes, lsi, valid_ratios, lvl_pos, mask_flat)
        return memory, tshapes, lsi, valid_ratios, mask_flat

    def _side_stream(self, device):
        side = self.__dict__.get("_gvl_side_stream")
        if side is None or side.device != torch.device(device):
            side = self.__dict__["_gvl_side_stream"] = torch.cuda.Stream(device=device)
        return side

    def supports_padded_targets(self, criterion, eval_mode, batch=None, slots=None):
        """Can a step on this model run in the layout-independent form (dt['_gvl_targets'] = PaddedTargets; what
        gvl_amd.parallel's captured steps use)?  Eval: whenever the criterion can.  Train: additionally the caption
        losses of all decoder layers must come from the one-pass teacher-forced path (caption_prediction_layers).
        batch / slots (videos per step, target slots per video): when given, the criterion's own static conditions
        (problem sizes of the on-device solver, B * Q limit, loss set) are evaluated too -- the same ones its forward
        checks, so a captured step never meets the list-form fallback."""
        if criterion is None or not getattr(criterion, "fused", False) or not getattr(criterion, "device_matching", False):
            return False
        if batch is not None and not criterion.padded_static_ok(batch, self.opt.num_queries, slots or 4):
            return False
        if eval_mode or self.opt.caption_loss_coef == 0:
            return True
        head = self.caption_head[-1]
        return (self.training and self.aux_loss and len(self.caption_head) > 1
                and getattr(self, "one_pass_captioning", True)
                and all(h_ is head for h_ in self.caption_head)
                and not vars(self.opt).get('enable_pos_emb_for_captioner', False)
                and getattr(head, "ss_prob", 0.0) == 0.0 and hasattr(head, "core")
                and getattr(head.core, "fused_train", True) and head.core.att_feat_size == 512
                and head.core.att_hid_size == 512 and head.core.n_levels * head.core.n_points == 16
                and self.opt.set_cost_caption == 0)

    @staticmethod
    def _targets(dt):
        """the criterion's targets: the padded fixed-shape form when the caller prepared one, else the reference's list"""
        pt = dt.get('_gvl_targets')
        return pt if pt is not None else dt['video_target']

    def train_planes(self):
        """operand planes of the weights the hand-written training products read (gvl_amd/train_planes.py), built on first use"""
        tp = self.__dict__.get("_gvl_train_planes")
        dev = self.query_embed.weight.device
        if tp is None or tp.device != dev:
            from .train_planes import build_train_planes
            tp = self.__dict__["_gvl_train_planes"] = build_train_planes(self, dev)
        return tp

    def forward(self, dt, criterion, contrastive_criterion, transformer_input_type, eval_mode=False):
        from . import linear as _linear
        if (self.training and torch.is_grad_enabled() and torch.is_autocast_enabled() and dt['video_tensor'].is_cuda
                and _linear.train_linear_enabled() and autocast_training_policy() != "bf16"):
            # TRAINING under torch.autocast: the hand-written training path at one (or three) fp16 products per fp32 product
            from . import MultiScaleDeformableAttention as MSDA
            with torch.autocast("cuda", enabled=False), MSDA.f16_products(1 if autocast_training_policy() == "f16" else 3):
                return self.forward(dt, criterion, contrastive_criterion, transformer_input_type, eval_mode)
        if (self.training and torch.is_grad_enabled() and not torch.is_autocast_enabled() and dt['video_tensor'].is_cuda
                and _linear.train_linear_enabled()):
            # TRAINING: the planes of every weight the hand-written Linear products read, for this forward's parameter values
            # (two launches); the products' autograd nodes keep what their backward needs
            tp = self.train_planes()
            # (tried: the refresh on a side stream under the feature pyramid, which needs none of these planes -- a parallel branch
            #  in the captured TRAIN graph cost +0.7 ms per replay, 6.20 -> 6.88 ms; the inference graph gains from its branch)
            tp.refresh()
            from . import train_layers as _tl
            _tl.arena_reset(dt['video_tensor'].device)                    # one fill for the step's row-maxima vectors
            prev = _linear.set_active_planes(tp)
            try:
                return self._forward(dt, criterion, contrastive_criterion, transformer_input_type, eval_mode)
            finally:
                _linear.set_active_planes(prev)
        return self._forward(dt, criterion, contrastive_criterion, transformer_input_type, eval_mode)

    def _forward(self, dt, criterion, contrastive_criterion, transformer_input_type, eval_mode=False):
        if (torch.is_autocast_enabled() and not torch.is_grad_enabled() and not self.training
                and dt['video_tensor'].is_cuda and autocast_inference_policy() != "bf16"):
            # INFERENCE under torch.autocast (BASELINE config 5 is named "bf16"): the forward stays on the hand-written
            # fp32-storage path (fused epilogues, fused token loop) and lowers precision where autocast would -- in the Linear
            # products -- by spending one fp16 matrix-core product per fp32 product instead of three (policy "f16", the
            # default: 11-bit operands against bf16's 8, 1.4x the fp32 path's throughput; tools/x1_probe.py).  The bf16
            # library GEMMs + ATen casts autocast would route the layers through are SLOWER than even the exact fp32 path
            # (profiles/r03_other_configs.json: yc2 T = 512, 2936 against 3121 videos/s).  GVL_AUTOCAST_INFERENCE=fp32 keeps
            # the exact products, =bf16 restores the bf16-storage path; training under autocast is unaffected.
            from . import MultiScaleDeformableAttention as MSDA
            with torch.autocast("cuda", enabled=False), MSDA.f16_products(autocast_products()):
                return self.forward(dt, criterion, contrastive_criterion, transformer_input_type, eval_mode)
        N = dt['video_tensor'].shape[0]
        memory, tshapes, lsi, valid_ratios, mask_flat = self.encode(dt)
        cut = getattr(self, "memory_cut", None)
        if cut is not None and torch.is_grad_enabled():
            # two-stage backward of the data-parallel captured step (gvl_amd.parallel): everything below sees a detached
            # leaf; the caller later feeds its gradient into the encoder's graph
            memory = cut(memory)
        if transformer_input_type == 'gt_proposals':                              # misc/utils.py:32-43
            proposals_mask = dt['gt_boxes_mask']
            criterion.matcher.cost_caption = 0
            for q_k in ['loss_length', 'loss_ce', 'loss_bbox', 'loss_giou']:
                for key in criterion.weight_dict.keys():
                    if q_k in key:
                        criterion.weight_dict[key] = 0
            disable_refine = True
            init_reference, tgt, reference_points, query_embed = \
                self.transformer.prepare_decoder_input_proposal(dt['gt_boxes'])
        elif transformer_input_type == 'queries':
            disable_refine = False
            query_embed = self.query_embed.weight
            proposals_mask = torch.ones(N, query_embed.shape[0], device=query_embed.device).bool()
            proposals_mask._gvl_all_true = (True, proposals_mask._version)      # (known without reading it back: gvl_amd/layers.py)
            init_reference, tgt, reference_points, query_embed = \
                self.transformer.prepare_decoder_input_query(memory, query_embed)
        else:
            raise ValueError('Wrong value of transformer_input_type, got {}'.format(transformer_input_type))
        hs, inter_references = self.transformer.forward_decoder(tgt, reference_points, memory, tshapes, lsi,
                                                                valid_ratios, query_embed, mask_flat, proposals_mask,
                                                                disable_refine)
        others = {'memory': memory, 'mask_flatten': mask_flat, 'spatial_shapes': tshapes, 'level_start_index': lsi,
                  'valid_ratios': valid_ratios, 'proposals_mask': proposals_mask, 'text_embed': None,
                  'event_embed': hs, 'pre_proj_text_embed': None}
        if eval_mode or self.opt.caption_loss_coef == 0:
            return self.parallel_prediction_full(dt, criterion, contrastive_criterion, hs, query_embed,
                                                 init_reference, inter_references, others, disable_refine,
                                                 self.opt.eval_disable_captioning)
        if self.opt.set_cost_caption > 0:
            # pdvc.py:305-309 -> parallel_prediction_full_train (:322-432) -> caption_prediction(..., indices=None) (:355,
            # :672): for the LSTM-DSA captioner (caption_decoder_type 'standard', the only one of this path) the reference
            # itself fails there -- `max([len(feat_ids) for feat_ids, _ in indices])` at pdvc.py:743 iterates None:
            # "TypeError: 'NoneType' object is not iterable" (reproduced by tests/golden/make_golden.py:make_full_train_probe,
            # recorded in tests/golden/full_train_probe.npz).  No reference config sets set_cost_caption > 0; the branch has
            # no behaviour to mirror other than this error, which is raised with the reference's own type and message.
            raise TypeError("'NoneType' object is not iterable (set_cost_caption > 0 is not supported: the reference itself "
                            "fails here, pdvc.py:743, with the LSTM-DSA captioner)")
        return self.parallel_prediction_matched(dt, criterion, contrastive_criterion, hs, query_embed,
                                                init_reference, inter_references, others, disable_refine)

    def predict_event_num(self, counter, hs_lid):
        # max over the queries (pdvc.py:316-319); amax = the same values without the argmax bookkeeping of torch.max
        from . import layers as _layers
        if _layers.count_head_eligible(counter, hs_lid):
            return _layers.count_head(counter, hs_lid)                  # pooling + Linear in one launch (inference)
        if _layers.count_pool_train_eligible(hs_lid):
            return counter(_layers.count_pool_train(hs_lid))            # (training: pooling node with a one-launch gradient)
        return counter(torch.amax(hs_lid, dim=1))

    def _layer_heads(self, l_id, hs_lid, reference, disable_refine):
        """class / count / box heads of one decoder layer (pdvc.py:452-474)."""
        # inference with box refinement: the decoder (gvl_amd/layers.py) has just applied this very box MLP, the class head
        # and the refinement arithmetic to these very rows (deformable_transformer.py:314-324 and pdvc.py:452-474 share
        # `bbox_head[l_id]`): its results are taken over
        dec = self.transformer.decoder.__dict__
        # (training: the decoder leaves the refined boxes of every layer WITH their autograd history; class head and deltas
        #  are shared by the inference path only)
        take = self.with_box_refine and not disable_refine and (
            not torch.is_grad_enabled() or os.environ.get("GVL_SHARE_COORDS", "1") != "0")
        shared = {k: (dec.get("_gvl_" + k) if take else None) for k in ("cls", "coords", "deltas")}
        same = lambda t_: t_ is not None and t_[l_id].shape[:2] == hs_lid.shape[:2]          # noqa: E731
        if not same(shared["cls"]) and _layers_mod.class_count_heads_eligible(self.class_head[l_id], hs_lid):
            # training, one class: the class head and the count head's pooling as one node (their gradients reach hs together)
            cls, pooled = _layers_mod.class_count_heads(self.class_head[l_id], hs_lid)
            cnt = self.count_head[l_id](pooled)
        else:
            cls = shared["cls"][l_id] if same(shared["cls"]) else self.class_head[l_id](hs_lid)
            cnt = self.predict_event_num(self.count_head[l_id], hs_lid)
        if same(shared["coords"]):
            return cls, cnt, shared["coords"][l_id]
        delta = shared["deltas"][l_id] if same(shared["deltas"]) else self.bbox_head[l_id](hs_lid)
        if disable_refine:
            coord = reference
        else:
            prior = inverse_sigmoid(reference)
            if prior.shape[-1] == 2:
                coord = (delta + prior).sigmoid()
            else:
                assert prior.shape[-1] == 1
                coord = torch.cat([delta[..., :1] + prior, delta[..., 1:]], -1).sigmoid()
        return cls, cnt, coord

    def _no_caption(self, hs):
        """the placeholders of a layer without captioning (pdvc.py:476-479).  Inference: constants, built once per shape
        instead of three zero fills per layer and forward.  They belong to THIS model (captured eval graphs and earlier
        output dicts alias them) and are never evicted; the set of (batch, queries) shapes a model sees is small."""
        N_, N_q = hs.shape[:2]
        if hs.is_cuda:                          # (training too: nothing writes into the placeholders, they carry no gradient)
            cache = self.__dict__.setdefault("_no_caption_consts", {})
            key = (N_, N_q, str(hs.device))
            hit = cache.get(key)
            if hit is None and not torch.cuda.is_current_stream_capturing():
                hit = cache[key] = (torch.zeros(1, device=hs.device), torch.zeros(N_, N_q, 3, device=hs.device),
                                    torch.zeros(N_, N_q, 3, device=hs.device))
            if hit is not None:                       # (capturing: constants created by the warm-up run are reused)
                return {'cap_prob_train': hit[0], 'cap_prob_eval': hit[1]}, hit[2]
        probs = {'cap_prob_train': torch.zeros(1, device=hs.device),
                 'cap_prob_eval': torch.zeros(N_, N_q, 3, device=hs.device)}
        return probs, torch.zeros(N_, N_q, 3, device=hs.device)

    def _pack(self, hs, classes, counts, coords, cap_probs, seqs):
        """the per-layer outputs under the reference's keys (pdvc.py:283-290 stacks the three prediction tensors and then
        only ever indexes the stack by layer: the per-layer tensors themselves are kept instead -- no stack launch, and no
        zero-filled SelectBackward + accumulation per layer and key in the train step)"""
        num_pred = hs.shape[0]
        all_out = {'pred_logits': classes, 'pred_count': counts, 'pred_boxes': coords, 'caption_probs': cap_probs,
                   'seq': seqs, 'cl_match_mats': [0] * num_pred}
        return all_out

    def parallel_prediction_full(self, dt, criterion, contrastive_criterion, hs, query_embed, init_reference,
                                 inter_references, others, disable_iterative_refine, disable_captioning=False):
        num_pred = hs.shape[0]
        # Inference: the count heads (one workgroup per video), matching (a one-workgroup assignment kernel) and the losses --
        # ~35 small launches, 0.25 ms -- depend on the decoder's output only, the greedy token loop (6 ms) on none of them: they run
        # on a side stream UNDER the token loop instead of in front of / behind it (inside a hipGraph capture: a parallel branch
        # of the graph).  GVL_EVAL_OVERLAP=0: one stream, as before.
        overlap = (criterion is not None and not torch.is_grad_enabled() and hs.is_cuda and not disable_captioning
                   and self.opt.caption_decoder_type == 'standard' and os.environ.get("GVL_EVAL_OVERLAP", "1") != "0")

        def heads_and_losses(with_captions):
            classes, counts, coords, cap_probs, seqs = [], [], [], [], []
            for l_id in range(num_pred):
                reference = init_reference if l_id == 0 else inter_references[l_id - 1]
                hs_lid = hs[l_id]
                if l_id == num_pred - 1 and getattr(hs, "_gvl_amax", None) is not None:
                    hs_lid._gvl_amax = hs._gvl_amax      # row maxima of the last layer's rows (gvl_amd/layers.py); a view
                    #                                      shares its base's version counter, so the tag stays checkable
                cls, cnt, coord = self._layer_heads(l_id, hs_lid, reference, disable_iterative_refine)
                hs_cap = torch.cat([hs_lid, query_embed], dim=-1) if vars(self.opt).get('enable_pos_emb_for_captioner',
                                                                                        False) else hs_lid
                if l_id != num_pred - 1 or disable_captioning or not with_captions:
                    probs, seq = self._no_caption(hs_cap)         # (overlap: placeholders, replaced after the join)
                else:
                    probs, seq = self.caption_prediction_eval(self.caption_head[l_id], dt, hs_cap, reference, others,
                                                              self.opt.caption_decoder_type)
                classes.append(cls); counts.append(cnt); coords.append(coord); cap_probs.append(probs); seqs.append(seq)
            if torch.is_grad_enabled():
                self.transformer.decoder.__dict__["_gvl_coords"] = None       # (see parallel_prediction_matched)
            all_out = self._pack(hs, classes, counts, coords, cap_probs, seqs)
            all_out['event_embed'] = others['event_embed']
            all_out['event_feat'] = hs
            out = {k: v[-1] for k, v in all_out.items()}
            if self.aux_loss:
                keys = list(all_out.keys())
                out['aux_outputs'] = [{k: all_out[k][j] for k in keys} for j in range(num_pred - 1)]
            if criterion is None:                               # pure-inference use (bench / graph capture)
                return out, {}
            loss, last_indices, *_ = criterion(out, targets)
            return out, loss

        targets = self._targets(dt) if criterion is not None else None
        if not overlap:
            return heads_and_losses(True)
        main = torch.cuda.current_stream(hs.device)
        side = self._side_stream(hs.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out, loss = heads_and_losses(False)                 # (the criterion reads the heads' outputs only)
        l_id = num_pred - 1
        reference = init_reference if l_id == 0 else inter_references[l_id - 1]
        hs_cap = torch.cat([hs[l_id], query_embed], dim=-1) if vars(self.opt).get('enable_pos_emb_for_captioner',
                                                                                  False) else hs[l_id]
        probs, seq = self.caption_prediction_eval(self.caption_head[l_id], dt, hs_cap, reference, others,
                                                  self.opt.caption_decoder_type)
        main.wait_stream(side)
        if not torch.cuda.is_current_stream_capturing():          # (a captured graph's memory is static: nothing to record)
            def hand_over(v):                                     # everything the side stream allocated is read on this one from here on
                if isinstance(v, torch.Tensor):
                    if v.is_cuda:
                        v.record_stream(main)
                elif isinstance(v, dict):
                    for w in v.values():
                        hand_over(w)
                elif isinstance(v, (list, tuple)):
                    for w in v:
                        hand_over(w)
            hand_over(loss)
            hand_over({k: v for k, v in out.items() if k not in ('event_embed', 'event_feat')})
        out['caption_probs'], out['seq'] = probs, seq
        return out, loss

    def caption_prediction_eval(self, cap_head, dt, hs, reference, others, decoder_type, indices=None):
        """pdvc.py:885-930 for the 'none' and 'standard' heads."""
        assert indices is None
        N_, N_q, C = hs.shape
        if decoder_type == 'none':
            return self._no_caption(hs)
        if decoder_type != 'standard':
            raise ValueError(f"caption_decoder_type {decoder_type!r} is not built by gvl_amd")
        with torch.no_grad():
            seq, cap_prob_eval = cap_head.sample(hs, reference, others)
            if len(seq):
                seq = seq.reshape(-1, N_q, seq.shape[-1])
                cap_prob_eval = cap_prob_eval.reshape(-1, N_q, cap_prob_eval.shape[-1])
        return {'cap_prob_eval': cap_prob_eval}, seq

    def parallel_prediction_matched(self, dt, criterion, contrastive_criterion, hs, query_embed, init_reference,
                                    inter_references, others, disable_iterative_refine):
        """Train step heads (pdvc.py:540-660): set losses on all queries, captioning only on the matched ones."""
        num_pred = hs.shape[0]
        # the per-layer slices taken ONCE: every `hs[l]` of the stacked decoder output is a SelectBackward of its own
        # (a zero-filled (layers, B, Q, C) tensor + a copy + an accumulation) -- unbind's backward is one stack
        hs_l = hs.unbind(0)
        classes, counts, coords, cap_probs, seqs = [], [], [], [], []
        for l_id in range(num_pred):
            reference = init_reference if l_id == 0 else inter_references[l_id - 1]
            cls, cnt, coord = self._layer_heads(l_id, hs_l[l_id], reference, disable_iterative_refine)
            probs, seq = self._no_caption(hs_l[l_id])
            classes.append(cls); counts.append(cnt); coords.append(coord); cap_probs.append(probs); seqs.append(seq)
        # the decoder's refined boxes carry this step's autograd history: they must not outlive the step on the module (a
        # graph kept alive across iterations makes the next backward wait on the old AccumulateGrad streams: +1.5 ms)
        self.transformer.decoder.__dict__["_gvl_coords"] = None
        all_out = self._pack(hs, classes, counts, coords, cap_probs, seqs)
        out = {k: v[-1] for k, v in all_out.items()}
        if self.aux_loss:
            keys = list(all_out.keys())
            out['aux_outputs'] = [{k: all_out[k][j] for k in keys} for j in range(num_pred - 1)]
            loss, last_indices, aux_indices = criterion(out, self._targets(dt))
            layers = range(num_pred)
        else:
            loss, last_indices = criterion(out, self._targets(dt))
            aux_indices = None
            layers = [num_pred - 1]
        layers = list(layers)
        matches = [last_indices if l_id == num_pred - 1 else aux_indices[l_id] for l_id in layers]
        refs = [init_reference if l_id == 0 else inter_references[l_id - 1] for l_id in layers]
        if getattr(matches[0], "plan", None) is not None and getattr(matches[0].plan, "padded", False):
            if not self._one_pass_captioning(layers, matches, hs, None):
                raise RuntimeError("padded targets need the one-pass teacher-forced caption path "
                                   "(PDVC.supports_padded_targets)")
            cap_losses, probs, seq = self.caption_prediction_layers_padded(
                self.caption_head[layers[-1]], dt['_gvl_targets'], [hs_l[l_id] for l_id in layers], refs, others, matches)
            for l_id, cap_loss in zip(layers, cap_losses):
                loss['loss_caption' if l_id == num_pred - 1 else f'loss_caption_{l_id}'] = cap_loss
            out.update({'caption_probs': probs, 'seq': seq})
            return out, loss
        if self._one_pass_captioning(layers, matches, hs, dt):
            # every decoder layer's caption loss from ONE pass of the shared caption head over all layers' matched
            # queries (the value / ctx2att slab, the token loop and every weight gradient are then computed once)
            cap_losses, probs, seq = self.caption_prediction_layers(self.caption_head[layers[-1]], dt,
                                                                    [hs_l[l_id] for l_id in layers], refs, others, matches)
            for l_id, cap_loss in zip(layers, cap_losses):
                loss['loss_caption' if l_id == num_pred - 1 else f'loss_caption_{l_id}'] = cap_loss
            out.update({'caption_probs': probs, 'seq': seq})
            return out, loss
        for l_id in layers:
            reference = init_reference if l_id == 0 else inter_references[l_id - 1]
            layer_match = last_indices if l_id == num_pred - 1 else aux_indices[l_id]
            indices = layer_match if hasattr(layer_match, "plan") else layer_match[0]
            hs_cap = torch.cat([hs_l[l_id], query_embed], dim=-1) if vars(self.opt).get('enable_pos_emb_for_captioner',
                                                                                        False) else hs_l[l_id]
            cap_loss, probs, seq = self.caption_prediction(self.caption_head[l_id], dt, hs_cap, reference, others,
                                                           indices)
            key = 'loss_caption' if l_id == num_pred - 1 else f'loss_caption_{l_id}'
            loss[key] = cap_loss
        out.update({'caption_probs': probs, 'seq': seq})
        return out, loss

    def _one_pass_captioning(self, layers, matches, hs, dt):
        head = self.caption_head[layers[-1]]
        return (self.training and len(layers) > 1 and getattr(self, "one_pass_captioning", True)
                and all(hasattr(m, "plan") for m in matches)
                and all(self.caption_head[l_id] is head for l_id in layers)
                and not vars(self.opt).get('enable_pos_emb_for_captioner', False)
                and getattr(head, "ss_prob", 0.0) == 0.0 and hasattr(head, "core")
                and head.core.fused_train_eligible(hs[0]) and torch.is_grad_enabled()
                and (dt is None or all(len(t_['boxes']) <= hs.shape[2] for t_ in dt['video_target'])))

    def caption_prediction_layers_padded(self, cap_head, pt, hs_layers, ref_layers, others, matches):
        """caption_prediction_layers on PaddedTargets, with a COMPACT fixed-capacity row set: decoder layer k owns rows
        [k * R, (k + 1) * R), R = pt.pair_rows (a capacity >= the number of matched pairs of the batch); row r of a layer
        is the r-th matched (query, caption) pair in (video, slot) order -- found on the device from the per-video pair
        counts (cumulative sum + searchsorted) -- and rows beyond the batch's pair count are marked unused
        (row_video = -1: the captioner kernels skip them, their caption is <pad> with an all-zero mask, loss and gradient
        contribution exactly 0).  The teacher-forced loop runs the static pt.cap_len - 1 steps: steps beyond the longest
        caption of the batch only see masked positions, so the loss equals the reference's, which leaves the loop at the
        first all-<pad> input column (LSTM_DSA.py:110-112).  The mean over (videos x max pairs) rows of pdvc.py:868
        divides by a DEVICE scalar.  No shape depends on the batch."""
        nl = len(matches)
        plan = matches[0].plan
        N_, N_q, C = hs_layers[0].shape
        G1 = plan.G1
        if pt.slots > N_q:
            raise RuntimeError("more target slots than queries")
        R = pt.pair_rows or N_ * G1
        dev = hs_layers[0].device
        cache = plan.__dict__.setdefault("_cap_rows", {})
        consts = cache.get((nl, R))
        if consts is None:
            consts = cache[(nl, R)] = (torch.arange(R, dtype=torch.int64, device=dev),
                                       torch.arange(nl, dtype=torch.int64, device=dev).repeat_interleave(R))
        r, lay = consts
        n1 = plan.pair_count
        hs_stack = torch.stack(hs_layers)
        ref_stack = torch.stack([r_ if r_.shape[-1] == 2 else torch.cat([r_, torch.full_like(r_, -1.0)], -1)
                                 for r_ in ref_layers])
        import os
        rows_kernel = (os.environ.get("GVL_CAPTION_ROWS", "") != "torch" and dev.type == "cuda" and nl <= 8 and pt.cap_mask.dtype == torch.float32 and n1.dtype == torch.int64
                       and all(m.q.dtype == torch.int64 and m.q.is_contiguous() and m.t.is_contiguous() for m in matches))
        if rows_kernel:
            # the pair rows -- video, slot, matched query / target, caption row and mask of every (layer, r) -- in ONE launch
            # (gvl_caption_rows) instead of the ~45 small index operations of the formulation below
            import ctypes
            from . import _lib
            cl = pt.cap_len
            flat = torch.empty(nl * R, dtype=torch.int64, device=dev)
            row_video = torch.empty(nl * R, dtype=torch.int64, device=dev)
            seq_flat = torch.empty(nl * R, cl, dtype=torch.int64, device=dev)
            mask_flat = torch.empty(nl * R, cl, dtype=torch.float32, device=dev)
            denom = torch.empty(1, dtype=torch.float32, device=dev)
            qs = (ctypes.c_void_p * nl)(*[m.q.data_ptr() for m in matches])
            ts = (ctypes.c_void_p * nl)(*[m.t.data_ptr() for m in matches])
            n1c, ct, cm = n1.contiguous(), pt.cap_tensor.contiguous(), pt.cap_mask.contiguous()
            with torch.cuda.device(dev):
                rc = _lib.lib().gvl_caption_rows(n1c.data_ptr(), qs, ts, nl, N_, G1, R, N_q, pt.slots, cl, ct.data_ptr(), cm.data_ptr(),
                                                 flat.data_ptr(), row_video.data_ptr(), seq_flat.data_ptr(), mask_flat.data_ptr(),
                                                 denom.data_ptr(), torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "caption_rows")
        else:
            incl = n1.cumsum(0)
            v = torch.searchsorted(incl, r, right=True)                     # video of the r-th pair
            used = v < N_
            v = v.clamp(max=N_ - 1)
            e = v * G1 + (r - (incl - n1)[v]).clamp(min=0, max=G1 - 1)      # its slot in the padded match layout
            q_all = torch.cat([m.q[e] for m in matches]).clamp(min=0)
            t_all = torch.cat([m.t[e] for m in matches]).clamp(min=0)
            v_all, used_all = v.repeat(nl), used.repeat(nl)
            row_video = torch.where(used_all, v_all, torch.full_like(v_all, -1))
            flat = (lay * N_ + v_all) * N_q + q_all
            seq_flat = pt.cap_tensor[v_all, t_all] * used_all[:, None]
            mask_flat = pt.cap_mask[v_all, t_all] * used_all[:, None]
            denom = N_ * n1.max().clamp(min=1)
        # index_select on the flattened (layer, video, query) axis: its backward is an atomic index_add.  Advanced
        # indexing (hs_stack[lay, vid, q]) differentiates into index_put_(accumulate=True), a rocPRIM radix sort -- the
        # same sort inside nn.Embedding's backward faulted when replayed from a hipGraph on MI355X / ROCm 7.2
        hs_m = hs_stack.view(-1, C).index_select(0, flat)
        ref_m = ref_stack.view(-1, 2).index_select(0, flat)
        cap_prob = cap_head(hs_m, ref_m, others, seq_flat, steps=pt.cap_len - 1, row_video=row_video,
                            nll=(seq_flat[:, 1:], mask_flat[:, 1:]))
        row_loss = cap_head.build_loss(cap_prob, seq_flat[:, 1:], mask_flat[:, 1:])
        # (a batch without a single event: every row is masked, the sum is exactly 0 -- 0 / 1, not 0 / 0: a NaN here would
        #  flow through clip_grad_norm_ into the captured Adam and poison parameters and moments for good)
        per_layer = row_loss.view(nl, R).sum(dim=1) / denom
        return unbind_tagged(per_layer), {}, pt.cap_tensor[plan.vid_of_entry, matches[-1].t.clamp(min=0)]

    def caption_prediction_layers(self, cap_head, dt, hs_layers, ref_layers, others, matches):
        """caption_prediction for all decoder layers at once (shared head): the matched queries of layer k of video v
        occupy rows [k * max_pairs, (k + 1) * max_pairs) of video v.  Layers whose reference points are centre-only
        (decoder input, ref-dim 1) travel in the two-component layout with length -1 (include/gvl_msda.h)."""
        nl = len(matches)
        plan = matches[0].plan
        N_, N_q, C = hs_layers[0].shape
        dev = hs_layers[0].device
        cap_len = dt['cap_tensor'].shape[-1]
        mp = max(plan.n1)
        cache = plan.__dict__.setdefault("_cap_rows", {})
        rows = cache.get(nl)
        if rows is None:          # static (host-known) positions, built once per matching plan
            t1 = plan.vid_of_entry.numel()
            rows = cache[nl] = (plan.vid_of_entry.repeat(nl),
                                torch.cat([plan.slot_of_entry + k * mp for k in range(nl)]),
                                torch.arange(nl, device=dev).repeat_interleave(t1))
        vid_all, slot_all, lay_all = rows
        q_all = torch.cat([m.q for m in matches])
        caps_all = torch.cat([m.t_global for m in matches])
        hs_stack = torch.stack(hs_layers)
        ref_stack = torch.stack([r if r.shape[-1] == 2 else torch.cat([r, torch.full_like(r, -1.0)], -1)
                                 for r in ref_layers])
        hs_m = hs_stack.new_zeros(N_, nl * mp, C)
        ref_m = ref_stack.new_zeros(N_, nl * mp, 2)
        seq_m = torch.zeros(N_, nl * mp, cap_len, dtype=torch.long, device=dev)
        mask_m = torch.zeros(N_, nl * mp, cap_len, dtype=torch.bool, device=dev)
        hs_m[vid_all, slot_all] = hs_stack[lay_all, vid_all, q_all]
        ref_m[vid_all, slot_all] = ref_stack[lay_all, vid_all, q_all]
        seq_m[vid_all, slot_all] = dt['cap_tensor'][caps_all].long()
        mask_m[vid_all, slot_all] = dt['cap_mask'][caps_all].bool()
        seq_flat, mask_flat = seq_m.flatten(0, 1), mask_m.flatten(0, 1)
        steps = dt.get('_gvl_cap_steps')
        if steps is None:
            live = (dt['cap_tensor'][:, 1:] != 0).any(0).cpu().tolist()
            steps = min(1 + (live.index(False) if False in live else len(live)), cap_len - 1)
            dt['_gvl_cap_steps'] = steps
        cap_prob = cap_head(hs_m, ref_m, others, seq_flat, steps=steps, nll=(seq_flat[:, 1:], mask_flat[:, 1:]))
        row_loss = cap_head.build_loss(cap_prob, seq_flat[:, 1:], mask_flat[:, 1:])
        per_layer = row_loss.view(N_, nl, mp).mean(dim=(0, 2))
        return unbind_tagged(per_layer), {}, dt['cap_tensor'][matches[-1].t_global]

    def caption_prediction(self, cap_head, dt, hs, reference, others, indices):
        """Teacher-forced caption loss on the matched (query, caption) pairs (pdvc.py:743-884, 'standard' head):
        per video the matched queries are packed to the front of a (N, max_pairs, .) tensor, padded rows carry an
        all-False mask and contribute 0 to the mean exactly as in the reference."""
        N_, N_q, C = hs.shape
        dev = hs.device
        cap_len = dt['cap_tensor'].shape[-1]
        if hasattr(indices, "plan"):
            # device-resident matching (LayerMatch): one scatter with host-known static positions, no host sync.
            # (captions of video i occupy rows [sum n_gt[:i], ...) of dt['cap_tensor'] -- the same offsets as the
            #  concatenated targets, i.e. LayerMatch.t_global)
            plan = indices.plan
            max_pairs = max(plan.n1)
            vid, slot, caps = plan.vid_of_entry, plan.slot_of_entry, indices.t_global
            hs_m = hs.new_zeros(N_, max_pairs, C)
            ref_m = reference.new_zeros(N_, max_pairs, reference.shape[-1])
            seq_m = torch.zeros(N_, max_pairs, cap_len, dtype=torch.long, device=dev)
            mask_m = torch.zeros(N_, max_pairs, cap_len, dtype=torch.bool, device=dev)
            hs_m[vid, slot] = hs[vid, indices.q]
            ref_m[vid, slot] = reference[vid, indices.q]
            seq_m[vid, slot] = dt['cap_tensor'][caps].long()
            mask_m[vid, slot] = dt['cap_mask'][caps].bool()
            all_caps = [caps]
        else:
            indices = indices[0] if not isinstance(indices, list) else indices
            gt_nums = [len(t_['boxes']) for t_ in dt['video_target']]
            cap_base = [0]
            for n in gt_nums[:-1]:
                cap_base.append(cap_base[-1] + int(n))
            max_pairs = max(len(f_) for f_, _ in indices)
            hs_m = hs.new_zeros(N_, max_pairs, C)
            ref_m = reference.new_zeros(N_, max_pairs, reference.shape[-1])
            seq_m = torch.zeros(N_, max_pairs, cap_len, dtype=torch.long, device=dev)
            mask_m = torch.zeros(N_, max_pairs, cap_len, dtype=torch.bool, device=dev)
            all_caps = []
            for i, (feat_ids, cap_ids) in enumerate(indices):
                k = len(feat_ids)
                f_dev = feat_ids.to(dev)
                caps = (cap_base[i] + cap_ids).to(dev)
                hs_m[i, :k] = hs[i, f_dev]
                ref_m[i, :k] = reference[i, f_dev]
                seq_m[i, :k] = dt['cap_tensor'][caps]
                mask_m[i, :k] = dt['cap_mask'][caps].bool()
                all_caps.append(caps)
        seq_flat, mask_flat = seq_m.flatten(0, 1), mask_m.flatten(0, 1)
        # teacher-forcing length: the reference stops at the first all-<pad> input column of the matched captions; when
        # every caption is matched (Q >= #GT) that is a property of dt['cap_tensor'] alone -> read it once per batch
        steps = None
        if all(len(t_['boxes']) <= N_q for t_ in dt['video_target']):
            steps = dt.get('_gvl_cap_steps')
            if steps is None:
                live = (dt['cap_tensor'][:, 1:] != 0).any(0).cpu().tolist()
                steps = min(1 + (live.index(False) if False in live else len(live)), cap_len - 1)
                dt['_gvl_cap_steps'] = steps
        if self.training:
            cap_prob = cap_head(hs_m, ref_m, others, seq_flat, steps=steps)
            probs, seq = {}, dt['cap_tensor'][torch.cat(all_caps)]
        else:
            with torch.no_grad():
                cap_prob = cap_head(hs_m, ref_m, others, seq_flat, steps=steps)
                seq, cap_prob_eval = cap_head.sample(hs, reference, others)
                if len(seq):
                    seq = seq.reshape(-1, N_q, seq.shape[-1])
                    cap_prob_eval = cap_prob_eval.reshape(-1, N_q, cap_prob_eval.shape[-1])
                probs = {'cap_prob_eval': cap_prob_eval}
        cap_prob = cap_prob.reshape(-1, cap_prob.shape[-2], cap_prob.shape[-1])
        cap_loss = cap_head.build_loss(cap_prob, seq_flat[:, 1:], mask_flat[:, 1:])
        return cap_loss.mean(), probs, seq


def build(args):
    """pdvc.py:1181-1238 -> (model, criterion, contrastive_criterion, postprocessors)."""
    if args.enable_contrastive:
        raise NotImplementedError("gvl_amd.pdvc.build: enable_contrastive=True needs the frozen RoBERTa text encoder "
                                  "(out of the accelerated path); set enable_contrastive=False")
    base_encoder = build_base_encoder(args)
    transformer = build_deforamble_transformer(args)
    captioner = build_captioner(args)
    model = PDVC(base_encoder, None, transformer, captioner, num_classes=args.num_classes,
                 num_queries=args.num_queries, num_feature_levels=args.num_feature_levels, aux_loss=args.aux_loss,
                 with_box_refine=args.with_box_refine, opt=args)
    matcher = build_matcher(args)
    weight_dict = {'loss_ce': args.cls_loss_coef, 'loss_bbox': args.bbox_loss_coef, 'loss_giou': args.giou_loss_coef,
                   'loss_counter': args.count_loss_coef, 'loss_caption': args.caption_loss_coef,
                   'contrastive_loss': vars(args).get('contrastive_loss_start_coef', 0.0)}
    if args.aux_loss:
        aux = {}
        for i in range(args.dec_layers - 1):
            aux.update({k + f'_{i}': v for k, v in weight_dict.items()})
        weight_dict.update(aux)
    criterion = SetCriterion(args.num_classes, matcher, weight_dict, ['labels', 'boxes', 'cardinality'],
                             focal_alpha=args.focal_alpha, focal_gamma=args.focal_gamma, opt=args)
    criterion.to(torch.device(args.device))
    postprocessors = {'bbox': PostProcess(args)}
    return model, criterion, None, postprocessors
