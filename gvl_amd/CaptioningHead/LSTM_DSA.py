"""LSTM captioner with deformable soft attention -- mirror of pdvc/CaptioningHead/LSTM_DSA.py.

Parameter names follow the reference (``embed, logit, core.rnn.weight_{ih,hh}_l0, core.deformable_att.*,
core.ctx2att, core.h2att, core.alpha_net``).  The arithmetic of one token step is the reference's
(ShowAttendTellCore.forward, LSTM_DSA.py:241-271, and Captioner.get_logprobs_state, :120-124):

    jq   = [h_{t-1} | hs]                                   (B, Q, 2C)
    clip = MSDeformAttnCap(jq, ref, memory)                 16 border-padded samples of value_proj(memory) per query
    e    = alpha_net(tanh(ctx2att(clip) + h2att(h)))        softmax over the 16 samples -> alpha
    att  = sum_k alpha_k clip_k
    h,c  = LSTM([embed(it) | att | hs], (h, c))             single layer, bias-free
    logp = log_softmax(logit(dropout(h)))

What is restructured for the GPU (results equal up to fp32 rounding):
  * value_proj(memory) does not depend on the token: it is computed once per forward, not once per step (the
    reference recomputes it at ms_deform_attn_for_caption.py:98 every step);
  * ctx2att is linear and border-padded interpolation weights sum to one, so ctx2att(clip) == samples of
    ctx2att(value): the per-step (B*Q*16, C) x (C, A) GEMM becomes one (B*S, C) x (C, A) GEMM per forward and the
    sampler reads the [value | ctx2att(value)] slab;
  * the hs / embedding parts of the LSTM input GEMM are hoisted out of the token loop;
  * the greedy loop runs without a per-step device->host sync (the reference tests ``unfinished.sum() == 0`` on
    the host every step, :186); the all-finished cut is applied once at the end.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops.functions import MSDASampleFunction
from ..ops.modules import MSDeformAttnCap
from ..ops.modules.ms_deform_attn import temporal_shapes_2d
from .. import MultiScaleDeformableAttention as MSDA
from ..linear import mirror_linear as _mirror_linear, mirror_linear_eligible as _mirror_ok
from ..linear import Linear, split_gemm_enabled, split_linear, vocab_nll, vocab_nll_eligible


_ATTEND_PRE = os.environ.get("GVL_ATTEND_PRE", "1") != "0"            # (A/B: the offsets' hidden-state product inside the attention kernel)
_GREEDY_MERGED = os.environ.get("GVL_GREEDY_MERGED", "1") != "0"      # (A/B: h2att(h) as a launch of its own)

class ShowAttendTellCore(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_size = opt.rnn_size
        self.num_layers = opt.num_layers
        self.drop_prob_lm = opt.drop_prob
        self.att_feat_size = int(opt.clip_context_dim / opt.cap_nheads)
        self.att_hid_size = opt.att_hid_size
        self.wordRNN_input_feats_type = opt.wordRNN_input_feats_type
        self.input_dim = opt.hidden_dim * (3 if vars(opt).get('enable_pos_emb_for_captioner', False) else 2)
        self.rnn = nn.LSTM(self.input_encoding_size + self.input_dim, self.rnn_size, self.num_layers, bias=False,
                           dropout=self.drop_prob_lm)
        self.att_drop = nn.Dropout(0.5)
        self.n_levels = opt.cap_num_feature_levels
        self.n_heads = opt.cap_nheads
        self.n_points = opt.cap_dec_n_points
        self.deformable_att = MSDeformAttnCap(opt.hidden_dim, self.n_levels, self.n_heads, self.n_points, opt)
        if self.att_hid_size > 0:
            self.ctx2att = Linear(self.att_feat_size, self.att_hid_size)       # (training: the hand-written product, as value_proj)
            self.ctx2att.defer_wgrad = True
            self.h2att = nn.Linear(self.rnn_size, self.att_hid_size)
            self.alpha_net = nn.Linear(self.att_hid_size, 1)
        if self.num_layers != 1 or self.n_heads != 1 or self.att_hid_size <= 0:
            raise ValueError("gvl_amd's LSTM-DSA core covers num_layers=1, cap_nheads=1, att_hid_size>0 "
                             "(every reference config)")

    # -- per-forward constants ---------------------------------------------------------------------------------
    def _prepare_inference(self, query, input_flatten, input_padding_mask, perm=None):
        """prepare() for inference on the split-fp16 layer kernel (gvl_amd/layers.py): value_proj(memory) with masked rows
        straight into the left half of the [value | ctx2att(value)] slab, ctx2att of it into the right half, and the two
        token-independent products of the event features (gate part, offset part) as one launch: 3 launches, no cat."""
        from .. import layers as L
        att = self.deformable_att
        B, S, C = input_flatten.shape
        A, H, E, Cf = self.att_hid_size, self.rnn_size, self.input_encoding_size, self.att_feat_size
        Rs = B * S
        mem = input_flatten.reshape(Rs, C)
        am_mem = L.amax_of(input_flatten, Rs)
        if am_mem is None:
            am_mem, _ = L.row_absmax(mem)
        slab = torch.empty(B, S, Cf + A, device=mem.device, dtype=torch.float32)
        slab2 = slab.view(Rs, Cf + A)
        am_v = torch.zeros(Rs, device=mem.device, dtype=torch.float32)
        mask = input_padding_mask.reshape(Rs).contiguous().view(torch.uint8) if input_padding_mask is not None else None
        L.linear(mem, L.cached(att, "capv", [(att.value_proj.weight, att.value_proj.bias)]),
                 [L.seg(0, slab2[:, :Cf], am_mem, rowmask=mask, amax_out=am_v)])
        L.linear(slab2[:, :Cf], L.cached(self, "ctx", [(self.ctx2att.weight, self.ctx2att.bias)]),
                 [L.seg(0, slab2[:, Cf:], am_v)])
        q = query.reshape(-1, query.shape[-1])
        am_q = L.amax_of(query, q.shape[0])
        if am_q is None:
            am_q, _ = L.row_absmax(q)
        K = self.n_levels * self.n_points
        ow = att.sampling_offsets.weight
        if perm is None:
            w = L.cached(self, "hs", [(self.rnn.weight_ih_l0[:, E + Cf:], None), (ow[:, H:], att.sampling_offsets.bias)])
        else:                                                           # gate rows in the order 4 unit + gate
            w = L.cached(self, "hs_perm", lambda: [(self.rnn.weight_ih_l0[:, E + Cf:][perm], None),
                                                   (ow[:, H:], att.sampling_offsets.bias)],
                         key_of=(self.rnn.weight_ih_l0, ow, att.sampling_offsets.bias))
        gates_hs = torch.empty(q.shape[0], 4 * H, device=q.device, dtype=torch.float32)
        off_hs = torch.empty(q.shape[0], K, device=q.device, dtype=torch.float32)
        L.linear(q, w, [L.seg(0, gates_hs, am_q), L.seg(w.starts[1], off_hs, am_q, width=K)])
        return slab.view(B, S, 1, Cf + A), gates_hs, off_hs.view(*query.shape[:-1], K)

    def _inference_layers_ok(self, query, input_flatten):
        from .. import layers as L
        return (L.enabled() and not torch.is_grad_enabled() and not torch.is_autocast_enabled() and query.is_cuda
                and query.dtype == torch.float32 and input_flatten.dtype == torch.float32 and self.att_hid_size % 64 == 0
                and self.att_feat_size % 64 == 0 and query.shape[-1] % 32 == 0 and (4 * self.rnn_size) % 64 == 0
                and self.att_feat_size == input_flatten.shape[-1] and query.is_contiguous()
                and self.rnn.weight_ih_l0.shape[1] == self.input_encoding_size + self.att_feat_size + query.shape[-1]
                and self.deformable_att.sampling_offsets.weight.shape[1] == self.rnn_size + query.shape[-1])

    def prepare(self, query, input_flatten, input_padding_mask):
        """Everything that does not depend on the token index."""
        att = self.deformable_att
        if self._inference_layers_ok(query, input_flatten):
            weights = self._inference_weights(torch.float32)
            slab, gates_hs, off_hs = self._prepare_inference(query, input_flatten, input_padding_mask,
                                                             weights.get("gate_perm"))
            B, S = slab.shape[:2]
            bias = self.alpha_net.bias
            if getattr(self, "_alpha_b_version", None) != bias._version:
                self._alpha_b = float(bias.detach().cpu())
                self._alpha_b_version = bias._version
            const = {"slab": slab, "gates_hs": gates_hs, "off_hs": off_hs}
            const.update(slab3=slab.view(B, S, -1), alpha_b=self._alpha_b, **weights)
            return const
        if torch.is_grad_enabled() and _cap_slab_eligible(self, input_flatten, input_padding_mask):
            B, S = input_flatten.shape[:2]
            mask_u8 = input_padding_mask.contiguous().view(torch.uint8) if input_padding_mask is not None else None
            slab = _CapSlab.apply(input_flatten, mask_u8, att.value_proj.weight, att.value_proj.bias, self.ctx2att.weight,
                                  self.ctx2att.bias).view(B, S, 1, -1)           # (training: both halves written in place)
        else:
            value = att.project_value(input_flatten, input_padding_mask)          # (B,S,1,C)
            B, S = value.shape[:2]
            v2 = value.reshape(B, S, -1)
            va = self.ctx2att(v2)                                                 # ctx2att pushed through sampling
            slab = torch.cat([v2, va], -1).view(B, S, 1, -1).contiguous()        # (B,S,1,C+A)
        C = query.shape[-1]
        w_ih = self.rnn.weight_ih_l0
        E = self.input_encoding_size
        weights = self._inference_weights(slab.dtype) if not torch.is_grad_enabled() else {}
        # the three column blocks of W_ih -- embedding | attention | event features (LSTM_DSA.py:267-269 concatenates its input in that
        # order) -- and the two of the offsets' weight, each taken by ONE split: its backward is one cat of the blocks' gradients,
        # where three separate slices differentiate into three zero-filled full-size matrices (12.6 MB each), three copies and two adds
        parts = w_ih.split([E, self.att_feat_size, w_ih.shape[1] - E - self.att_feat_size], 1)
        ow_parts = att.sampling_offsets.weight.split([self.rnn_size, att.sampling_offsets.weight.shape[1] - self.rnn_size], 1)
        w_hs = parts[2]
        if weights.get("gate_perm") is not None:                                  # fused cell: gates as 4 unit + gate
            w_hs = w_hs[weights["gate_perm"]]
        gates_hs = F.linear(query.reshape(-1, C), w_hs)                           # hs part of the LSTM input
        # the offsets projection splits into an h part (per step) and an hs part (constant)
        off_hs = F.linear(query, ow_parts[1], att.sampling_offsets.bias)               # (B,Q,16)
        const = {"slab": slab, "gates_hs": gates_hs, "off_hs": off_hs.float().contiguous(),    # offsets stay fp32
                 "w_ih_parts": parts, "ow_parts": ow_parts}
        if not torch.is_grad_enabled():
            # inference: everything the fused token-step kernel (gvl_cap_attend_f32) needs, laid out once
            bias = self.alpha_net.bias
            if getattr(self, "_alpha_b_version", None) != bias._version:
                self._alpha_b = float(bias.detach().cpu())        # one host read per weight update, not per step
                self._alpha_b_version = bias._version
            const.update(slab3=slab.view(B, S, -1), alpha_b=self._alpha_b, **weights)
        return const

    def fused_train_eligible(self, query):
        """domain of gvl_cap_attend_train_* (include/gvl_msda.h)"""
        return (getattr(self, "fused_train", True) and query.is_cuda and query.dtype == torch.float32
                and self.att_feat_size == 512 and self.att_hid_size == 512
                and self.n_levels * self.n_points == 16)

    def teacher_forced(self, xt_all, query, reference_points, temporal_shapes, level_start_index, const,
                       row_video=None, time_major=False):
        """every teacher-forced token step as one autograd node (TeacherForcedLoop) -> hidden (n, steps, H); time_major:
        xt_all and hidden are (steps, n, .) -- the loop's own layout, no transposed copies either way"""
        att = self.deformable_att
        H, E, C = self.rnn_size, self.input_encoding_size, self.att_feat_size
        K = self.n_levels * self.n_points
        ow_h = const["ow_parts"][0] if "ow_parts" in const else att.sampling_offsets.weight[:, :H]
        w_hcat = torch.cat([self.h2att.weight, self.rnn.weight_hh_l0, ow_h], 0)
        from .. import train_layers as _tl
        b_hcat = torch.cat([self.h2att.bias, _tl.step_zeros(4 * H + K, query.device)])       # (zeros: the step's one fill)
        slab = const["slab"]
        B, S = slab.shape[:2]
        return TeacherForcedLoop.apply(
            slab.view(B, S, -1), reference_points.contiguous(), const["off_hs"].reshape(-1, K), const["gates_hs"],
            xt_all, w_hcat, b_hcat,
            const["w_ih_parts"][1] if "w_ih_parts" in const else self.rnn.weight_ih_l0[:, E:E + C],    # (a column block: strided)
            self.alpha_net.weight.reshape(-1),
            self.alpha_net.bias.reshape(1), temporal_shapes_2d(temporal_shapes, level_start_index), level_start_index,
            self.n_levels, self.n_points, row_video, time_major)

    def _inference_weights(self, gemm_dtype=torch.float32):
        """Weight-only operands of the fused token step (re-laid-out / concatenated weights): rebuilt only when a
        parameter changes (its autograd version counter), not once per forward.  gemm_dtype = bfloat16 under autocast:
        the GEMM operands are stored in that type once (autocast would re-cast these derived, non-parameter tensors on
        every call); the operands of the attention kernel stay fp32."""
        params = (self.deformable_att.sampling_offsets.weight, self.h2att.weight, self.h2att.bias, self.rnn.weight_hh_l0,
                  self.rnn.weight_ih_l0, self.alpha_net.weight)
        cell_fused = os.environ.get("GVL_CELL_FUSED", "1") != "0"
        gates_fused = cell_fused and os.environ.get("GVL_GATES_FUSED", "1") != "0"
        key = tuple((p_.data_ptr(), p_._version) for p_ in params) + (split_gemm_enabled(), cell_fused, gates_fused, gemm_dtype)
        cache = self.__dict__.setdefault("_inf_w", {})
        if cache and next(iter(cache))[:-1] != key[:-1]:
            cache.clear()
        w = cache.get(key)
        if w is None:
            ow, E = params[0], self.input_encoding_size
            with torch.no_grad(), torch.autocast("cuda", enabled=False):
                w = dict(w_off_h=ow[:, :self.rnn_size].contiguous(),
                         alpha_w=self.alpha_net.weight.reshape(-1).contiguous(),
                         # one GEMM over h per step yields both h2att(h) and the recurrent gate pre-activations
                         w_h_cat=torch.cat([self.h2att.weight, self.rnn.weight_hh_l0], 0).to(gemm_dtype),
                         b_h_cat=torch.cat([self.h2att.bias, self.h2att.bias.new_zeros(4 * self.rnn_size)]).to(gemm_dtype),
                         w_att_t=self.rnn.weight_ih_l0[:, E:E + self.att_feat_size].t().contiguous().to(gemm_dtype))
                if (gemm_dtype == torch.float32 and split_gemm_enabled() and self.rnn_size % 32 == 0
                        and self.att_feat_size % 32 == 0):
                    # fp32 products on the fp16 matrix cores (gvl_gemm_f16x3_f32): the weight side is split once
                    w_hh, w_att = self.rnn.weight_hh_l0, self.rnn.weight_ih_l0[:, E:E + self.att_feat_size]
                    if cell_fused and self.rnn_size % 8 == 0:
                        # the LSTM cell runs in the epilogue of the attention product (gvl_gemm_f16x3_lstm_f32): every
                        # gate operand of the step is laid out as 4 unit + gate (GVL_CELL_FUSED=0: separate cell kernel)
                        perm = w["gate_perm"] = MSDA.gate_permutation(self.rnn_size, w_hh.device)
                        w_hh, w_att = w_hh[perm], w_att[perm]
                    w["w_h_cat_p"] = MSDA.split_rows(torch.cat([self.h2att.weight, w_hh], 0))
                    w["w_att_p"] = MSDA.split_rows(w_att.contiguous())
                    if gates_fused and w.get("gate_perm") is not None:
                        # BOTH halves of the gate product + the cell in one launch (gvl_gemm_f16x3_gates_f32: contraction
                        # [h | att]); the product over h in front of the attention then only yields h2att(h).
                        # GVL_GATES_FUSED=0: the (n, A + 4H) product + gates_h operand of round 4.
                        w["w_gate_cat_p"] = MSDA.split_rows(torch.cat([w_hh, w_att], 1).contiguous())
                        # ... and, as 16 more output columns, the hidden-state part of the sampling offsets
                        # (gvl_cap_attend_pre_f32 then reads neither h nor that weight; GVL_ATTEND_PRE=0: the kernel's own product)
                        if _ATTEND_PRE:
                            w["w_h2att_p"] = MSDA.split_rows(torch.cat([self.h2att.weight, ow[:, :self.rnn_size]], 0).contiguous())
                            w["b_h2att"] = torch.cat([self.h2att.bias.detach(), self.h2att.bias.new_zeros(ow.shape[0])])
                            w["off_pre"] = True
                        else:
                            w["w_h2att_p"] = MSDA.split_rows(self.h2att.weight.contiguous())
                            w["b_h2att"] = self.h2att.bias.detach().contiguous()
            cache[key] = w
        return w

    # -- the fused inference token step in two halves: the first does not depend on the input token -----------------
    def attend_part(self, h, reference_points, temporal_shapes, level_start_index, const):
        """h -> (attention part of the gate pre-activations (n, 4H), [h2att(h) | h W_hh^T] (n, A + 4H))"""
        shapes2d = const.get("shapes2d")
        if shapes2d is None:
            shapes2d = const["shapes2d"] = temporal_shapes_2d(temporal_shapes, level_start_index)
            const["ref_in"] = reference_points.contiguous()
            host = getattr(temporal_shapes, "_gvl_host_lengths", None)          # (lengths, starts) known without a read-back
            const["host_starts"] = tuple(host[1]) if host is not None else None
        A = self.att_hid_size
        split = "w_h_cat_p" in const and h.dtype == torch.float32
        gates_one = split and self.gates_in_one_launch(h, const)
        if split:
            hp = getattr(h, "_gvl_planes", None)                         # left by the vocabulary product of the last step
            hp = hp if hp is not None else MSDA.split_rows(h)
            if gates_one:
                g_h = getattr(h, "_gvl_h2att", None)                     # left by the greedy step's launch (_greedy_iterations)
                if g_h is None:
                    g_h = MSDA.gemm_f16x3(hp, const["w_h2att_p"], const["b_h2att"])        # (n, A): h2att(h) only
            else:
                g_h = MSDA.gemm_f16x3(hp, const["w_h_cat_p"], const["b_h_cat"])
        else:
            h_gemm = getattr(h, "_gvl_lowp", h)                          # bf16 copy left by the cell kernel (autocast)
            g_h = F.linear(h_gemm, const["w_h_cat"], const["b_h_cat"])  # (n, A + 4H): [h2att(h) | h W_hh^T]
        split = split and const["slab3"].dtype == torch.float32
        pre = (gates_one and g_h.shape[1] == A + self.n_levels * self.n_points and self.attend_pre_ok(const, temporal_shapes))
        if pre:
            att_res = MSDA.cap_attend_pre(const["slab3"], shapes2d, level_start_index, const["ref_in"], const["off_hs"],
                                          g_h[:, A:], g_h[:, :A], const["alpha_w"], const["alpha_b"], self.n_levels,
                                          self.n_points, const["host_starts"])
        else:
            if getattr(h, "_gvl_planes_only", False):
                raise RuntimeError("gvl_amd: the hidden state of this step exists as operand planes only (cell_part under "
                                   "const['h_planes_only']) but the attention kernel that reads it as fp32 was chosen")
            att_res = MSDA.cap_attend(const["slab3"], shapes2d, level_start_index, const["ref_in"], const["off_hs"],
                                      h, const["w_off_h"], g_h[:, :A], const["alpha_w"], const["alpha_b"],
                                      self.n_levels, self.n_points, planes=split, host_starts=const.get("host_starts"))
        if gates_one:                                                   # the recurrent operand travels as planes
            return att_res, (hp,)
        if split and const.get("gate_perm") is not None:                # ... whose epilogue is the cell (step)
            return att_res, g_h
        if split:                                                       # att_res arrives as the planes of the product
            return MSDA.gemm_f16x3(att_res, const["w_att_p"]), g_h
        return torch.mm(att_res, const["w_att_t"]), g_h                 # (the hs part, gates_hs, is added in the cell)

    def attend_pre_ok(self, const, temporal_shapes):
        """whether this decode's attention runs as gvl_cap_attend_pre_f32 (the offsets' hidden-state product rides in the h2att(h)
        launch); decided once per set of step constants"""
        ok = const.get("pre_ok")
        if ok is None:
            host = getattr(temporal_shapes, "_gvl_host_lengths", None)
            ok = const["pre_ok"] = bool(const.get("off_pre") and host is not None and const["slab3"].dtype == torch.float32
                                        and MSDA.cap_attend_pre_applicable(const["slab3"].shape[1], self.n_levels, self.n_points,
                                                                           tuple(host[1])))
        return ok

    def gates_in_one_launch(self, h, const):
        """whether this step's gate product runs as gvl_gemm_f16x3_gates_f32 (the product over h in front of the attention is
        then h2att(h) alone)"""
        return ("w_gate_cat_p" in const and h.dtype == torch.float32 and const["slab3"].dtype == torch.float32
                and MSDA.f16_products_now() in (1, 3) and MSDA.gates_applicable(h.shape[0], self.rnn_size))

    def cell_part(self, g_x, g_h, xt_gates, c, const):
        """(gate parts, input token) -> (h', c')"""
        if not isinstance(xt_gates, tuple):                             # per-row pre-activations given directly
            if const.get("gate_perm") is not None:
                xt_gates = xt_gates[:, const["gate_perm"]]
            xt_gates = (xt_gates.contiguous(), torch.arange(xt_gates.shape[0], device=xt_gates.device))
        emb_gates, it = xt_gates                                        # (table (V+1,4H), token ids)
        if isinstance(g_h, tuple):                                      # (attend_part: the planes of h)
            return MSDA.gemm_f16x3_gates(g_x, g_h[0], const["w_gate_cat_p"], const["gates_hs"], emb_gates, it, c,
                                         need_h=not const.get("h_planes_only", False))
        if const.get("gate_perm") is not None:                          # g_x: planes of the attended feature (attend_part)
            return MSDA.gemm_f16x3_lstm(g_x, const["w_att_p"], g_h[:, self.att_hid_size:], const["gates_hs"], emb_gates,
                                        it, c)
        return MSDA.lstm_cell(g_x, g_h[:, self.att_hid_size:], emb_gates, it, c, gates_c=const["gates_hs"],
                              planes="w_h_cat_p" in const and g_x.dtype == torch.float32)

    def step(self, xt_gates, state, query, reference_points, temporal_shapes, level_start_index, const):
        """one token.  xt_gates = embed(it) @ W_ih[:, :E]^T  (B*Q, 4H)"""
        att = self.deformable_att
        B, Q, L, _ = reference_points.shape
        h, c = state                                                                # (B*Q, H)
        K = self.n_levels * self.n_points
        C = self.att_feat_size
        E = self.input_encoding_size
        if "w_off_h" in const and not torch.is_grad_enabled():
            g_x, g_h = self.attend_part(h, reference_points, temporal_shapes, level_start_index, const)
            h2, c2 = self.cell_part(g_x, g_h, xt_gates, c, const)
            return h2, (h2, c2)
        if isinstance(xt_gates, tuple):                                   # (pre-multiplied table, token ids)
            xt_gates = xt_gates[0].index_select(0, xt_gates[1])
        off = const["off_hs"] + F.linear(h, att.sampling_offsets.weight[:, :self.rnn_size]).view(B, Q, K)
        off = off.view(B, Q, 1, self.n_levels, self.n_points)
        if reference_points.shape[-1] == 1:
            x = reference_points[:, :, None, :, None, 0] + off / temporal_shapes[None, None, None, :, None]
        else:
            x = reference_points[:, :, None, :, None, 0] \
                + off / self.n_points * reference_points[:, :, None, :, None, 1] * 0.5
        loc = torch.stack((x, torch.full_like(x, 0.5)), -1).contiguous()
        shapes2d = temporal_shapes_2d(temporal_shapes, level_start_index)
        samp = MSDASampleFunction.apply(const["slab"], shapes2d, level_start_index, loc, "border")     # (B, C+A, Q, L, P)
        samp = samp.view(B, C + self.att_hid_size, Q, K).permute(0, 2, 3, 1).reshape(B * Q, K, C + self.att_hid_size)
        clip, att_ctx = samp[..., :C], samp[..., C:]
        dot = torch.tanh(att_ctx + self.h2att(h)[:, None, :])
        e = self.alpha_net(dot).squeeze(-1)                                         # (B*Q, K)
        alpha = F.softmax(e, dim=1)
        att_res = torch.bmm(alpha.unsqueeze(1), clip).squeeze(1)                    # (B*Q, C)
        gates = xt_gates + const["gates_hs"] + F.linear(att_res, self.rnn.weight_ih_l0[:, E:E + C]) \
            + F.linear(h, self.rnn.weight_hh_l0)
        i, f, g, o = gates.chunk(4, 1)
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h2 = torch.sigmoid(o) * torch.tanh(c2)
        return h2, (h2, c2)

    def forward(self, xt, state, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask):
        """Reference call form (LSTM_DSA.py:241): state = (h (1,N,H), c (1,N,H)); returns (output, state)."""
        const = self.prepare(query, input_flatten, input_padding_mask)
        xt_gates = F.linear(xt.reshape(-1, xt.shape[-1]), self.rnn.weight_ih_l0[:, :self.input_encoding_size])
        out, (h, c) = self.step(xt_gates, (state[0][-1].contiguous(), state[1][-1].contiguous()), query,
                                reference_points, input_spatial_shapes, input_level_start_index, const)
        return out, (h[None], c[None])


class _CapSlab(torch.autograd.Function):
    """The captioner's sampling slab [value_proj(memory) with padded rows zeroed | ctx2att(of that)] (B, S, C + A) in TRAINING as one
    node (ms_deform_attn_for_caption.py:98-101 + LSTM_DSA.py:253, ctx2att pushed through the sampling): both products write their
    half of the slab in place (row mask and row maxima in the first one's epilogue) -- no masked copy, no row-maximum pass, no cat;
    backward: the ctx2att input gradient lands ON the value half's gradient in the product's epilogue (no add), the padded rows'
    zeroing and the row maxima in one pass, both weight gradients through the backward pass's grouped launches."""

    @staticmethod
    def forward(ctx, mem, mask_u8, wv, bv, wc, bc):
        from .. import layers as L
        from .. import train_layers as TL
        from ..linear import _operands, _row_amax
        B, S, C = mem.shape
        R, Cf, A = B * S, wv.shape[0], wc.shape[0]
        x2 = mem.reshape(R, C)
        if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        am_x = _row_amax(x2, mem)
        opv, opv_t = _operands((wv,), (bv,))
        opc, opc_t = _operands((wc,), (bc,))
        slab = torch.empty(B, S, Cf + A, device=mem.device, dtype=torch.float32)
        s2 = slab.view(R, Cf + A)
        am_v = TL.step_zeros(R, mem.device)
        L.linear(x2, opv, [L.seg(0, s2[:, :Cf], am_x, rowmask=mask_u8, amax_out=am_v)])
        L.linear(s2[:, :Cf], opc, [L.seg(0, s2[:, Cf:], am_v)])
        ctx.save_for_backward(x2, am_x, slab, am_v, mask_u8 if mask_u8 is not None else x2.new_empty(0), wv, wc)
        ctx.ops, ctx.params, ctx.has_mask, ctx.in_shape = (opv_t, opc_t), (wv, bv, wc, bc), mask_u8 is not None, mem.shape
        return slab

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from .. import _lib
        from .. import layers as L
        from ..linear import queued_wgrad
        x2, am_x, slab, am_v, mask_u8, wv, wc = ctx.saved_tensors
        opv_t, opc_t = ctx.ops
        R, Cf = x2.shape[0], wv.shape[0]
        g2 = g.reshape(R, slab.shape[-1])
        if g2.stride(1) != 1 or g2.stride(0) % 4 or g2.data_ptr() % 16:
            g2 = g2.contiguous()
        g_v, g_a = g2[:, :Cf], g2[:, Cf:]
        am_ga = L.row_absmax(g_a)[0]
        dv = torch.empty(R, Cf, device=g.device, dtype=torch.float32)
        L.linear(g_a, opc_t, [L.seg(0, dv, am_ga, resid=g_v)])                    # d value = d(value half) + d(ctx2att half) Wc
        if ctx.has_mask:
            dvm = torch.empty_like(dv)
            am_dv = torch.empty(R, device=g.device, dtype=torch.float32)
            with torch.cuda.device(g.device):
                rc = _lib.lib().gvl_mask_rows_backward_f32(dv.data_ptr(), mask_u8.data_ptr(), R, Cf, dvm.data_ptr(), am_dv.data_ptr(),
                                                           torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "mask_rows_backward")
        else:
            dvm, am_dv = dv, L.row_absmax(dv)[0]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(R, x2.shape[1], device=g.device, dtype=torch.float32)
            L.linear(dvm, opv_t, [L.seg(0, dx, am_dv)])
            dx = dx.view(ctx.in_shape)
        s2 = slab.view(R, slab.shape[-1])
        gwc, gbc = queued_wgrad(g_a, s2[:, :Cf], am_ga, am_v, True, (ctx.params[2], ctx.params[3]), True)
        gwv, gbv = queued_wgrad(dvm, x2, am_dv, am_x, True, (ctx.params[0], ctx.params[1]), True)
        return dx, None, gwv, gbv, gwc, gbc


def _cap_slab_eligible(core, mem, mask):
    from ..linear import train_linear_eligible
    att = core.deformable_att
    vp, ca = att.value_proj, core.ctx2att
    return (mem.dim() == 3 and vp.bias is not None and ca.bias is not None and att.n_heads == 1
            and train_linear_eligible(mem, (vp.weight,), (vp.bias,))
            and train_linear_eligible(mem.new_empty(1).expand(mem.shape[0] * mem.shape[1], vp.weight.shape[0]), (ca.weight,), (ca.bias,))
            and (mask is None or (mask.dtype == torch.bool and mask.shape == mem.shape[:2]))
            and vp.weight.shape[0] % 4 == 0 and os.environ.get("GVL_CAP_SLAB", "") != "torch")


def _bf16_decode(fn):
    """Greedy / multinomial decoding under torch.autocast(bfloat16): the token loop's GEMMs run in bf16 and the three
    token-step kernels read their bf16 outputs directly (gvl_cap_attend_bf16, gvl_lstm_cell_bf16, gvl_greedy_step_bf16);
    positions (reference points, valid ratios, offsets) and the recurrent state stay fp32.  Any other autocast type
    falls back to the fp32 island."""
    import functools
    island = _fp32_island(fn)

    @functools.wraps(fn)
    def wrapped(self, hs, reference, others, *args, **kwargs):
        if not torch.is_autocast_enabled():
            return fn(self, hs, reference, others, *args, **kwargs)
        if torch.get_autocast_dtype("cuda") != torch.bfloat16 or torch.is_grad_enabled():
            return island(self, hs, reference, others, *args, **kwargs)
        others = dict(others)
        others["valid_ratios"] = others["valid_ratios"].float()
        return fn(self, hs, reference.float(), others, *args, **kwargs)
    return wrapped


class TeacherForcedLoop(torch.autograd.Function):
    """All teacher-forced token steps of the LSTM-DSA core (LSTM_DSA.py:63-117 loop over :241-271) as ONE autograd
    node -> hidden states (n, steps, H).

    Why: in training the captioner sees only the matched queries (n = 48 rows at cfg A), so the reference formulation is
    ~64 launch-bound kernels per token forward and ~150 backward, and autograd re-accumulates the 12 MB gradient of
    the [value | ctx2att(value)] slab after every token.  Here a token is GEMM, k_cap_train_fwd, GEMM,
    k_lstm_train_fwd; its backward k_lstm_train_bwd, GEMM, k_cap_train_bwd, GEMM (gvl_amd/csrc/gvl_cap_train.hip);
    slab / reference / alpha_net gradients accumulate in place across the steps, and every weight gradient is ONE
    GEMM over all steps after the loop (the weights are constant across the loop, so sum_i dY_i^T X_i = dY^T X).

    Inputs: slab (B,S,2C); ref_in (B,Q,L,RD); off_hs (n,16) and gates_hs (n,4H) the token-independent parts of the
    offsets / gate pre-activations; xt_all (n,steps,4H) embedding part of the gates; w_hcat (A+4H+16, H) =
    [h2att.weight; W_hh; sampling_offsets.weight[:, :H]] with bias b_hcat; w_att (4H, C) = W_ih[:, E:E+C];
    alpha_w (A,), alpha_b (1,).
    time_major: xt_all is (steps, n, 4H) and the result (steps, n, H) -- the layout the loop keeps its per-step buffers in, so
    neither the result nor the two gradients crossing the node's boundary (9 + 9 + 18 MB at 192 rows x 23 tokens) is transposed
    into a copy; the caller orders its token ids / targets / weights (steps, n) instead, which are a few KB."""

    @staticmethod
    def forward(ctx, slab, ref_in, off_hs, gates_hs, xt_all, w_hcat, b_hcat, w_att, alpha_w, alpha_b, shapes2d, lsi,
                n_levels, n_points, row_video=None, time_major=False):
        if time_major:
            steps, n, H4 = xt_all.shape
        else:
            n, steps, H4 = xt_all.shape
        H = H4 // 4
        C = w_att.shape[1]
        W = w_hcat.shape[0]
        A = W - H4 - 16
        new = lambda *shape: torch.empty(shape, device=slab.device, dtype=torch.float32)      # noqa: E731
        # the token-independent gate part joins the per-token embedding part ONCE (one pass over (n, steps, 4H)); as the
        # matrix addend of the attention product below it cost a (n, 4H) copy into `out` per token (addmm with beta = 1)
        xt_all = xt_all + (gates_hs[None] if time_major else gates_hs[:, None, :])
        g_h, hc_all = new(steps, n, W), new(2, steps + 1, n, H)
        h_all, c_all = hc_all[0], hc_all[1]
        att, alpha, act, g_x = new(steps, n, C), new(steps, n, 16), new(steps, n, H4), new(n, H4)
        hc_all[:, 0].zero_()                                           # (h_0 = c_0 = 0: one fill for both)
        w_hcat_t, w_att_t = w_hcat.t(), w_att.t()
        for i in range(steps):
            torch.addmm(b_hcat, h_all[i], w_hcat_t, out=g_h[i])                   # [h2att(h) | h W_hh^T | offsets(h)]
            MSDA.cap_attend_train_forward(slab, shapes2d, lsi, ref_in, off_hs, g_h[i][:, A + H4:], g_h[i][:, :A],
                                          alpha_w, alpha_b, n_levels, n_points, att_res=att[i], alpha_out=alpha[i],
                                          row_video=row_video)
            torch.mm(att[i], w_att_t, out=g_x)                                    # attention part of W_ih x
            MSDA.lstm_cell_train_forward(g_x, g_h[i][:, A:A + H4], xt_all[i] if time_major else xt_all[:, i], c_all[i], act[i],
                                         h_all[i + 1], c_all[i + 1])
        ctx.save_for_backward(slab, ref_in, off_hs, w_hcat, w_att, alpha_w, shapes2d, lsi, g_h, h_all, c_all, att,
                              alpha, act)
        ctx.cfg = (n_levels, n_points, A)
        ctx.row_video, ctx.time_major = row_video, time_major
        return h_all[1:] if time_major else h_all[1:].permute(1, 0, 2).contiguous()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_hidden):
        slab, ref_in, off_hs, w_hcat, w_att, alpha_w, shapes2d, lsi, g_h, h_all, c_all, att, alpha, act = \
            ctx.saved_tensors
        n_levels, n_points, A = ctx.cfg
        steps, n, W = g_h.shape
        H, C = h_all.shape[-1], att.shape[-1]
        H4 = 4 * H
        new = lambda *shape: torch.empty(shape, device=slab.device, dtype=torch.float32)      # noqa: E731
        d_h = (d_hidden if ctx.time_major else d_hidden.permute(1, 0, 2)).contiguous()
        dg = new(steps, n, W)                          # per step [d h2att(h) | d gates | d offsets]: fully overwritten
        g_slab = torch.zeros_like(slab)
        n_r, n_a = (ref_in.numel() + 3) // 4 * 4, (alpha_w.numel() + 3) // 4 * 4      # (16-byte aligned pieces)
        small = slab.new_zeros(n_r + n_a + 4)                          # the three small accumulators: one fill
        g_ref = small[:ref_in.numel()].view(ref_in.shape)
        g_aw = small[n_r:n_r + alpha_w.numel()].view(alpha_w.shape)
        g_ab = small[n_r + n_a:n_r + n_a + 1]
        d_att, dh_carry, dc = new(n, C), None, None
        d_gates_sum = new(n, H4)                       # sum over the steps of the gate gradients: kept by the cell's backward kernel
        dh_buf, dc_buf = (new(n, H), new(n, H)), (new(n, H), new(n, H))
        for i in range(steps - 1, -1, -1):
            dgates = dg[i][:, A:A + H4]
            MSDA.lstm_cell_train_backward(d_h[i], dh_carry, dc, act[i], c_all[i], c_all[i + 1], dgates, dc_buf[i & 1],
                                          gates_sum=d_gates_sum, first=i == steps - 1)
            dc = dc_buf[i & 1]
            torch.mm(dgates, w_att, out=d_att)
            MSDA.cap_attend_train_backward(slab, shapes2d, lsi, ref_in, off_hs, g_h[i][:, A + H4:], g_h[i][:, :A],
                                           alpha_w, alpha[i], d_att, n_levels, n_points, g_slab, dg[i][:, :A],
                                           dg[i][:, A + H4:], g_ref, g_aw, g_ab, row_video=ctx.row_video)
            if i > 0:                                  # h_{-1} = 0 is a constant
                dh_carry = torch.mm(dg[i], w_hcat, out=dh_buf[i & 1])
        dgf = dg.view(steps * n, W)
        d_gates = dg[:, :, A:A + H4]
        h_prev, att2 = h_all[:steps].reshape(steps * n, H), att.view(steps * n, C)
        from ..linear import train_linear_enabled
        if train_linear_enabled() and steps * n >= 512 and MSDA.wgrad_eligible(dgf, h_prev) and C % 4 == 0:
            # the two weight gradients over ALL steps on the fp16 matrix cores (gvl_wgrad_f16x3_f32; the bias gradient comes
            # out of the same pass over dg): 84 + 6 + 50 us as fp32 library GEMMs + column sum at (4416, 2576 | 2048, 512)
            from .. import layers as L
            am_dg = L.row_absmax(dgf)[0]
            d_w_hcat, d_b_hcat = MSDA.wgrad(dgf, h_prev, am_dg, L.row_absmax(h_prev)[0])
            d_w_att, _ = MSDA.wgrad(dgf[:, A:A + H4], att2, am_dg, L.row_absmax(att2)[0], want_bias=False)
        else:
            d_w_hcat = dgf.t().mm(h_prev)
            d_b_hcat = MSDA.col_sum(dgf)
            d_w_att = d_gates.reshape(steps * n, H4).t().mm(att2)
        return (g_slab, g_ref, dg[:, :, A + H4:].sum(0), d_gates_sum, d_gates if ctx.time_major else d_gates.permute(1, 0, 2),
                d_w_hcat, d_b_hcat, d_w_att, g_aw, g_ab, None, None, None, None, None, None)


def _fp32_island(fn):
    """Mixed-precision policy of gvl_amd under torch.autocast(bfloat16) (BASELINE config 4): the transformer runs its
    GEMMs / deformable attention on bf16 storage, the recurrent captioner stays fp32 -- its token-step kernels
    (gvl_cap_attend_f32, gvl_lstm_cell_f32, gvl_row_argmax_lse_f32) are fp32, the recurrence feeds its own rounding
    back 30 times, and greedy argmax over ~8.5k logits flips on bf16 near-ties.  Inputs arriving in bf16 are widened."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, hs, reference, others, *args, **kwargs):
        if not torch.is_autocast_enabled():
            return fn(self, hs, reference, others, *args, **kwargs)
        others = dict(others)
        for k in ("memory", "valid_ratios"):
            others[k] = others[k].float()
        with torch.autocast("cuda", enabled=False):
            return fn(self, hs.float(), reference.float(), others, *args, **kwargs)
    return wrapped


class RowLoss:
    """what Captioner.forward(..., nll=(target, mask)) returns instead of log-probs when the loss never needs them"""
    __slots__ = ("row_loss",)

    def __init__(self, row_loss):
        self.row_loss = row_loss


class Captioner(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.vocab_size = opt.vocab_size
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_size = opt.rnn_size
        self.num_layers = opt.num_layers
        self.drop_prob_lm = opt.drop_prob
        self.max_caption_len = opt.max_caption_len
        self.ss_prob = 0.0
        self.embed = nn.Embedding(self.vocab_size + 1, self.input_encoding_size)
        self.logit = Linear(self.rnn_size, self.vocab_size + 1)
        self.dropout = nn.Dropout(self.drop_prob_lm)
        self.init_weights()

    def init_weights(self):
        self.embed.weight.data.uniform_(-0.1, 0.1)
        self.logit.bias.data.fill_(0)
        self.logit.weight.data.uniform_(-0.1, 0.1)

    def init_hidden(self, batch_size):
        w = next(self.parameters())
        return (w.new_zeros(self.num_layers, batch_size, self.rnn_size),
                w.new_zeros(self.num_layers, batch_size, self.rnn_size))

    def build_loss(self, input, target, mask):
        """LSTM_DSA.py:48-52.  The reference multiplies by a materialised one-hot (n, len, vocab+1) tensor and sums the
        vocabulary axis: a sum of one term and zeros, i.e. exactly the gathered log-prob -- taken directly here."""
        if isinstance(input, RowLoss):                                  # forward(..., nll=(target, mask)) already did it
            return input.row_loss
        max_len = input.shape[1]
        picked = input.gather(2, target[:, :max_len, None]).squeeze(2)
        return -(picked * mask[:, :max_len]).sum(1) / (mask.sum(1) + 1e-6)

    def _scaled_reference(self, reference, others):
        vr = others['valid_ratios']
        if reference.shape[-1] == 2:
            return reference[:, :, None] * torch.stack([vr] * 2, -1)[:, None]
        return reference[:, :, None] * vr[:, None, :, None]

    def _level_inputs(self, others, reference_points):
        n_levels = self.core.n_levels
        if n_levels < self.core.opt.num_feature_levels:
            # (the reference's own branch cannot run on temporal features: LSTM_DSA.py:151 takes torch.prod(dim=1) of the
            #  1-D (L,) tensor of level lengths and raises IndexError -- verified by running the reference with
            #  cap_num_feature_levels = 2; there is no behaviour to mirror)
            raise NotImplementedError("cap_num_feature_levels < num_feature_levels: the reference itself raises on this "
                                      "branch for temporal features (LSTM_DSA.py:151) and no config uses it")
        return (others['memory'], others['spatial_shapes'], others['level_start_index'], others['mask_flatten'],
                reference_points)

    def get_logprobs_state(self, it, state, query, reference_points, input_flatten, input_spatial_shapes,
                           input_level_start_index, mask):
        """LSTM_DSA.py:120-124 (single step, reference call form)."""
        xt = self.embed(it)
        output, state = self.core(xt, state, query, reference_points, input_flatten, input_spatial_shapes,
                                  input_level_start_index, mask)
        return F.log_softmax(self.logit(self.dropout(output)), dim=1), state

    @_fp32_island
    def forward(self, hs, reference, others, cap_tensor, steps=None, row_video=None, nll=None):
        """Teacher-forced log-probs (LSTM_DSA.py:63-117) -> (B*Q, steps, vocab+1).
        nll = (target, mask): the caller only wants build_loss(log-probs, target, mask) -- on the fused teacher-forced path
        that loss is returned directly as RowLoss (gvl_amd.linear.vocab_nll: the (n, steps, vocab+1) log-prob tensor, its
        gather and their backward chain never exist); otherwise the log-probs as always.
        row_video (n,) int64: the COMPACT row form of the layout-independent train step -- hs (n, C) and reference (n, RD)
        are rows of any video (row_video[r], negative = unused row of the fixed-capacity row set) instead of (B, Q, .)."""
        seq = cap_tensor.long()
        if row_video is not None:
            if not (torch.is_grad_enabled() and self.core.fused_train_eligible(hs)
                    and not (self.training and self.ss_prob > 0.0)):
                raise RuntimeError("compact caption rows need the fused teacher-forced path")
            vr = others['valid_ratios'][row_video.clamp(min=0)]                      # (n, L)
            ref_in = reference[:, None, :] * (torch.stack([vr] * 2, -1) if reference.shape[-1] == 2 else vr[..., None])
            n = hs.shape[0]
        else:
            vid_num, query_num, _ = hs.shape
            ref_in = self._scaled_reference(reference, others)
            n = vid_num * query_num
        memory, tshapes, lsi, mask, ref_in = self._level_inputs(others, ref_in)
        const = self.core.prepare(hs, memory, mask)
        h = c = None                                   # (zero states: built where the step-by-step loop below needs them)
        w_x = const["w_ih_parts"][0] if "w_ih_parts" in const else self.core.rnn.weight_ih_l0[:, :self.input_encoding_size]
        outputs = []
        # the reference leaves the loop at the first step i >= 1 whose input column is all <pad> (:110-112), testing
        # it on the host every step; the same cut computed with ONE read (or none: pass `steps` when the caller
        # already knows the caption lengths, e.g. under graph capture)
        if steps is None:
            live = (seq[:, 1:] != 0).any(0).cpu().tolist()
            steps = 1 + (live.index(False) if False in live else len(live))
            steps = min(steps, seq.size(1) - 1)
        if not (self.training and self.ss_prob > 0.0):
            # Pure teacher forcing: the inputs of every step are known, only h/c carry the recurrence.  Everything that
            # does not depend on it is batched over time -- the embedding part of the LSTM input GEMM before the loop,
            # the vocabulary GEMM + log_softmax after it -- so their weights see ONE forward and ONE gradient GEMM
            # instead of `steps` of each followed by `steps` accumulations of a 17 MB gradient.
            # embedding rows by index_select: its backward is an atomic index_add.  nn.Embedding's backward switches to
            # a rocPRIM radix sort above 3072 indices (512 padded rows x 11 steps), and that path faulted when replayed
            # from a hipGraph on MI355X / ROCm 7.2 (eager was fine)
            fused = torch.is_grad_enabled() and self.core.fused_train_eligible(hs)
            # loss-only on the fused path: everything per token is laid out (steps, n, .) as the token loop keeps it, and the
            # few KB of ids / targets / weights are transposed instead of the 9-18 MB activations and gradients
            tm = (fused and nll is not None and vocab_nll_eligible(hs, self.logit.weight, self.logit.bias)
                  and os.environ.get("GVL_CAP_TIME_MAJOR", "1") != "0")
            ids = seq[:, :steps].t() if tm else seq[:, :steps]
            from .. import layers as _layers
            emb = _layers.embed_rows(self.embed.weight, ids.reshape(-1)).view(*ids.shape, -1)
            mirror = self.core.__dict__.get("_gvl_wx_mirror")
            if (mirror is not None and torch.is_grad_enabled() and os.environ.get("GVL_XT_MIRROR", "1") != "0"
                    and _mirror_ok(emb, w_x, mirror)):
                # the embedding part of every token's gate pre-activations on the hand-written products (planes of W_ih's
                # embedding columns: gvl_amd/train_planes.py register_mirror) instead of the fp32 library GEMMs
                xt_all = _mirror_linear(emb, w_x, mirror)
            else:
                xt_all = F.linear(emb, w_x)                                       # (n, steps, 4H); tm: (steps, n, 4H)
            if fused:
                hidden = self.core.teacher_forced(xt_all, hs, ref_in, tshapes, lsi, const, row_video, time_major=tm)
                if tm:
                    target, tmask = nll
                    picked = vocab_nll(self.dropout(hidden), self.logit.weight, self.logit.bias, target[:, :steps].t(),
                                       tmask[:, :steps].t()).view(steps, n)
                    return RowLoss(-picked.sum(0) / (tmask.sum(1) + 1e-6))               # build_loss (:48-52)
                if nll is not None and vocab_nll_eligible(hidden, self.logit.weight, self.logit.bias):
                    target, tmask = nll
                    picked = vocab_nll(self.dropout(hidden), self.logit.weight, self.logit.bias, target[:, :steps],
                                       tmask[:, :steps]).view(n, steps)
                    return RowLoss(-picked.sum(1) / (tmask.sum(1) + 1e-6))
                return F.log_softmax(split_linear(self.dropout(hidden), self.logit.weight, self.logit.bias), dim=2)
            hidden = []
            h, c = hs.new_zeros(n, self.rnn_size), hs.new_zeros(n, self.rnn_size)
            for i in range(steps):
                out, (h, c) = self.core.step(xt_all[:, i], (h, c), hs, ref_in, tshapes, lsi, const)
                hidden.append(out)
            hidden = torch.stack(hidden, 1)                                       # (n, steps, H)
            return F.log_softmax(split_linear(self.dropout(hidden), self.logit.weight, self.logit.bias), dim=2)
        h, c = hs.new_zeros(n, self.rnn_size), hs.new_zeros(n, self.rnn_size)
        for i in range(steps):                                                    # scheduled sampling (:92-105)
            it = seq[:, i].clone()
            if i >= 1:
                prob = hs.new_zeros(n).uniform_(0, 1)
                take = prob < self.ss_prob
                if take.any():
                    ind = take.nonzero().view(-1)
                    prev = torch.exp(outputs[-1].detach())
                    it.index_copy_(0, ind, torch.multinomial(prev, 1).view(-1).index_select(0, ind))
            out, (h, c) = self.core.step(F.linear(self.embed(it), w_x), (h, c), hs, ref_in, tshapes, lsi, const)
            outputs.append(F.log_softmax(self.logit(self.dropout(out)), dim=1))
        return torch.stack(outputs, 1)

    def _embedding_gates(self, perm=None):
        """embedding table pre-multiplied by its slice of W_ih ((V+1, 4H); a 17.9 GFLOP GEMM at vocabulary 8517):
        depends on weights only -> rebuilt when they change (or the autocast type does), not per forward.  perm: gate
        columns in the order 4 unit + gate (the fused cell of ShowAttendTellCore._inference_weights)"""
        w_e, w_ih = self.embed.weight, self.core.rnn.weight_ih_l0
        ac = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else torch.float32
        key = (w_e.data_ptr(), w_e._version, w_ih.data_ptr(), w_ih._version, ac, perm is not None)
        cached = getattr(self, "_emb_gates", None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                w_x = w_ih[:, :self.input_encoding_size]
                cached = self._emb_gates = (key, F.linear(w_e, w_x if perm is None else w_x[perm]).contiguous())
        return cached[1]

    def _logit_planes(self, out):
        """fp16 planes of the vocabulary layer's weight (gvl_split_rows_f16), rebuilt when the weight changes; None when
        the split product does not apply (training / dropout active, autocast, GVL_GEMM=f32)"""
        w = self.logit.weight
        if (self.training or torch.is_grad_enabled() or torch.is_autocast_enabled() or out.dtype != torch.float32
                or hasattr(out, "_gvl_lowp") or not split_gemm_enabled() or not MSDA.split_eligible(w)):
            return None
        key = (w.data_ptr(), w._version)
        cached = getattr(self, "_logit_p", None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                cached = self._logit_p = (key, MSDA.split_rows(w.detach()))
        return cached[1]

    def _decode_device(self, hs, reference, memory, mask, valid_ratios, tshapes, lsi, sample_max, temperature):
        """The whole decoding loop on the device, no host interaction: -> (seq (n, T), logprob (n, T), alive (T,))
        with T = max_caption_len.  Capturable in a hipGraph."""
        n = hs.shape[0] * hs.shape[1]
        ref_in = self._scaled_reference(reference, {'valid_ratios': valid_ratios})
        const = self.core.prepare(hs, memory, mask)
        # embedding rows pre-multiplied by their slice of W_ih: one gather per step instead of a GEMM
        emb_gates = self._embedding_gates(const.get("gate_perm"))
        # the recurrent state is fp32, also under autocast; h, c (and the greedy loop's log-prob table) share ONE zero fill
        # ... and so do the greedy loop's integer state (<bos> tokens, the token table) and its `alive` flags: byte ranges of it
        T_ = T = self.max_caption_len
        n_f = 2 * n * self.rnn_size + n * T_
        b_f = (4 * n_f + 15) // 16 * 16
        b_l = (8 * (n + n * T) + 15) // 16 * 16
        raw = torch.zeros(b_f + b_l + T, dtype=torch.uint8, device=hs.device)
        zbuf = raw[:4 * n_f].view(torch.float32)
        longs = raw[b_f:b_f + 8 * (n + n * T)].view(torch.long)
        h = zbuf[:n * self.rnn_size].view(n, self.rnn_size)
        c = zbuf[n * self.rnn_size:2 * n * self.rnn_size].view(n, self.rnn_size)
        it = longs[:n]                                                                # <bos>
        if sample_max:
            # greedy: argmax, log-prob and the per-step bookkeeping (unfinished / alive / seq / seq_lp) are one kernel.
            # seq starts as zeros: a loop that is cut short (decode_stop / decode_continue below) leaves exactly what
            # the full loop would have written after every row has ended (seq = token * unfinished, :183-188)
            st = {"hs": hs, "ref_in": ref_in, "tshapes": tshapes, "lsi": lsi, "const": const, "emb_gates": emb_gates,
                  "h": h, "c": c, "it": it, "logits": None,
                  "unfinished": torch.empty(n, dtype=torch.uint8, device=hs.device),
                  "seq": longs[n:].view(n, T),
                  "seq_lp": zbuf[2 * n * self.rnn_size:].view(n, T),
                  # alive[t]: some row is still unfinished after token t -- set by the greedy kernel itself
                  "alive": raw[b_f + b_l:]}
            # (running the vocabulary GEMM + argmax of token t on a second stream beside the token-independent half of
            #  step t+1 was tried -- fork / join inside the captured graph -- and measured no gain: 742-757 vs 750
            #  videos/s; the GEMMs already occupy every CU)
            stop = getattr(self, "decode_stop", None)             # iterations [0, stop) now, the rest by decode_continue
            stop = T + 1 if stop is None else max(1, min(int(stop), T + 1))
            self._greedy_iterations(st, 0, stop)
            self._decode_state = st if stop < T + 1 else None
            # a row is unfinished at step t exactly while its tokens are non-zero (seq = token * unfinished, :183-188)
            return st["seq"], st["seq_lp"], st["alive"].view(torch.bool)
        unfinished = torch.ones(n, dtype=torch.bool, device=hs.device)
        seq, seq_lp, alive = [], [], []
        for t in range(T + 1):
            if t > 0:
                logprobs = F.log_softmax(logits, dim=1)
                prev = torch.exp(logprobs) if temperature == 1.0 else torch.exp(logprobs / temperature)
                it = torch.multinomial(prev, 1)
                lp = logprobs.gather(1, it).view(-1)
                it = it.view(-1)
            if t < T:
                out, (h, c) = self.core.step((emb_gates, it), (h, c), hs, ref_in, tshapes, lsi, const)
                logits = self.logit(self.dropout(out))
            if t >= 1:
                unfinished = (it > 0) if t == 1 else (unfinished & (it > 0))
                alive.append(unfinished.any())
                seq.append(it * unfinished.type_as(it))
                seq_lp.append(lp)
        return torch.stack(seq, 1), torch.stack(seq_lp, 1), torch.stack(alive)

    def _greedy_iterations(self, st, t0, t1):
        """iterations [t0, t1) of the greedy loop (LSTM_DSA.py:162-190) on the state dict of _decode_device: iteration t
        first books token t - 1 from the logits of the previous iteration, then (t < max_caption_len) runs the LSTM
        step + vocabulary logits for token t.  (The reference also evaluates that step after the LAST token,
        :189-190, and then leaves the loop without reading it.)"""
        T = self.max_caption_len
        const = st["const"]
        if "h_planes_only" not in const:
            # every reader of h' in this loop takes its operand planes (the vocabulary product, h2att(h) + offsets, the next gate
            # product): the fp32 copy (a third of the gate kernel's stores) is then not written
            const["h_planes_only"] = bool("w_off_h" in const and _GREEDY_MERGED and self.core.attend_pre_ok(const, st["tshapes"])
                                          and self.core.gates_in_one_launch(st["h"], const)
                                          and self._logit_planes(st["h"]) is not None)
        for t in range(t0, t1):
            if t > 0:
                hp = getattr(st["h"], "_gvl_planes", None)
                if (t < T and hp is not None and _GREEDY_MERGED and isinstance(st["logits"], MSDA.GreedyPartials)
                        and "w_off_h" in st["const"] and self.core.gates_in_one_launch(st["h"], st["const"])):
                    # the reduction of token t - 1 and h2att(h) of token t depend on different results of the last step: one launch
                    st["it"], st["h"]._gvl_h2att = MSDA.greedy_step_and_gemm(
                        st["logits"], t - 1, st["unfinished"], st["seq"], st["seq_lp"], st["alive"], hp,
                        st["const"]["w_h2att_p"], st["const"]["b_h2att"])
                else:
                    st["it"] = MSDA.greedy_step(st["logits"], t - 1, st["unfinished"], st["seq"], st["seq_lp"], st["alive"])
            if t < T:
                out, (st["h"], st["c"]) = self.core.step((st["emb_gates"], st["it"]), (st["h"], st["c"]), st["hs"],
                                                         st["ref_in"], st["tshapes"], st["lsi"], st["const"])
                planes = self._logit_planes(out)
                if planes is not None:                                   # fp32 product on the fp16 matrix cores
                    if getattr(out, "_gvl_planes", None) is None:        # (the cell kernel normally leaves them)
                        out._gvl_planes = MSDA.split_rows(out)           # the next step's h product reads the same planes
                    st["logits"] = MSDA.gemm_f16x3_argmax(out._gvl_planes, planes, self.logit.bias)   # never written out
                else:
                    st["logits"] = self.logit(self.dropout(getattr(out, "_gvl_lowp", out)))

    def decode_continue(self, t0, t1):
        """continue a greedy loop that `decode_stop` cut at iteration t0 (gvl_amd.parallel.GraphedEvalForward captures
        the loop in segments and stops replaying them once every caption has ended) -> alive flags of the tokens this
        segment booked, i.e. tokens [t0 - 1, t1 - 1)"""
        st = self._decode_state
        self._greedy_iterations(st, t0, t1)
        return st["alive"][t0 - 1:t1 - 1].view(torch.bool)

    def _decode_graphed(self, hs, reference, memory, mask, valid_ratios, tshapes, lsi):
        """Greedy decoding replayed from a hipGraph: the 31-step loop is ~800 kernel launches whose host-side issue
        cost exceeds their run time; captured once per input shape, replayed with one launch."""
        cache = self.__dict__.setdefault("_decode_graphs", {})
        # weight-derived operands are cached per parameter version (_inference_weights, _embedding_gates) and become
        # constants of the captured graph: a parameter update must therefore lead to a new capture
        key = (tuple(hs.shape), tuple(reference.shape), tuple(memory.shape), str(hs.device),
               tuple(tshapes._gvl_host_lengths[0]),
               torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None,
               sum(p_._version for p_ in self.parameters()))
        if cache and next(iter(cache))[-1] != key[-1]:
            cache.clear()                                    # parameters changed: graphs of the old weights are dead
        entry = cache.get(key)
        if entry is None:
            static = [t_.clone() for t_ in (hs, reference, memory, mask, valid_ratios)]
            side = torch.cuda.Stream(device=hs.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                       # warm-up off the capture (lazy inits, LDS attributes)
                self._decode_device(*static, tshapes, lsi, 1, 1.0)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):           # capture on the warm-up stream (see gvl_amd.parallel)
                outs = self._decode_device(*static, tshapes, lsi, 1, 1.0)
            entry = cache[key] = (graph, static, outs)
        graph, static, outs = entry
        for dst, src in zip(static, (hs, reference, memory, mask, valid_ratios)):
            dst.copy_(src)
        graph.replay()
        return outs

    @_bf16_decode
    def sample(self, hs, reference, others, opt={}):
        """Greedy / multinomial decoding (LSTM_DSA.py:126-194) -> (seq (B*Q, <=max_len), logprobs)."""
        sample_max = opt.get('sample_max', 1)
        temperature = opt.get('temperature', 1.0)
        memory, tshapes, lsi, mask, _ = self._level_inputs(others, reference)
        if getattr(self, "defer_trim", False):
            # caller is capturing the whole forward in a hipGraph (gvl_amd.parallel.GraphedEvalForward): no host read
            # here; the untrimmed sequences are returned and the caller cuts them at `last_alive` after the replay
            seq, seq_lp, self.last_alive = self._decode_device(hs, reference, memory, mask, others['valid_ratios'],
                                                               tshapes, lsi, sample_max, temperature)
            return seq, seq_lp
        use_graph = (getattr(self, "graph_decode", False) and sample_max and not torch.is_grad_enabled()
                     and getattr(tshapes, "_gvl_host_lengths", None) is not None)
        if use_graph:
            seq, seq_lp, alive = self._decode_graphed(hs, reference, memory, mask, others['valid_ratios'], tshapes, lsi)
        else:
            seq, seq_lp, alive = self._decode_device(hs, reference, memory, mask, others['valid_ratios'], tshapes,
                                                     lsi, sample_max, temperature)
        # the reference stops at the first step where every row has finished (:186-187): one host read here
        alive = alive.cpu().tolist()
        keep = alive.index(False) if False in alive else len(alive)
        if keep == 0:
            return [], []
        return seq[:, :keep].clone(), seq_lp[:, :keep].clone()


class LSTMDSACaptioner(Captioner):
    def __init__(self, opt):
        super().__init__(opt)
        self.core = ShowAttendTellCore(opt)
