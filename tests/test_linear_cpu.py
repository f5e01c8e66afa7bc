"""Host-side dispatch of gvl_amd.linear (no GPU): which products go to the hand-written kernels and which to PyTorch."""
import torch
import torch.nn.functional as F


def test_split_linear_outside_the_kernel_domain_is_plain_linear():
    from gvl_amd.linear import split_linear, projection, linear
    g = torch.Generator().manual_seed(0)
    x, w, b = torch.randn(3, 7, 64, generator=g), torch.randn(10, 64, generator=g), torch.randn(10, generator=g)
    ref = F.linear(x, w, b)
    assert torch.equal(split_linear(x, w, b), ref)                       # CPU tensors: never the HIP path
    assert torch.equal(linear(x, w, b), ref)
    assert torch.equal(projection(x, w, b), ref)
    assert torch.equal(split_linear(x, w, None), F.linear(x, w))


def test_gemm_switch_follows_the_environment(monkeypatch):
    from gvl_amd.linear import split_gemm_enabled
    monkeypatch.delenv("GVL_GEMM", raising=False)
    assert split_gemm_enabled()
    monkeypatch.setenv("GVL_GEMM", "f32")
    assert not split_gemm_enabled()


def test_split_gemm_entry_points_are_declared_and_exported():
    """the ABI-6 symbols of include/gvl_msda.h resolve in the built library (no compute call: there is no GPU here)"""
    from gvl_amd import _lib
    L = _lib.lib()
    for name in ("gvl_split_rows_f16", "gvl_gemm_f16x3_f32", "gvl_gemm_f16x3_argmax_f32", "gvl_gemm_f16x3_argmax_chunks",
                 "gvl_greedy_step_partials_f32", "gvl_cap_attend_split_f32", "gvl_lstm_cell_split_f32"):
        assert hasattr(L, name), name
    assert L.gvl_gemm_f16x3_argmax_chunks(8518) == 134 and L.gvl_gemm_f16x3_argmax_chunks(1) == 2
    assert L.gvl_msda_abi_version() == _lib.ABI_VERSION >= 6


def test_inference_layer_entry_points_are_declared_and_exported():
    """the ABI-7 symbols (gvl_layers.hip) resolve; argument checking runs without a GPU (no launch is reached)"""
    from gvl_amd import _lib
    L = _lib.lib()
    for name in ("gvl_linear_f16x3_f32", "gvl_layer_norm_rows_f32", "gvl_row_absmax_f32", "gvl_box_refine_f32",
                 "gvl_count_head_f32", "gvl_msda1d_fused_forward_amax_f32"):
        assert hasattr(L, name), name
    assert L.gvl_msda_abi_version() == _lib.ABI_VERSION >= 7
    # K not a multiple of 32 / N not a multiple of 64 are refused before anything is launched
    assert L.gvl_linear_f16x3_f32(None, 0, None, 0, 0, 4, 48, None, None, None, None, 64, None, 1, 0, None) == -1
    assert b"K % 32" in L.gvl_last_error()
    assert L.gvl_layer_norm_rows_f32(None, 4, 6, None, None, 1e-5, None, 0, None, None, None, None) == -1
    assert L.gvl_box_refine_f32(None, 1, None, 2, None, 1, 1, 1, None, None, None) == -1


def test_round3_token_loop_entry_points_are_declared_and_exported():
    """the ABI-8 symbols (cell in the gate product's epilogue, attention kernel with host level starts, caption loss rows,
    few-row product) resolve; argument checking runs without a GPU; the gate permutation is what the header states"""
    import torch
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    from gvl_amd import _lib
    L = _lib.lib()
    for name in ("gvl_gemm_f16x3_lstm_f32", "gvl_cap_attend_split_levels_f32", "gvl_ce_rows_forward_f32",
                 "gvl_ce_rows_backward_f32"):
        assert hasattr(L, name), name
    assert L.gvl_msda_abi_version() == _lib.ABI_VERSION >= 8
    assert L.gvl_ce_rows_forward_f32(None, 4, 2, 8, None, None, None, None, None) == -1        # ld < V
    assert b"bad sizes" in L.gvl_last_error()
    assert L.gvl_gemm_f16x3_lstm_f32(None, None, None, 4, None, None, None, 6, 32, None, 24, None, 0, None, None, None, None,
                                     None, None, None, None, None) == -1                       # H not a multiple of 32
    assert b"multiple of 32" in L.gvl_last_error()
    perm = MSDA.gate_permutation(3)
    assert perm.tolist() == [0, 3, 6, 9, 1, 4, 7, 10, 2, 5, 8, 11]                             # row g * H + u at 4 * u + g
    w = torch.arange(12.0)[:, None]
    assert torch.equal(w[perm].view(3, 4), torch.tensor([[0.0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8, 11]]))


def test_row_maxima_tag_is_dropped_after_an_in_place_write():
    """row maxima ride on the producing tensor only for the version of the data they were computed from"""
    import torch
    from gvl_amd.layers import tag_amax, amax_of, enc_ref_of
    x = torch.randn(6, 8)
    am = x.abs().amax(1)
    tag_amax(x, am)
    assert amax_of(x, 6) is am and amax_of(x, 5) is None
    view = x[:]                                   # a view shares the version counter: the tag can be handed over
    view._gvl_amax = x._gvl_amax
    assert amax_of(view, 6) is am
    x.mul_(4.0)                                   # a mask / scale / hook between producer and consumer
    assert amax_of(x, 6) is None and amax_of(view, 6) is None
    vr = torch.ones(2, 4)
    vr._gvl_enc_ref = (torch.zeros(2, 3, 4, 1), vr._version)
    assert enc_ref_of(vr) is not None
    vr[0, 0] = 0.5
    assert enc_ref_of(vr) is None and enc_ref_of(torch.ones(2, 4)) is None


def test_autocast_policies_and_version_bump(monkeypatch):
    """host logic without a GPU: the autocast policy switches accept exactly their three values, and gvl_amd.optim.bump_versions
    moves the version counters the weight-derived caches are keyed on"""
    import pytest
    import torch
    from gvl_amd import pdvc
    from gvl_amd.optim import bump_versions
    for fn, var in ((pdvc.autocast_training_policy, "GVL_AUTOCAST_TRAINING"), (pdvc.autocast_inference_policy, "GVL_AUTOCAST_INFERENCE")):
        monkeypatch.delenv(var, raising=False)
        assert fn() == "f16"
        for v in ("fp32", "bf16", "f16"):
            monkeypatch.setenv(var, v)
            assert fn() == v
        monkeypatch.setenv(var, "fp8")
        if fn is pdvc.autocast_training_policy:
            with pytest.raises(ValueError):
                fn()
        else:
            assert fn() == "f16"                                # (the inference switch falls back to its default)
        monkeypatch.delenv(var, raising=False)
    ps = [torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.ones(2, 2))]
    before = [p._version for p in ps]
    bump_versions(ps)
    assert [p._version for p in ps] == [b + 1 for b in before]


def test_wgrad_queue_takes_a_second_gradient_of_the_same_parameter_at_once(monkeypatch):
    """gvl_amd.linear._WgradQueue (host logic, kernels stubbed): problems wait for the group; a parameter that already has a
    queued gradient in this backward pass is flushed and taken immediately (autograd adds the two as soon as the second is
    returned); a full group is flushed; the queue keeps detached aliases of the returned tensors"""
    import torch
    from gvl_amd import linear as GL
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    calls = []
    monkeypatch.setattr(MSDA, "wgrad_group_max", lambda: 3)
    monkeypatch.setattr(MSDA, "wgrad_group", lambda items: calls.append(("group", len(items))))
    monkeypatch.setattr(MSDA, "wgrad", lambda dy, x, a, b, grad_w=None, grad_b=None, want_bias=True, accumulate=False:
                        (calls.append(("single", 1)), (grad_w, grad_b))[1])
    q = GL._WgradQueue()
    dy, x, am = torch.zeros(8, 4), torch.zeros(8, 6), torch.zeros(8)
    w = [torch.nn.Parameter(torch.zeros(4, 6)) for _ in range(5)]
    gw0, gb0 = q.push(dy, x, am, am, True, [id(w[0])])
    assert calls == [] and tuple(gw0.shape) == (4, 6) and tuple(gb0.shape) == (4,)
    assert q.items[0][4].data_ptr() == gw0.data_ptr() and q.items[0][4] is not gw0          # a detached alias, not the tensor itself
    q.push(dy, x, am, am, False, [id(w[1])])
    q.push(dy, x, am, am, True, [id(w[0])])                    # the same parameter again: flush (2 queued), then at once
    assert calls == [("group", 2), ("single", 1)] and q.items == []
    for k in (2, 3, 4):
        q.push(dy, x, am, am, True, [id(w[k])])
    assert calls[-1] == ("group", 3) and q.items == []         # a full group leaves by itself
    lone = torch.nn.Parameter(torch.zeros(1))                  # (kept alive: a freed parameter's id() may be handed out again)
    q.push(dy, x, am, am, True, [id(lone)])
    q.flush()
    assert calls[-1] == ("single", 1)                          # a lone problem takes the single launch
    # a group another node filled exactly (train_mha._InProj appends two entries itself) leaves before the next problem joins
    q.items.extend([(dy, x, am, am, torch.zeros(4, 6), None)] * 3)
    fresh = torch.nn.Parameter(torch.zeros(1))
    q.push(dy, x, am, am, True, [id(fresh)])
    assert calls[-1] == ("group", 3) and len(q.items) == 1


def test_weighted_loss_sum_is_the_plain_weighted_sum_and_never_multiplies_an_unweighted_entry():
    """gvl_amd.criterion.weighted_loss_sum (train.py:403 as one dot product over the loss vectors): equals the term-by-term sum for
    tagged vector entries, loose scalars and a mix; a NaN in an UNWEIGHTED vector entry (loss_self_iou is 0/0 for a single match)
    does not reach the sum; gradients flow to the vectors; a changed weight is not served from the cache"""
    import torch
    from gvl_amd.criterion import unbind_tagged, weighted_loss_sum
    torch.manual_seed(0)
    table = torch.randn(12, requires_grad=True)
    caps = torch.randn(2, requires_grad=True)
    loose = torch.randn((), requires_grad=True)
    with torch.no_grad():
        table[4] = float("nan")                                        # an unweighted entry
    names = [f"t{i}" for i in range(12)]
    loss = dict(zip(names, unbind_tagged(table)))
    loss.update(zip(("cap", "cap_0"), unbind_tagged(caps)))
    loss["loose"] = loose * 2.0
    loss["not_a_loss"] = 3                                            # (ints and unweighted keys are ignored)
    wd = {n: 0.5 + 0.1 * i for i, n in enumerate(names) if i not in (4, 5)}
    wd.update(cap=2.0, cap_0=1.5, loose=0.25, absent=7.0)
    cache = {}
    got = weighted_loss_sum(loss, wd, cache)
    want = sum(loss[k].float() * wd[k] for k in loss if k in wd)
    assert torch.isfinite(got) and abs(float(got) - float(want)) <= 1e-5 * max(1.0, abs(float(want)))
    got.backward()
    g = table.grad.clone()
    assert g[4] == 0 and g[5] == 0 and abs(float(g[0]) - 0.5) < 1e-6 and abs(float(caps.grad[1]) - 1.5) < 1e-6
    assert abs(float(loose.grad) - 0.5) < 1e-6
    wd["t0"] = 9.0                                                    # (criterion.weight_dict is edited in place by gt_proposals mode)
    got2 = weighted_loss_sum(loss, wd, cache)
    assert abs(float(got2) - float(got) - (9.0 - 0.5) * float(table[0])) <= 1e-4
    assert float(weighted_loss_sum({"a": torch.tensor(1.0), "b": torch.tensor(2.0)}, {"a": 2.0, "b": 3.0})) == 8.0    # (few terms: plain)


def test_training_nodes_fall_back_to_their_pytorch_formulation_off_the_device():
    """the round-6 training nodes (gvl_amd/layers.py, train_layers.py) are eligibility-gated: on CPU tensors every one of them is
    its PyTorch formulation (nothing here may touch the library); the per-step snapshot of the step counter is taken once per
    open step and again after the next one opens"""
    import torch
    from gvl_amd import layers as L
    from gvl_amd import train_layers as TL
    x, sub = torch.randn(2, 5, 8, requires_grad=True), torch.randn(2, 5, 8)
    norm, drop = torch.nn.LayerNorm(8), torch.nn.Dropout(0.0)
    ys = TL.residual_dropout_norm(x, sub, drop, norm, fan=3)
    assert len(ys) == 3 and ys[0] is ys[1] is ys[2] and torch.allclose(ys[0], norm(x + sub))
    assert TL.residual_dropout_norm(x, sub, drop, norm).shape == x.shape
    emb = torch.randn(5, 16, requires_grad=True)
    assert not L.expand_parts_eligible(emb, 2) and not L.level_pos_embed_eligible(emb, [torch.randn(2, 16, 3)])
    assert not L.count_pool_train_eligible(x) and not L.class_count_heads_eligible(torch.nn.Linear(8, 1), x)
    mask = torch.tensor([[False, True, False, False, True]] * 2)
    y = torch.nn.Linear(8, 8)(x)
    assert torch.equal(L.mask_rows(y, mask), y.masked_fill(mask[..., None], 0.0))
    ids = torch.tensor([0, 3, 3, 1])
    assert torch.equal(L.embed_rows(emb, ids), emb.index_select(0, ids))
    TL.arena("cpu").buf = None
    a, b = TL.step_snapshot("cpu"), TL.step_snapshot("cpu")
    assert a is not b                                                  # no open step: a copy per call
    TL.arena("cpu").want = 64
    TL.arena_reset("cpu")
    a, b = TL.step_snapshot("cpu"), TL.step_snapshot("cpu")
    assert a is b                                                      # one per open step
    TL.arena_reset("cpu")
    assert TL.step_snapshot("cpu") is not a                            # the next step takes its own
    TL.arena("cpu").buf = None
