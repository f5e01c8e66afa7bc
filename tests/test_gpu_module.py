"""GPU parity of the module mirrors (MSDeformAttn / MSDeformAttnCap) against goldens from the imported reference."""
import pytest
import torch

from helpers import load, t, module_state, maxerr, synth_array

pytestmark = pytest.mark.gpu


def strip(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items()}


@pytest.mark.parametrize("refdim", [1, 2])
@pytest.mark.parametrize("pad", ["zeros", "border"])
@pytest.mark.parametrize("fused", [False, True])
def test_msdeformattn_module_golden(refdim, pad, fused):
    """fused=True: level tensors carry host lengths -> the fused projection-epilogue + sampling kernels
    (gvl_msda1d_fused_*); fused=False: plain device tensors as the reference passes -> unfused op."""
    from gvl_amd.ops.modules import MSDeformAttn
    from gvl_amd.deformable_transformer import make_level_tensors
    from gvl_amd import _lib
    dev = torch.device("cuda:0")
    f = load(f"module_ref{refdim}")
    B, Q, C, M, L, P, _ = [int(v) for v in f["meta"]]
    S = int(f["tshapes"].sum())
    mod = MSDeformAttn(C, L, M, P, pad_mode=pad)
    mod.load_state_dict(strip(module_state("attn.", seed=100 + refdim), "attn."), strict=True)
    mod = mod.to(dev).eval()
    query = t(synth_array(f"mod{refdim}.query", (B, Q, C), 1)).to(dev).requires_grad_()
    inp = t(synth_array(f"mod{refdim}.input", (B, S, C), 1)).to(dev).requires_grad_()
    ref = t(synth_array(f"mod{refdim}.ref", (B, Q, L, refdim), 1, 0.05, 0.95))
    if refdim == 2:
        ref[..., 1] = ref[..., 1] * 0.5
    gout = t(synth_array(f"mod{refdim}.gout", (B, Q, C), 1)).to(dev)
    if fused:
        tsh, lsi = make_level_tensors(f["tshapes"].tolist(), dev)
        ref = ref.clone().requires_grad_()
    else:
        tsh, lsi = t(f["tshapes"]).to(dev), t(f["lsi"]).to(dev)
    ref_d = ref.to(dev)
    out = mod(query, ref_d, inp, tsh, lsi, t(f["mask"]).to(dev))
    assert _lib.lib().gvl_msda_last_impl() == (3 if fused else 2)
    assert maxerr(out, f[f"out_{pad}"]) < 1e-4
    out.backward(gout)
    assert maxerr(query.grad, f[f"gquery_{pad}"]) < 1e-3
    assert maxerr(inp.grad, f[f"ginput_{pad}"]) < 1e-3
    assert maxerr(mod.sampling_offsets.weight.grad, f[f"g_off_w_{pad}"]) < 2e-3
    assert maxerr(mod.attention_weights.bias.grad, f[f"g_aw_b_{pad}"]) < 1e-3
    assert maxerr(mod.value_proj.bias.grad, f[f"g_vproj_b_{pad}"]) < 1e-3


def test_msdeformattncap_module_golden():
    from gvl_amd.ops.modules import MSDeformAttnCap
    dev = torch.device("cuda:0")
    f = load("module_cap")
    B, Q, C, M, L, P, _ = [int(v) for v in f["meta"]]
    S = int(f["tshapes"].sum())
    mod = MSDeformAttnCap(C, L, 1, P)
    mod.load_state_dict(strip(module_state("cap.", M=1, qdim=2 * C, seed=200), "cap."), strict=True)
    mod = mod.to(dev).eval()
    query = t(synth_array("cap.query", (B, Q, 2 * C), 1)).to(dev)
    inp = t(synth_array("cap.input", (B, S, C), 1)).to(dev)
    ref = t(synth_array("cap.ref", (B, Q, L, 2), 1, 0.05, 0.95))
    ref[..., 1] *= 0.5
    with torch.no_grad():
        out = mod(query, ref.to(dev), inp, t(f["tshapes"]).to(dev), t(f["lsi"]).to(dev), t(f["mask"]).to(dev))
    assert maxerr(out, f["out"]) < 1e-4
