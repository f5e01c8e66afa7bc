"""BaseEncoder.forward_flat_train (the TRAINING pyramid on the hand-written kernels: _PyramidTrainFunction) against the PyTorch
formulation BaseEncoder.forward() + the flattening of DeformableTransformer.prepare_encoder_inputs (pdvc/base_encoder.py:60-80,
deformable_transformer.py:85-100): the flattened levels and the gradients of every conv / norm parameter."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,T,Cin,C,nl", [(16, 100, 512, 512, 4), (3, 37, 128, 256, 3), (5, 16, 64, 64, 4), (2, 512, 128, 512, 4),
                                          (2, 257, 64, 128, 4)])       # (long videos: a group norm unit walks hundreds of rows)
def test_training_pyramid_equals_the_pytorch_formulation(N, T, Cin, C, nl):
    from gvl_amd.base_encoder import BaseEncoder
    dev = torch.device("cuda:0")
    torch.manual_seed(N + T)
    enc = BaseEncoder(nl, Cin, C).to(dev).train()
    with torch.no_grad():
        for seq in enc.input_proj:                           # (the reference initialises the conv biases to 0 and the norms to 1 / 0)
            seq[0].bias.uniform_(-0.1, 0.1)
            seq[1].weight.uniform_(0.5, 1.5)
            seq[1].bias.uniform_(-0.2, 0.2)
    vf = torch.randn(N, T, Cin, device=dev)
    mask = torch.zeros(N, T, dtype=torch.bool, device=dev)
    mask[0, T // 2:] = True
    assert enc.flat_train_eligible(vf, mask)
    params = [p for seq in enc.input_proj for p in (seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias)]
    # reference: forward() builds (N, C, T_l) levels; flatten as prepare_encoder_inputs does
    dur = torch.full((N,), 100.0, device=dev)
    srcs, masks, _ = enc(vf, mask, dur)
    ref = torch.cat([s.transpose(1, 2) for s in srcs], 1)
    g = torch.randn_like(ref)
    gref = torch.autograd.grad(ref, params, g)
    out = enc.forward_flat_train(vf)
    assert out.shape == ref.shape
    scale = float(ref.detach().abs().max())
    assert float((out.detach() - ref.detach()).abs().max()) <= 2e-5 * scale
    gout = torch.autograd.grad(out, params, g)
    names = ["conv.weight", "conv.bias", "norm.weight", "norm.bias"]
    for i, (a, b) in enumerate(zip(gout, gref)):
        assert a.shape == b.shape
        tol = 2e-4 * max(1e-3, float(b.abs().max()))
        assert float((a - b).abs().max()) <= tol, (i // 4, names[i % 4], float((a - b).abs().max()), float(b.abs().max()))
    # masks / positions of the training geometry equal forward()'s
    masks2, poses2 = enc.train_geometry(vf, mask, dur)
    _, _, poses = enc(vf, mask, dur)
    for m1, m2, p1, p2 in zip(masks, masks2, poses, poses2):
        assert torch.equal(m1, m2) and torch.equal(p1, p2)
