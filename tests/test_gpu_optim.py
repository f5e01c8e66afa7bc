"""gvl_amd.optim.ClipAdam (gvl_clip_adam_step_f32) against what it replaces: torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step()
(train.py:405-409) -- parameters, both moments, the step counters and the reported norm over several steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("max_norm,wd", [(0.1, 1e-4), (1e6, 0.0), (0.0, 1e-4)])
def test_clip_and_adam_equal_torch(max_norm, wd):
    from gvl_amd.optim import ClipAdam
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    shapes = [(512, 512), (2048, 512), (512,), (1,), (3, 7), (8518, 512), (33,), (2, 4097)]
    flat = torch.randn(sum(torch.Size(s).numel() for s in shapes) + 3, device=dev, generator=g)

    def make():
        ps, off = [], 1                                      # (offset 1: views that are NOT 16-byte aligned, as slices of a flat buffer)
        for i, s in enumerate(shapes):
            n = torch.Size(s).numel()
            src = flat[off:off + n] if i % 2 else flat[off:off + n].clone()
            ps.append(torch.nn.Parameter(src.view(s).clone() if i % 2 == 0 else src.view(s).detach().clone()))
            off += n
        return ps
    pa, pb = make(), make()
    oa = torch.optim.Adam(pa, lr=5e-3, weight_decay=wd, capturable=True, fused=True)
    ob = torch.optim.Adam(pb, lr=5e-3, weight_decay=wd, capturable=True, fused=True)
    ca = ClipAdam(oa, max_norm)
    used = []
    for step in range(5):
        grads = [torch.randn(p.shape, device=dev, generator=g) * (10.0 if step in (2, 3) else 0.01) for p in pa]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        if step == 2:
            # some parameters sit this step out (a batch without events leaves the captioner without gradients, train.py's loop
            # and TrainStep hide them): torch.optim.Adam counts steps PER PARAMETER, and so must the table
            for i in (1, 4, 5):
                pa[i].grad = pb[i].grad = None
        act_a, act_b = [p for p in pa if p.grad is not None], [q for q in pb if q.grad is not None]
        va = [p._version for p in act_a]
        used.append(ca.step(act_a))
        if not used[-1]:                                     # (first step: torch creates the state)
            torch.nn.utils.clip_grad_norm_(act_a, max_norm)
            oa.step()
        else:
            assert all(p._version > v for p, v in zip(act_a, va))         # raw-pointer update, but the version counters moved
        total = torch.nn.utils.clip_grad_norm_(act_b, max_norm)   # max_norm = 0 zeroes the gradients on both paths (train.py:407)
        ob.step()
        if used[-1]:
            assert abs(float(ca.last[0]) - float(total)) <= 1e-5 * float(total)
        for i, (p, q) in enumerate(zip(pa, pb)):
            if p.grad is None:
                assert q.grad is None and float(oa.state[p]["step"]) == float(ob.state[q]["step"]) == 2.0
                continue
            assert float((p.grad - q.grad).abs().max()) <= 1e-6 * max(1e-6, float(q.grad.abs().max()))      # clipped in place
            assert float((p.detach() - q.detach()).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max())), (step, tuple(p.shape))
            sa, sb = oa.state[p], ob.state[q]
            assert float(sa["step"]) == float(sb["step"]) == step + 1 - (step > 2 and i in (1, 4, 5))
            assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 1e-6 * max(1e-3, float(sb["exp_avg"].abs().max()))
            assert float((sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max()) <= 1e-6 * max(1e-6, float(sb["exp_avg_sq"].abs().max()))
    assert used == [False, True, True, True, True]
