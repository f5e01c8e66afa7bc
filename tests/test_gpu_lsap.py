"""GPU: the on-device Hungarian solver (one wavefront per problem) is bit-identical to scipy.optimize.linear_sum_assignment
(the reference's solver, matcher.py:124,126): golden (cost -> indices) pairs incl. ties, random / heavily tied
matrices checked against scipy on the GPU host, and the 4x-tiled many-to-one variant (matcher.py:125-127)."""
import numpy as np
import pytest
import torch

from helpers import load

pytestmark = pytest.mark.gpu


def solve(mats, tile=1):
    """mats: list of (Q, n) float32 arrays -> list of (rows, cols) numpy int64"""
    from gvl_amd.matcher import lsap_batch_device
    dev = torch.device("cuda:0")
    flat, desc, off, out = [], [], 0, 0
    for c in mats:
        Q, n = c.shape
        desc.append([off, n, Q, n, tile, out, 0, 0])
        flat.append(np.ascontiguousarray(c, np.float32).reshape(-1))
        off += c.size
        out += min(Q, n * tile)
    C = torch.from_numpy(np.concatenate(flat)).to(dev)
    P = torch.tensor(desc, dtype=torch.int64, device=dev)
    max_r = max(min(c.shape[0], c.shape[1] * tile) for c in mats)
    max_c = max(max(c.shape[0], c.shape[1] * tile) for c in mats)
    rows, cols, status = lsap_batch_device(C, P, out, max_r, max_c)
    assert int(status) == 0
    rows, cols = rows.cpu().numpy(), cols.cpu().numpy()
    res, o = [], 0
    for c in mats:
        k = min(c.shape[0], c.shape[1] * tile)
        res.append((rows[o:o + k], cols[o:o + k]))
        o += k
    return res


def test_device_lsap_matches_scipy_goldens():
    f = load("lsap_cases")
    names = sorted({k.split(".")[0] for k in f if "." in k})
    res = solve([f[f"{n}.C"] for n in names])
    for n, (r, c) in zip(names, res):
        assert np.array_equal(r, f[f"{n}.rows"]), n
        assert np.array_equal(c, f[f"{n}.cols"]), n


@pytest.mark.parametrize("tile", [1, 4])
def test_device_lsap_matches_scipy_random_and_ties(tile):
    from scipy.optimize import linear_sum_assignment
    rs = np.random.RandomState(321 + tile)
    mats = []
    for trial in range(200):
        Q, n = rs.randint(1, 120), rs.randint(1, 31)
        kind = trial % 4
        if kind == 0:
            C = rs.rand(Q, n)
        elif kind == 1:
            C = np.round(rs.rand(Q, n) * 3)                       # heavy ties
        elif kind == 2:
            C = rs.randn(Q, n) * 5
        else:
            C = np.round(rs.rand(Q, n) * 2) + (rs.rand(Q, n) < 0.3) * 1e-7      # near ties
        mats.append(C.astype(np.float32))
    mats.append(rs.rand(300, 30).astype(np.float32))              # the largest GVL case: Q=300, 30 GT, tiled -> 300 x 120
    res = solve(mats, tile)
    for C, (r, c) in zip(mats, res):
        er, ec = linear_sum_assignment(np.tile(C, (1, tile)))
        assert np.array_equal(r, er), C.shape
        assert np.array_equal(c, ec % C.shape[1]), C.shape


def test_device_lsap_flags_invalid_costs():
    from gvl_amd.matcher import lsap_batch_device
    dev = torch.device("cuda:0")
    C = torch.tensor([[1.0, float("nan")], [0.0, 1.0]], device=dev)
    P = torch.tensor([[0, 2, 2, 2, 1, 0, 0, 0]], dtype=torch.int64, device=dev)
    _, _, status = lsap_batch_device(C.reshape(-1), P, 2, 2, 2)
    assert int(status) == 1


def test_more_events_than_the_on_chip_solver_holds_falls_back_to_the_host_solver():
    """Q = 300 with > 64 events in one video exceeds the device solver's min(Q, 4 n) <= 256 limit; the criterion
    must then take the host solver (gvl_hungarian_batch_f32) instead of raising -- the reference handles any size."""
    from scipy.optimize import linear_sum_assignment
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    dev = torch.device("cuda:0")
    opt = make_opt(num_queries=300, device="cuda")
    _, criterion, _, _ = build(opt)
    g = torch.Generator().manual_seed(9)
    B, Q, ns = 2, 300, [70, 3]
    out = {"pred_logits": torch.randn(B, Q, 1, generator=g).to(dev), "pred_count": torch.randn(B, 11, generator=g).to(dev),
           "pred_boxes": (torch.rand(B, Q, 2, generator=g) * 0.5 + 0.2).to(dev)}
    targets = [{"boxes": (torch.rand(n, 2, generator=g) * 0.4 + 0.2).to(dev), "labels": torch.zeros(n, dtype=torch.long, device=dev)}
               for n in ns]
    losses, idx = criterion(dict(out), targets)
    C = criterion.matcher.cost_matrix(out, targets).float().cpu()
    for i, (c, (rows, cols)) in enumerate(zip(C.split(ns, -1), idx[0])):
        r, k = linear_sum_assignment(c[i].numpy())
        assert torch.equal(rows.cpu(), torch.as_tensor(r)) and torch.equal(cols.cpu(), torch.as_tensor(k))
    assert all(torch.isfinite(v) for k, v in losses.items() if "self_iou" not in k)
