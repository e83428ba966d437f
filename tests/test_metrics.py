"""CPU: the panoptic-quality restatement (cellvit_amd/metrics.py) against outputs of the imported reference
(tests/golden/pq_cases.npz, tools/make_golden_pq.py) and hand-computed cases."""
import os

import numpy as np
import pytest

from cellvit_amd.metrics import binary_pq_batch, pairwise_iou, panoptic_quality, remap_label

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pq_cases.npz"))


@pytest.mark.parametrize("i", range(int(G["n"])))
def test_pq_matches_reference(i):
    k = f"c{i}"
    by_size, thr = bool(G[k + "_bysize"]), float(G[k + "_thr"])
    t, p = remap_label(G[k + "_true"], by_size), remap_label(G[k + "_pred"], by_size)
    assert np.array_equal(t, G[k + "_true_remap"]) and np.array_equal(p, G[k + "_pred_remap"])
    stats, pairs = panoptic_quality(t, p, thr)
    assert np.allclose(stats, G[k + "_stats"], rtol=0, atol=1e-12)
    for got, nm in zip(pairs, ("pt", "pp", "ut", "up")):
        assert sorted(got) == sorted(G[k + "_" + nm].tolist()), nm
    # pairs are aligned element-wise as well
    assert list(zip(pairs[0], pairs[1])) == list(zip(G[k + "_pt"].tolist(), G[k + "_pp"].tolist()))


def test_pq_hand_cases():
    a = np.zeros((8, 8), np.int32); a[1:5, 1:5] = 1; a[5:8, 5:8] = 2
    b = np.zeros((8, 8), np.int32); b[1:5, 1:5] = 1           # instance 2 missed
    (dq, sq, pq), (pt, pp, ut, up) = panoptic_quality(a, b)
    assert pt == [1] and pp == [1] and ut == [2] and up == []
    assert abs(dq - 1 / (1 + 0.5 + 1e-6)) < 1e-12 and abs(sq - 1 / (1 + 1e-6)) < 1e-12
    iou, ti, pi = pairwise_iou(a, np.roll(b, 1, axis=1))
    assert iou.shape == (2, 1) and abs(iou[0, 0] - 12 / 20) < 1e-12
    assert panoptic_quality(a, np.zeros_like(a))[0] == [0.0, 0.0, 0.0]
    assert binary_pq_batch([a, a], [a, b]) > 0.8
    with pytest.raises(AssertionError):
        panoptic_quality(a, b, -0.1)


def test_generate_instance_nuclei_map_matches_the_imported_reference():
    """P9 (cellvit.py:385-414): fixtures produced by the reference method itself (tools/make_golden_instmap.py)."""
    import os
    import torch
    from cellvit_amd.model import CellViT256
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "instmap_cases.npz"))
    B = int(g["n"])
    type_preds = [{int(i): {"type": int(t)} for i, t in zip(g[f"ids{b}"], g[f"types{b}"])} for b in range(B)]
    m = CellViT256(None, 6, 19)
    out = m.generate_instance_nuclei_map(torch.from_numpy(g["inst"]).float(), type_preds)
    assert out.dtype == torch.float32 and tuple(out.shape) == g["out"].shape
    assert np.array_equal(out.numpy(), g["out"])
    painted = sum(len(np.unique(g["out"][b])) - 1 for b in range(B))
    assert painted > 60                                              # and some instances deliberately absent from the dicts
    assert any(len(np.setdiff1d(np.unique(g["inst"][b])[1:], g[f"ids{b}"])) for b in range(B))
