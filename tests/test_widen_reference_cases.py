"""More of the reference's OWN test cases (late-sorting file: written after round 1's last device run), restated against ``lightning_pose_amd`` (SURVEY.md section 8c lists them).

Each test names the reference test it mirrors (file::class::test under /root/reference/tests).  They call the product's public API -
the same class / function names and keyword arguments a user of ``lightning_pose`` would use - so they run on the emulated kernels in
the CPU suite and on the MI355X through ``liblp_hip.so`` under ``-m gpu``.  Cases that need the reference's toy dataset fixtures
(``heatmap_data_module`` ...) are rebuilt from literal tensors with the same structure.
"""

import math

import numpy as np
import pytest
import torch

STAGE = "train"


@pytest.fixture
def dev(stack_backend):
    return stack_backend


def test_evaluate_heatmaps_at_location_cases(dev):
    """test_evaluate_heatmaps_at_location (tests/data/test_heatmaps.py:457-563): five 0.2 blobs around the location sum to exactly
    1 for 1 / 5 frames x 1 / 6 keypoints incl. locations at the border; delta and Gaussian maps at the right, adjacent, wrong spot"""
    from lightning_pose_amd.data.heatmaps import evaluate_heatmaps_at_location, generate_heatmaps

    height, width = 24, 12
    g = torch.Generator().manual_seed(4)
    for n_batch in (1, 5):
        for n_keypoints in (1, 6):
            heatmaps = torch.zeros(n_batch, n_keypoints, height, width)
            h_locs = torch.randint(0, height, (n_batch, n_keypoints), generator=g)
            w_locs = torch.randint(0, width, (n_batch, n_keypoints), generator=g)
            if n_batch == 5 and n_keypoints == 6:  # corners and edges
                h_locs[0, :4], w_locs[0, :4] = torch.tensor([0, 0, height - 1, height - 1]), torch.tensor([0, width - 1, 0, width - 1])
            locs = torch.stack([w_locs, h_locs], dim=2)
            for i in range(n_batch):
                for j in range(n_keypoints):
                    for dy, dx in ((1, 1), (-1, -1), (0, 0), (1, -1), (-1, 1)):
                        y = int(torch.clamp(locs[i, j, 1] + dy, 0, height - 1))
                        x = int(torch.clamp(locs[i, j, 0] + dx, 0, width - 1))
                        heatmaps[i, j, y, x] += 0.2
            vals = evaluate_heatmaps_at_location(heatmaps=heatmaps.to(dev), locs=locs.to(dev)).cpu()
            assert vals.shape == (n_batch, n_keypoints)
            assert torch.all(vals == 1.0)
    heatmaps = torch.zeros(1, 1, 32, 32)
    heatmaps[0, 0, 5, 5] = 1
    loc = lambda v: torch.full((1, 1, 2), float(v))  # noqa: E731
    conf = lambda hm, v: evaluate_heatmaps_at_location(hm.to(dev), loc(v).to(dev)).cpu()  # noqa: E731
    assert conf(heatmaps, 5).shape == (1, 1)
    assert torch.allclose(conf(heatmaps, 5)[0], torch.tensor(1.0))
    assert torch.allclose(conf(heatmaps, 6)[0], torch.tensor(1.0))
    assert torch.allclose(conf(heatmaps, 25)[0], torch.tensor(0.0))
    assert torch.allclose(conf(heatmaps, 5.9)[0], torch.tensor(1.0)) and torch.allclose(conf(heatmaps, 8.0)[0], torch.tensor(0.0))  # int64 truncation
    hm_g = generate_heatmaps(loc(5).to(dev), height=32, width=32, output_shape=(32, 32)).cpu()
    c0, c1, c2 = conf(hm_g, 5)[0], conf(hm_g, 6)[0], conf(hm_g, 25)[0]
    assert 0 < float(c0) <= 1.0 and float(c0) > float(c1)
    assert torch.allclose(c2, torch.tensor(0.0))


def test_temporal_heatmap_loss_cases(dev):
    """TestTemporalHeatmapLoss (tests/losses/test_losses.py:411-503): invalid name, zero for constant maps (mse exactly, kl to 1e-5),
    positive for varying maps, compute_loss shape, low-confidence masking, epsilon rectification; plus the composition identity"""
    from lightning_pose_amd.losses import TemporalHeatmapLoss
    from lightning_pose_amd.losses.factory import get_loss_classes

    assert get_loss_classes()["temporal_heatmap_mse"] is TemporalHeatmapLoss and get_loss_classes()["temporal_heatmap_kl"] is TemporalHeatmapLoss
    with pytest.raises(ValueError):
        TemporalHeatmapLoss(loss_name="bad_name")
    mse, kl = TemporalHeatmapLoss(loss_name="temporal_heatmap_mse"), TemporalHeatmapLoss(loss_name="temporal_heatmap_kl")
    g = torch.Generator().manual_seed(8)
    S, K, h, w = 4, 3, 16, 16
    ones = torch.ones(S, K).to(dev)
    frame = torch.rand(1, K, h, w, generator=g)
    loss, logs = mse(heatmaps_pred=frame.expand(S, -1, -1, -1).clone().to(dev), confidences=ones, stage=STAGE)
    assert loss.shape == torch.Size([]) and loss.item() == 0.0
    assert logs[0]["name"] == f"{STAGE}_temporal_heatmap_mse_loss" and logs[1]["name"] == "temporal_heatmap_mse_weight"
    loss, _ = mse(heatmaps_pred=torch.rand(S, K, h, w, generator=g).to(dev), confidences=ones, stage=STAGE)
    assert loss.item() > 0.0
    sm = lambda x: torch.softmax(x.reshape(*x.shape[:2], -1), -1).reshape(x.shape)  # noqa: E731  (kornia spatial_softmax2d)
    frame = sm(torch.randn(1, K, h, w, generator=g))
    loss, logs = kl(heatmaps_pred=frame.expand(S, -1, -1, -1).clone().to(dev), confidences=ones, stage=STAGE)
    assert loss.shape == torch.Size([]) and torch.isclose(loss.cpu(), torch.tensor(0.0), atol=1e-5)
    assert logs[0]["name"] == f"{STAGE}_temporal_heatmap_kl_loss"
    varying = sm(torch.randn(S, K, h, w, generator=g))
    loss, _ = kl(heatmaps_pred=varying.to(dev), confidences=ones, stage=STAGE)
    assert loss.item() > 0.0
    assert mse.compute_loss(torch.rand(5, K, h, w, generator=g).to(dev)).shape == (4, K)
    assert kl.compute_loss(sm(torch.randn(5, K, h, w, generator=g)).to(dev)).shape == (4, K)
    # low-confidence masking
    mse.prob_threshold = torch.tensor(0.5)
    pred = torch.rand(3, 2, 8, 8, generator=g)
    conf = torch.zeros(3, 2)
    conf[:, 1] = 1.0
    diffs = mse.compute_loss(pred.to(dev))
    want = ((pred[1:] - pred[:-1]) ** 2).mean((-1, -2))
    torch.testing.assert_close(diffs.cpu(), want, rtol=1e-5, atol=1e-8)
    clean = mse.remove_nans(confidences=conf.to(dev), loss=diffs.clone())
    assert torch.all(clean[:, 0] == 0.0) and torch.all(clean[:, 1] > 0.0)
    # epsilon rectification
    mse.epsilon = torch.tensor(1.0)
    rect = mse.rectify_epsilon(torch.tensor([[0.5, 2.0], [1.5, 0.3]]))
    assert rect[0, 0] == 0.0 and rect[0, 1] > 0.0 and rect[1, 0] > 0.0 and rect[1, 1] == 0.0
    # __call__ == reduce(rectify(remove_nans(compute_loss))) with a per-keypoint epsilon
    loss_obj = TemporalHeatmapLoss(loss_name="temporal_heatmap_kl", epsilon=[0.0, 0.05, 10.0], prob_threshold=0.3)
    conf = torch.rand(S, K, generator=g)
    got, _ = loss_obj(heatmaps_pred=varying.to(dev), confidences=conf.to(dev), stage=None)
    parts = loss_obj.rectify_epsilon(loss_obj.remove_nans(conf.to(dev), loss_obj.compute_loss(varying.to(dev))))
    assert float(got) == pytest.approx(float(loss_obj.reduce_loss(parts)), rel=1e-5)


def test_pca_loss_keeps_any_number_of_components(dev):
    """components_to_keep close to 1 on noisy labels keeps most of the 2K dimensions (the reference has no cap): 28 kept of 34, and
    40 keypoints (80 dimensions) single-view"""
    from lightning_pose_amd import ops

    g = torch.Generator().manual_seed(0)
    for K, ncomp in ((17, 28), (40, 33)):
        D = 2 * K
        q, _ = torch.linalg.qr(torch.randn(D, D, generator=g))
        kept, mean = q[:ncomp].contiguous(), torch.randn(D, generator=g)
        kp = (torch.randn(6, D, generator=g) * 5).requires_grad_(True)
        idx = torch.arange(K, dtype=torch.int32).reshape(1, K)  # (rows = 1, points = K): single-view
        x = kp.reshape(6, D) - mean
        r = x - (x @ kept.T) @ kept
        want = torch.relu(r.reshape(6, K, 2).norm(dim=-1) - 0.3).mean()
        want.backward()
        kd = kp.detach().to(dev).requires_grad_(True)
        got = ops.pca_loss(kd, idx.to(dev), mean.to(dev), kept.to(dev), 0.3)
        got.backward()
        assert float(got.detach()) == pytest.approx(float(want.detach()), rel=1e-4)
        torch.testing.assert_close(kd.grad.cpu(), kp.grad, atol=1e-5, rtol=1e-3)
