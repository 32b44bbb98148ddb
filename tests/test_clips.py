"""a20 (clip aggregation: mean / max / LSE, LSE training loss, inference scores) against the oracle's restatement of
src/tasks/run_video_retrieval.py:402-418, 664-690 -- forward values and gradients, on the host emulator and on the GPU."""
import pytest
import torch

from clipbert_amd import clips
from oracle import clipbert_oracle as O


def _logits(n_clips, b, c, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(b, c, generator=g) * 3.0 for _ in range(n_clips)]


@pytest.mark.parametrize("method", ["mean", "max"])
@pytest.mark.parametrize("n_clips,b,c", [(1, 3, 2), (4, 5, 2), (2, 7, 5), (16, 3, 1)])
def test_mean_max_match_reference_and_autograd(hw, method, n_clips, b, c):
    ref_in = [t.clone().requires_grad_(True) for t in _logits(n_clips, b, c, 1)]
    ours_in = [hw(t.clone()).requires_grad_(True) for t in _logits(n_clips, b, c, 1)]
    ref = O.aggregate_clip_logits(ref_in, method)
    out = clips.aggregate_clip_logits(ours_in, method)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-6)
    w = torch.randn(b, c, generator=torch.Generator().manual_seed(2))
    (ref * w).sum().backward()
    (out * hw(w)).sum().backward()
    for a, r in zip(ours_in, ref_in):
        torch.testing.assert_close(a.grad.cpu(), r.grad, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n_clips,b,c", [(1, 4, 2), (4, 6, 2), (3, 5, 5)])
def test_lse_loss_and_inference(hw, n_clips, b, c):
    ref_in = [t.clone().requires_grad_(True) for t in _logits(n_clips, b, c, 3)]
    ours_in = [hw(t.clone()).requires_grad_(True) for t in _logits(n_clips, b, c, 3)]
    labels = torch.randint(0, c, (b,), generator=torch.Generator().manual_seed(4))
    ref_bnc = O.aggregate_clip_logits(ref_in, "lse")
    out_bnc = clips.aggregate_clip_logits(ours_in, "lse")
    assert tuple(out_bnc.shape) == (b, n_clips, c)
    ref_loss = O.lse_train_loss(ref_bnc, labels)
    loss = clips.lse_train_loss(out_bnc, hw(labels))
    torch.testing.assert_close(loss.cpu(), ref_loss, rtol=1e-5, atol=1e-5)
    ref_loss.mean().backward()
    loss.mean().backward()
    for a, r in zip(ours_in, ref_in):
        torch.testing.assert_close(a.grad.cpu(), r.grad, rtol=1e-5, atol=1e-6)
    pooled = clips.lse_inference_logits(out_bnc.detach())
    torch.testing.assert_close(pooled.cpu(), torch.logsumexp(ref_bnc.detach(), dim=1), rtol=1e-5, atol=1e-5)
    if c == 2:
        assert clips.retrieval_scores(pooled) == [round(float(s), 4) for s in O.lse_inference_scores(ref_bnc.detach())]


def test_bad_pool_method_raises(hw):
    with pytest.raises(ValueError, match="Invalid value for pool_method"):
        clips.aggregate_clip_logits([hw(torch.zeros(2, 2))], "median")
