"""
Loss terms of the reference's `empose/nn/loss.py` (reporting and the autograd training path; the in-loop reconstruction
residual and the hand-written training step's loss terms live in the HIP kernels: csrc/smpl*.hip, csrc/train.hip
`lgd_losses_kernel`).
"""
import torch


def mask_from_seq_lengths(seq_lengths, max_seq_len=None):
    """reference helpers/utils.py:105-123"""
    max_seq_len = int(seq_lengths.max()) if max_seq_len is None else max_seq_len
    t = torch.arange(max_seq_len, device=seq_lengths.device, dtype=seq_lengths.dtype)
    return t[None, :] < seq_lengths[:, None]


def reconstruction_loss(markers_gt, markers_hat, seq_lengths=None, marker_mask=None):
    """reference nn/loss.py:23-41 (used for reporting loss values only; the in-loop residual lives in the kernels)."""
    diff = markers_hat - markers_gt
    per = torch.sqrt((diff * diff).sum(dim=-1)).sum(dim=-1)
    if marker_mask is not None:
        per = per * marker_mask.logical_not().any(dim=-1).logical_not()
    if seq_lengths is not None:
        mask = mask_from_seq_lengths(seq_lengths, per.shape[1]).to(per.dtype)
        per = (per * mask).sum(-1) / seq_lengths.to(per.dtype)
    return per.mean()


def normal_mse(x_gt, x_hat, seq_lengths=None, marker_mask=None):
    """reference nn/loss.py:44-62: squared error summed over joints, padded mean over frames, mean over the batch."""
    diff = x_hat - x_gt
    per = (diff * diff).sum(dim=-1).sum(dim=-1)
    if marker_mask is not None:
        per = per * marker_mask.logical_not().any(dim=-1).logical_not()
    if seq_lengths is not None:
        mask = mask_from_seq_lengths(seq_lengths, per.shape[1]).to(per.dtype)
        per = (per * mask).sum(-1) / seq_lengths.to(per.dtype)
    return per.mean()


def padded_loss(gt, hat, loss_fn, seq_lengths):
    """reference nn/loss.py:13-20"""
    unreduced = loss_fn(gt, hat).mean(-1)
    mask = mask_from_seq_lengths(seq_lengths, unreduced.shape[1]).to(unreduced.dtype)
    return ((unreduced * mask).sum(-1) / seq_lengths.to(unreduced.dtype)).mean()
