"""L_simple of DDPM training: a pointwise distance between the network output and its target with the reference's
three choices and reductions (diffusion_model_nemo/loss/simple_loss.py:10-45).

`loss_type`: 'l1' | 'l2' | 'huber' (smooth L1 with beta 1). `reduction`: torch's 'mean' / 'sum' / 'none', or
'batch_mean' = sum over everything but the batch axis, then the mean over the batch. The native training step
(`training.UnetTrainer`) evaluates the same three distances with the 'mean' reduction inside its loss kernel; this
module is the host-side definition for tensors that are already on hand (tests, step-by-step use).
"""
import torch
import torch.nn.functional as F

POINTWISE = {
    "l1": F.l1_loss,
    "l2": F.mse_loss,
    "huber": F.smooth_l1_loss,
}


class DiffusionLoss(torch.nn.Module):
    def __init__(self, loss_type: str, reduction: str = 'mean'):
        super().__init__()
        if loss_type not in POINTWISE:   # same exception type and text as simple_loss.py:17
            raise AssertionError(f"Loss type {loss_type} is not implemented !")
        self.loss_type = loss_type
        self._reduction = reduction
        # what torch's functional is asked for: per-element values when the batch-wise reduction follows
        self.reduction = 'none' if reduction == 'batch_mean' else reduction

    @property
    def loss_fn(self):
        return POINTWISE[self.loss_type]

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        value = POINTWISE[self.loss_type](input, target, reduction=self.reduction)
        if self._reduction != 'batch_mean':
            return value
        return value.flatten(1).sum(dim=1).mean()
