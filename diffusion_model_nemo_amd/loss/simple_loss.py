"""L_simple of DDPM training. Mirror of diffusion_model_nemo/loss/simple_loss.py:10-45."""
import torch.nn.functional as F
from torch.nn.modules.loss import _Loss

_LOSSES = {"l1": F.l1_loss, "l2": F.mse_loss, "huber": F.smooth_l1_loss}


class DiffusionLoss(_Loss):
    def __init__(self, loss_type: str, reduction='mean'):
        self._reduction = reduction
        super().__init__(reduction='none' if reduction == 'batch_mean' else reduction)
        assert loss_type in _LOSSES, f"Loss type {loss_type} is not implemented !"
        self.loss_type = loss_type
        self.loss_fn = _LOSSES[loss_type]

    def forward(self, input, target):
        loss = self.loss_fn(input, target, reduction=self.reduction)
        if self._reduction == 'batch_mean':
            loss = loss.view(input.size(0), -1).sum(-1).mean()
        return loss
