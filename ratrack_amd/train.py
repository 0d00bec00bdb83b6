"""Training step of the backbone (counterpart of the inner loop of main_utils.epoch, main_utils.py:122-156,
248-251, minus the host-side label/GT machinery): forward in train mode (batch-statistic BatchNorm, the
module path with HIP ops + PyTorch-ROCm dense layers and autograd), multi-task loss, backward, gradient
all-reduce across ranks, optimizer step."""
import torch

from . import loss as L
from .ddp import FlatGradAllReducer


def make_optimizer(model, lr=1e-3):
    """Adam(lr=1e-3, weight_decay=1e-10) + StepLR(step=1, gamma=0.97) as main.py:61-62."""
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=1e-10)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.97)
    return opt, sched


class Trainer:
    def __init__(self, model, lr=1e-3, process_group=None):
        self.model = model
        self.opt, self.sched = make_optimizer(model, lr)
        self.reducer = FlatGradAllReducer(model, process_group)

    def step(self, pc1, pc2, feature1, feature2, gt_warp, gt_cls, h=None, pretrain=False):
        """One optimisation step on this rank's shard.  Returns the loss items (python floats are NOT taken here:
        no device->host sync inside the step)."""
        self.model.train()
        flow, h_out, cls, *_ = self.model.backbone(pc1, pc2, feature1, feature2, h)
        total, items = L.backbone_loss(pc1 + flow, cls, gt_warp, gt_cls, pretrain=pretrain)
        self.opt.zero_grad(set_to_none=True)
        total.backward()
        self.reducer.reduce()
        self.opt.step()
        return items, h_out.detach()
