"""SharedMLP / Conv2d / BatchNorm2d containers with the reference's state-dict layout
(`layer{i}.conv.weight`, `layer{i}.bn.bn.{weight,bias,running_mean,running_var,num_batches_tracked}`
-- the doubled `bn.bn` comes from the reference's _BNBase wrapper, lib/pytorch_utils.py:104-123).
Only the configuration RaTrack instantiates is provided: 1x1 Conv2d without bias, BatchNorm2d,
ReLU, post-activation (lib/pytorch_utils.py:5-32,163-197)."""
import torch.nn as nn


class BatchNorm2d(nn.Sequential):
    def __init__(self, channels):
        super().__init__()
        self.add_module("bn", nn.BatchNorm2d(channels))
        nn.init.constant_(self.bn.weight, 1.0)
        nn.init.constant_(self.bn.bias, 0.0)


class Conv2d(nn.Sequential):
    def __init__(self, in_size, out_size, bn=False, activation=True):
        super().__init__()
        conv = nn.Conv2d(in_size, out_size, kernel_size=(1, 1), bias=not bn)
        nn.init.kaiming_normal_(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0)
        self.add_module("conv", conv)
        if bn:
            self.add_module("bn", BatchNorm2d(out_size))
        if activation:
            self.add_module("activation", nn.ReLU(inplace=True))


class SharedMLP(nn.Sequential):
    def __init__(self, channels, bn=False):
        super().__init__()
        for i in range(len(channels) - 1):
            self.add_module("layer%d" % i, Conv2d(channels[i], channels[i + 1], bn=bn))
