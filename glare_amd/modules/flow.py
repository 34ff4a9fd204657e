"""Parameter containers of the flow layers, with the reference's names.
Mirrors code/models/modules/flow.py:13-70 (Conv2d, Conv2dZeros), FlowActNorms.py:10-112 (ActNorm2d),
Permutations.py:12-59 (InvertibleConv1x1).  They hold state only; the arithmetic of a flow step is
fused in FlowUpsamplerNet (csrc/flow.hip + csrc/conv_igemm.hip)."""
import numpy as np
import torch
import torch.nn as nn


class ActNorm2d(nn.Module):
    def __init__(self, num_features, scale=1.0):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(1, num_features, 1, 1))
        self.logs = nn.Parameter(torch.zeros(1, num_features, 1, 1))
        self.num_features, self.scale = num_features, float(scale)
        self.inited = False   # FlowActNorms.py:23: the first TRAINING forward initialises an all-zero ActNorm from its batch
                              # (FlowUpsamplerNet.initialize_actnorms_nhwc; eval / inference never does, :34-35)


class InvertibleConv1x1(nn.Module):
    def __init__(self, num_channels, LU_decomposed=False):
        super().__init__()
        assert not LU_decomposed
        w = np.linalg.qr(np.random.randn(num_channels, num_channels))[0].astype(np.float32)
        self.weight = nn.Parameter(torch.from_numpy(w))
        self.w_shape = [num_channels, num_channels]


class Conv2d(nn.Conv2d):
    """bias-free conv followed by ActNorm2d: y = (conv(x) + actnorm.bias) * exp(actnorm.logs)."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding="same", do_actnorm=True,
                 weight_std=0.05):
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        super().__init__(in_channels, out_channels, k, 1, (k - 1) // 2, bias=not do_actnorm)
        self.weight.data.normal_(mean=0.0, std=weight_std)
        assert do_actnorm
        self.actnorm = ActNorm2d(out_channels)
        self.do_actnorm = True

    def folded(self):
        """(weight, bias) of the equivalent plain conv."""
        s = torch.exp(self.actnorm.logs.detach().reshape(-1))
        return self.weight.detach() * s.view(-1, 1, 1, 1), self.actnorm.bias.detach().reshape(-1) * s


class Conv2dZeros(nn.Conv2d):
    """y = conv(x) * exp(3 * logs)."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding="same", logscale_factor=3):
        super().__init__(in_channels, out_channels, 3, 1, 1)
        self.logscale_factor = logscale_factor
        self.logs = nn.Parameter(torch.zeros(out_channels, 1, 1))
        self.weight.data.zero_()
        self.bias.data.zero_()

    def folded(self):
        s = torch.exp(self.logs.detach().reshape(-1) * self.logscale_factor)
        return self.weight.detach() * s.view(-1, 1, 1, 1), self.bias.detach() * s
