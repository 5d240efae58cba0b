"""Oracle: DAGEM bridge block (TEST INFRASTRUCTURE).  Follows DAGEM_md.py:7-111.

Everything except the deformable conv is reference arithmetic; the deformable
conv is oracle.deform (torchvision semantics restated, PARITY UNPINNED).
"""
import torch
import torch.nn as nn

from .deform import DeformConv2d


def plain_conv_stand_in(module):
    """Replace `module.deform_conv.forward` by an ordinary 3x3 convolution of the same weight / bias plus 0.05 x the channel
    mean of the offsets (so that offset_conv still receives a gradient).  Applied IDENTICALLY to the reference's DAGEM (in
    tests/golden/make_golden.py), to this oracle and to the HIP model's DAGEM, it takes the one third-party operator
    (torchvision.ops.DeformConv2d, parity unpinned) out of the block: everything else in DAGEM_md.py:56-111 -- roll-edge
    products, the four Linear+BatchNorm1d MLPs, the final 1x1 conv + BatchNorm2d -- is then pinned by the dagem_plain_*
    fixtures."""
    import types
    import torch.nn.functional as F

    def forward(self, x, offset):
        return F.conv2d(x, self.weight, self.bias, padding=1) + 0.05 * offset.mean(dim=1, keepdim=True)

    module.deform_conv.forward = types.MethodType(forward, module.deform_conv)
    return module


def _mlp(i, o):
    return nn.Sequential(nn.Linear(i, o), nn.BatchNorm1d(o), nn.ReLU(inplace=True))


class DAGEM(nn.Module):
    def __init__(self, sync_bn=False, input_channels=256):
        super().__init__()
        c = self.input_channels = input_channels
        self.edge_aggregation_func = _mlp(4, 1)
        self.vertex_update_func = _mlp(2 * c, c // 2)
        self.edge_update_func = _mlp(2 * c, c // 2)
        self.update_edge_reduce_func = _mlp(4, 1)
        self.offset_conv = nn.Conv2d(c, 18, 3, padding=1)
        self.deform_conv = DeformConv2d(c, c, 3, padding=1)
        self.final_aggregation_layer = nn.Sequential(
            nn.Conv2d(c + c // 2, c, 1, bias=False), nn.BatchNorm2d(c), nn.ReLU(inplace=True))

    def forward(self, x):
        b, c, h, w = x.shape
        # 4-neighbour (wrap-around) products  (DAGEM_md.py:62-69)
        edge = torch.stack((torch.roll(x, 1, 2), torch.roll(x, -1, 2),
                            torch.roll(x, 1, 3), torch.roll(x, -1, 3)), dim=-1) * x.unsqueeze(-1)
        agg = self.edge_aggregation_func(edge.reshape(-1, 4)).reshape(b, c, h, w)
        vert = self.vertex_update_func(
            torch.cat((x, agg), 1).permute(0, 2, 3, 1).reshape(-1, 2 * c)
        ).reshape(b, h, w, c // 2).permute(0, 3, 1, 2)
        ef = torch.cat((x.unsqueeze(-1).expand(-1, -1, -1, -1, 4), edge), 1).permute(0, 2, 3, 4, 1).reshape(-1, 2 * c)
        ue = self.edge_update_func(ef).reshape(b, h, w, 4, c // 2).permute(0, 4, 1, 2, 3).reshape(-1, 4)
        ue = self.update_edge_reduce_func(ue).reshape(b, c // 2, h, w)
        deformed = self.deform_conv(x, self.offset_conv(x)) + x
        return self.final_aggregation_layer(torch.cat((deformed, vert * ue), 1))
