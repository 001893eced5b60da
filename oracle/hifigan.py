"""HiFi-GAN generator, CPU restatement with weight-norm folded.
Reference: satools/satools/hifigan/archi.py:77-91 (forward_resnet), hifigan/nn.py:179-186
(ResBlock1.forward), get_padding nn.py:17."""
import torch
import torch.nn.functional as F

UP_RATES = [5, 4, 4, 2, 2]
UP_KERNELS = [11, 8, 8, 4, 4]
RB_KERNELS = [3, 7, 11]
RB_DIL = [1, 3, 5]


def folded(sd, prefix):
    """w = g * v / ||v|| over all dims but 0 (torch weight_norm, dim=0)"""
    if prefix + "weight" in sd:
        return sd[prefix + "weight"]
    v, g = sd[prefix + "weight_v"], sd[prefix + "weight_g"]
    return v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape))


def resblock(sd, prefix, x, k):
    for i, d in enumerate(RB_DIL):
        xt = F.leaky_relu(x, 0.1)
        xt = F.conv1d(xt, folded(sd, f"{prefix}convs1.{i}."), sd[f"{prefix}convs1.{i}.bias"], dilation=d,
                      padding=(k * d - d) // 2)
        xt = F.leaky_relu(xt, 0.1)
        xt = F.conv1d(xt, folded(sd, f"{prefix}convs2.{i}."), sd[f"{prefix}convs2.{i}.bias"], padding=(k - 1) // 2)
        x = xt + x
    return x


def generator(sd, x, hook=None):
    """sd: generator state dict (keys without 'hifigan.'), x [B, C_in, T] -> [B, 1, T*320 + 1]"""
    x = F.conv1d(x, folded(sd, "conv_pre."), sd["conv_pre.bias"], padding=3)
    if hook:
        hook("conv_pre", x)
    for i, (u, k) in enumerate(zip(UP_RATES, UP_KERNELS)):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, folded(sd, f"ups.{i}."), sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if hook:
            hook(f"ups.{i}", x)
        xs = torch.zeros_like(x)
        for j, rk in enumerate(RB_KERNELS):
            r = resblock(sd, f"resblocks.{3 * i + j}.", x, rk)
            if hook:
                hook(f"resblocks.{3 * i + j}", r)
            xs += r
        x = xs / len(RB_KERNELS)
        if hook:
            hook(f"mrf.{i}", x)
    x = F.leaky_relu(x)
    x = F.pad(x, (1, 0), mode="reflect")
    x = F.conv1d(x, folded(sd, "conv_post."), sd["conv_post.bias"], padding=3)
    return torch.tanh(x)
