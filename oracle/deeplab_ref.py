"""Functional CPU restatement of the reference generator (DeepLabV3+ on MobileNetV2 or ResNet-101).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The network is evaluated directly
on a flat ``state_dict`` (reference key names) with ``torch.nn.functional`` ops,
so it needs no module tree and cannot share code with the product.

Reference anchors
  networks/deeplabv3.py:32-41      DeepLab.forward (7-tuple order)
  networks/backbone/mobilenet.py   :8-13 stem, :16-22 fixed_padding, :25-67
                                   InvertedResidual, :77-86 block table,
                                   :93-101 output-stride/dilation rule, :116-122 split
  networks/backbone/resnet.py      :23-43 Bottleneck.forward, :47-70 strides/dilations/MG unit,
                                   :113-124 ResNet.forward (low-level = layer1 output)
  networks/aspp.py:65-78           ASPP.forward
  networks/decoder.py:45-56        Decoder.forward

Dropout: the reference draws masks with ``nn.Dropout`` from the global CPU
generator.  ``masks=None`` draws them the same way (same shapes, same order, so
the same stream under the same seed); ``masks={name: keep}`` injects keep-masks
(1 = keep) so the HIP path can be compared on identical masks.  ``record`` (a
dict) receives the keep-masks actually used.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# (expand_ratio t, out_channels c, repeats n, stride s) -- mobilenet.py:77-86
_MBV2_TABLE = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
               (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))

DROPOUT_SITES = (  # name, p  (aspp.py:62; decoder.py:36,40,31)
    ("aspp.dropout", 0.5),
    ("decoder.last_conv_boundary.3", 0.5),
    ("decoder.last_conv_boundary.7", 0.1),
    ("decoder.last_conv.2", 0.1),
)


def mbv2_blocks(output_stride=16):
    """Per-block (inp, oup, stride, dilation, expand) following mobilenet.py:88-111."""
    blocks = []
    inp, cur, rate = 32, 2, 1
    for t, c, n, s in _MBV2_TABLE:
        if cur == output_stride:
            stride, dil = 1, rate
            rate *= s
        else:
            stride, dil = s, 1
            cur *= s
        for i in range(n):
            blocks.append((inp, c, stride if i == 0 else 1, dil, t))
            inp = c
    return blocks


class _Ctx:
    def __init__(self, sd, training, masks, record, bn_training=None):
        self.sd, self.training, self.masks, self.record = sd, training, masks, record
        # DeepLab.freeze_bn() (deeplabv3.py:43-50) puts the BatchNorm modules in eval mode while the model (dropout) keeps training
        self.bn_training = training if bn_training is None else bn_training

    def transnorm(self, x, prefix):
        """TransNorm (``--use_TN``; networks/sync_batchnorm/batchnorm.py:436-520).  Training: the first
        N//2 images of WHATEVER batch arrives are the "source" half, the rest the "target" half; each half is
        batch-normalised with its own statistics (and updates its own running buffers), then both are scaled
        per channel by 1 + alpha, alpha = C * p / sum(p), p = 1 / (1 + |mu_s/sqrt(var_s+eps) - mu_t/sqrt(var_t+eps)|)
        with the UNBIASED variances of the two halves; alpha carries no gradient.  Eval: target running
        statistics normalise, alpha comes from the two pairs of running statistics."""
        sd = self.sd
        w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
        rms, rvs = sd[prefix + ".running_mean_source"], sd[prefix + ".running_var_source"]
        rmt, rvt = sd[prefix + ".running_mean_target"], sd[prefix + ".running_var_target"]
        C = x.shape[1]
        if self.bn_training:
            sd[prefix + ".num_batches_tracked"] += 1
            n0 = x.shape[0] // 2
            halves = (x[:n0], x[n0:])
            z = torch.cat([F.batch_norm(h, rm, rv, w, b, True, BN_MOMENTUM, BN_EPS)
                           for h, rm, rv in zip(halves, (rms, rmt), (rvs, rvt))], 0)
            # [pixels, C] rows reduced along dim 0: the reference's reduction layout, so fp32 sums round identically
            flat = [h.permute(0, 2, 3, 1).reshape(-1, C) for h in halves]
            ratio = [f.mean(0) / torch.sqrt(f.var(0) + BN_EPS) for f in flat]
        else:
            z = F.batch_norm(x, rmt, rvt, w, b, False, BN_MOMENTUM, BN_EPS)
            ratio = [rms / torch.sqrt(rvs + BN_EPS), rmt / torch.sqrt(rvt + BN_EPS)]
        prob = 1.0 / (1.0 + (ratio[0] - ratio[1]).abs())
        total = prob[0]
        for v in prob[1:]:               # the reference adds the C terms one by one (Python ``sum``): same fp32 rounding order
            total = total + v
        alpha = (C * prob / total).detach()
        return z * (1.0 + alpha.view(1, C, 1, 1))

    def bn(self, x, prefix):
        sd = self.sd
        if (prefix + ".running_mean_source") in sd:
            return self.transnorm(x, prefix)
        if self.bn_training and (prefix + ".num_batches_tracked") in sd:
            sd[prefix + ".num_batches_tracked"] += 1
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                            sd[prefix + ".weight"], sd[prefix + ".bias"],
                            self.bn_training, BN_MOMENTUM, BN_EPS)

    def dropout(self, x, name, p):
        if not self.training:
            return x
        if self.masks is not None:
            keep = self.masks[name].to(x.dtype)
            noise = keep.div(1.0 - p)
        else:
            noise = F.dropout(torch.ones_like(x), p, True)   # same draw as nn.Dropout(x)
            keep = noise != 0
        if self.record is not None:
            self.record[name] = keep.to(torch.uint8)
        return x * noise


def _inverted_residual(c, x, pre, inp, oup, stride, dil, t):
    # mobilenet.py:61-67 : pad the BLOCK INPUT, run the whole conv stack on it (quirk Q1)
    sd = c.sd
    xp = F.pad(x, (dil, dil, dil, dil))
    hid = round(inp * t)
    if t == 1:
        h = F.conv2d(xp, sd[pre + ".conv.0.weight"], None, stride, 0, dil, hid)
        h = F.hardtanh(c.bn(h, pre + ".conv.1"), 0.0, 6.0)
        h = c.bn(F.conv2d(h, sd[pre + ".conv.3.weight"]), pre + ".conv.4")
    else:
        h = F.conv2d(xp, sd[pre + ".conv.0.weight"])
        h = F.hardtanh(c.bn(h, pre + ".conv.1"), 0.0, 6.0)
        h = F.conv2d(h, sd[pre + ".conv.3.weight"], None, stride, 0, dil, hid)
        h = F.hardtanh(c.bn(h, pre + ".conv.4"), 0.0, 6.0)
        h = c.bn(F.conv2d(h, sd[pre + ".conv.6.weight"]), pre + ".conv.7")
    if stride == 1 and inp == oup:
        h = x + h
    return h


def resnet_blocks(output_stride=16, layers=(3, 4, 23)):
    """[(prefix, inplanes, planes, stride, dilation, has_downsample)] of ResNet-101 as the reference
    builds it (resnet.py:47-111): three plain layers, then the multi-grid unit [1,2,4] x dilation."""
    strides, dils = ((1, 2, 2, 1), (1, 1, 1, 2)) if output_stride == 16 else ((1, 2, 1, 1), (1, 1, 2, 4))
    out, inp = [], 64
    for li, (planes, n) in enumerate(zip((64, 128, 256), layers)):
        for b in range(n):
            s = strides[li] if b == 0 else 1
            out.append(("backbone.layer%d.%d" % (li + 1, b), inp, planes, s, dils[li],
                        b == 0 and (s != 1 or inp != planes * 4)))
            inp = planes * 4
    for b, mg in enumerate((1, 2, 4)):
        s = strides[3] if b == 0 else 1
        out.append(("backbone.layer4.%d" % b, inp, 512, s, mg * dils[3], b == 0 and (s != 1 or inp != 2048)))
        inp = 2048
    return out


def _bottleneck(c, x, pre, stride, dil, has_ds):
    # resnet.py:23-43
    sd = c.sd
    h = F.relu(c.bn(F.conv2d(x, sd[pre + ".conv1.weight"]), pre + ".bn1"))
    h = F.relu(c.bn(F.conv2d(h, sd[pre + ".conv2.weight"], None, stride, dil, dil), pre + ".bn2"))
    h = c.bn(F.conv2d(h, sd[pre + ".conv3.weight"]), pre + ".bn3")
    res = x
    if has_ds:
        res = c.bn(F.conv2d(x, sd[pre + ".downsample.0.weight"], None, stride), pre + ".downsample.1")
    return F.relu(h + res)


def _resnet_backbone(c, x, output_stride):
    # resnet.py:113-124
    sd = c.sd
    h = F.relu(c.bn(F.conv2d(x, sd["backbone.conv1.weight"], None, 2, 3), "backbone.bn1"))
    h = F.max_pool2d(h, 3, 2, 1)
    low = None
    for pre, inp, planes, stride, dil, has_ds in resnet_blocks(output_stride):
        if pre == "backbone.layer2.0":
            low = h
        h = _bottleneck(c, h, pre, stride, dil, has_ds)
    return h, low


def _mobilenet_backbone(c, x, output_stride):
    # mobilenet.py:8-13, 116-122
    sd = c.sd
    h = F.conv2d(x, sd["backbone.features.0.0.weight"], None, 2, 1)
    h = F.hardtanh(c.bn(h, "backbone.features.0.1"), 0.0, 6.0)
    low = None
    for i, (inp, oup, stride, dil, t) in enumerate(mbv2_blocks(output_stride), start=1):
        h = _inverted_residual(c, h, "backbone.features.%d" % i, inp, oup, stride, dil, t)
        if i == 3:
            low = h
    return h, low


def deeplab_forward(sd, x, training=True, masks=None, record=None, output_stride=16, bn_training=None):
    """Returns (x1, x2, feature, x_bu_feature, x_feature, x1_before, x2_before).

    ``sd`` maps reference state-dict keys to tensors; BN running stats in it are
    updated in place when ``training`` (deeplabv3.py:32-41).  The backbone is told
    from the keys (``backbone.conv1.weight`` exists only in the ResNet).
    """
    c = _Ctx(sd, training, masks, record, bn_training)
    if "backbone.conv1.weight" in sd:
        h, low = _resnet_backbone(c, x, output_stride)
    else:
        h, low = _mobilenet_backbone(c, x, output_stride)
    # --- ASPP (aspp.py:65-78)
    dils = (1, 6, 12, 18) if output_stride == 16 else (1, 12, 24, 36)
    br = []
    for j, d in enumerate(dils, start=1):
        w = sd["aspp.aspp%d.atrous_conv.weight" % j]
        y = F.conv2d(h, w, None, 1, 0 if j == 1 else d, d)
        br.append(F.relu(c.bn(y, "aspp.aspp%d.bn" % j)))
    g = F.adaptive_avg_pool2d(h, 1)
    g = F.relu(c.bn(F.conv2d(g, sd["aspp.global_avg_pool.1.weight"]), "aspp.global_avg_pool.2"))
    g = F.interpolate(g, size=h.shape[2:], mode="bilinear", align_corners=True)
    y = torch.cat(br + [g], 1)
    y = F.relu(c.bn(F.conv2d(y, sd["aspp.conv1.weight"]), "aspp.bn1"))
    feature = c.dropout(y, "aspp.dropout", 0.5)
    # --- decoder (decoder.py:45-56)
    lo = F.relu(c.bn(F.conv2d(low, sd["decoder.conv1.weight"]), "decoder.bn1"))
    up = F.interpolate(feature, size=lo.shape[2:], mode="bilinear", align_corners=True)
    x_bu = torch.cat((up, lo), 1)
    b = F.conv2d(x_bu, sd["decoder.last_conv_boundary.0.weight"], None, 1, 1)
    b = F.relu(c.bn(b, "decoder.last_conv_boundary.1"))
    b = c.dropout(b, "decoder.last_conv_boundary.3", 0.5)
    b = F.conv2d(b, sd["decoder.last_conv_boundary.4.weight"], None, 1, 1)
    b = F.relu(c.bn(b, "decoder.last_conv_boundary.5"))
    b = c.dropout(b, "decoder.last_conv_boundary.7", 0.1)
    x2_before = F.conv2d(b, sd["decoder.last_conv_boundary.8.weight"],
                         sd["decoder.last_conv_boundary.8.bias"])
    x_feature = torch.cat((x_bu, x2_before), 1)
    s = F.relu(c.bn(x_feature, "decoder.last_conv.0"))
    s = c.dropout(s, "decoder.last_conv.2", 0.1)
    x1_before = F.conv2d(s, sd["decoder.last_conv.3.weight"], sd["decoder.last_conv.3.bias"])
    # --- heads (deeplabv3.py:39-40)
    size = x.shape[2:]
    x2 = F.interpolate(x2_before, size=size, mode="bilinear", align_corners=True)
    x1 = F.interpolate(x1_before, size=size, mode="bilinear", align_corners=True)
    return x1, x2, feature, x_bu, x_feature, x1_before, x2_before


def dropout_mask_shapes(batch, height, width):
    """Shapes (NCHW) of the four keep-masks for an input of batch x 3 x height x width."""
    h16, w16, h4, w4 = height // 16, width // 16, height // 4, width // 4
    return {
        "aspp.dropout": (batch, 256, h16, w16),
        "decoder.last_conv_boundary.3": (batch, 256, h4, w4),
        "decoder.last_conv_boundary.7": (batch, 256, h4, w4),
        "decoder.last_conv.2": (batch, 305, h4, w4),
    }


def draw_masks(batch, height, width, generator=None):
    """Seedable keep-masks (uint8) for parity tests."""
    ps = dict(DROPOUT_SITES)
    out = {}
    for name, shp in dropout_mask_shapes(batch, height, width).items():
        out[name] = (torch.rand(shp, generator=generator) >= ps[name]).to(torch.uint8)
    return out


def _is_running_stat(k):
    return k.rsplit(".", 1)[-1].startswith(("running_mean", "running_var"))     # incl. TransNorm's _source / _target


def canonical_state(sd, requires_grad=False):
    """Drop the aliased low_level_/high_level_ keys (mobilenet.py:116-117, quirk Q10)
    and return float32 clones; parameters optionally ``requires_grad``."""
    out = {}
    for k, v in sd.items():
        if ".low_level_features." in k or ".high_level_features." in k:
            continue
        t = v.detach().clone()
        if requires_grad and t.is_floating_point() and not (
                _is_running_stat(k)):
            t.requires_grad_(True)
        out[k] = t
    return out


def parameter_keys(sd):
    return [k for k, v in sd.items() if v.is_floating_point() and not _is_running_stat(k)]


class OracleDeepLab(torch.nn.Module):
    """Thin nn.Module face over the functional forward so the reference-shaped
    Trainer loops can drive the oracle on CPU (tests / cpu_baseline only)."""

    def __init__(self, state_dict, output_stride=16):
        super().__init__()
        sd = canonical_state(state_dict)
        self._keys = list(sd.keys())
        self.output_stride = output_stride
        self.masks = None          # inject keep-masks for the next forward
        self.record = None
        for k, v in sd.items():
            name = k.replace(".", "__")
            if k in parameter_keys(sd):
                self.register_parameter(name, torch.nn.Parameter(v))
            else:
                self.register_buffer(name, v)

    def flat_state(self):
        return {k: getattr(self, k.replace(".", "__")) for k in self._keys}

    def forward(self, x):
        return deeplab_forward(self.flat_state(), x, self.training, self.masks, self.record,
                               self.output_stride)
