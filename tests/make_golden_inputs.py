"""Seeded synthetic targets shared by tests/golden/make_golden.py consumers (same arithmetic as the
generator script's ``synth_targets`` so the fixtures' inputs can be rebuilt from their seeds)."""
import torch


def synth_targets(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    maps, bds = [], []
    for _ in range(B):
        cy, cx = (0.4 + 0.2 * torch.rand(2, generator=g)) * torch.tensor([H, W])
        a, b = (0.18 + 0.09 * torch.rand(2, generator=g)) * min(H, W)
        k = 0.4 + 0.3 * torch.rand(1, generator=g)
        r = torch.sqrt(((yy - cy) / a) ** 2 + ((xx - cx) / b) ** 2)
        disc, cup = (r <= 1).float(), (r <= k).float()
        ring = torch.exp(-((r - 1) * min(a, b) / 3.0) ** 2) + torch.exp(-((r - k) * min(a, b) / 3.0) ** 2)
        maps.append(torch.stack([cup, disc]))
        bds.append(ring.clamp(0, 1)[None])
    return torch.stack(maps), torch.stack(bds)


def synth_loader(n_batches, B, S, seed):
    out = []
    for i in range(n_batches):
        g = torch.Generator().manual_seed(seed + i)
        tmap, tbd = synth_targets(B, S, S, seed + 100 + i)
        img = (torch.rand(B, 3, S, S, generator=g) * 2 - 1) * 0.5 + (tmap[:, 1:2] * 0.3 + tmap[:, 0:1] * 0.3)
        out.append({"image": img, "map": tmap, "boundary": tbd, "img_name": ["s%d" % i] * B})
    return out


def fundus_u8(B, H, W, seed):
    """Seeded uint8 image batch [B,H,W,3] + grey-coded mask [B,H,W] (255 background, 128 disc rim, 0 cup): ellipses (the first
    sample's touches the image border), isolated pixels on the 50 / 51 / 200 / 201 class thresholds, smooth image content plus
    noise.  numpy's legacy RandomState stream (frozen by numpy's compatibility policy), so a seed rebuilds the input."""
    import numpy as np
    rs = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    img = np.zeros((B, H, W, 3), np.uint8)
    lab = np.full((B, H, W), 255, np.uint8)
    for b in range(B):
        cy, cx = (0.15 if b == 0 else rs.uniform(0.35, 0.65)) * H, rs.uniform(0.35, 0.65) * W
        a, c = rs.uniform(0.2, 0.35) * H, rs.uniform(0.2, 0.35) * W
        r = np.sqrt(((yy - cy) / a) ** 2 + ((xx - cx) / c) ** 2)
        lab[b][r <= 1.0] = 128
        lab[b][r <= rs.uniform(0.4, 0.7)] = 0
        lab[b][rs.rand(H, W) < 0.002] = rs.choice([0, 50, 51, 60, 128, 200, 201, 255])
        lab[b][0, :7] = 50
        lab[b][-1, -9:] = 51
        lab[b][H // 2, :5] = 200
        lab[b][:6, -1] = 201
        smooth = 110.0 + 90.0 * np.exp(-r * r)[..., None] * np.array([1.0, 0.7, 0.4]) + 40.0 * np.sin(xx / 7.0 + b)[..., None]
        img[b] = np.clip(smooth + rs.randint(-25, 26, (H, W, 3)), 0, 255).astype(np.uint8)
    return img, lab
