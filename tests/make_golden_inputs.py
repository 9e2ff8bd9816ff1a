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
