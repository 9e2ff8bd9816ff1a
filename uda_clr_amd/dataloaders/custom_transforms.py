"""Sample transforms named as train_use_fix_initial.py:150-166 uses them.  Each takes and returns the
dict {'image', 'label', 'img_name'}; the geometric ones work on PIL images, the photometric ones on
HxWx3 uint8 arrays (after ``elastic_transform``), ``Normalize_tf`` + ``ToTensor`` end the chain with
float tensors image [3,H,W] in [-1,1], map [2,H,W] (ch0 cup subset of ch1 disc), boundary [1,H,W].

Semantics follow dataloaders/custom_transforms.py of the reference (probabilities, ranges, the grey-level
coding of the masks: >200 background, 51..200 disc rim, <=50 cup; boundary = |dilate - erode| ring of
width 5, Gaussian sigma 3), written against numpy / PIL / scipy only."""
import numbers
import os
import random

import numpy as np
import torch
from PIL import Image, ImageOps
from scipy import ndimage


# UDA_CLR_DEVICE_INPUT selects how much of the chain runs on the GPU (the entry script stays unchanged):
#   1  the deterministic tail (Normalize_tf + ToTensor: /127.5 - 1, mask decoding, boundary ring and its Gaussian blur - the
#      scipy.ndimage part of a worker's time): the workers hand over the uint8 image and the uint8 grey mask (4x fewer bytes over
#      PCIe), the Trainer decodes the whole batch with uda_normalize_tf, bit-identical to this file's CPU arithmetic;
#   2  additionally elastic_transform, add_salt_pepper_noise, adjust_light and eraser: a worker still makes every random draw
#      of those transforms, in the same order (same consumption of `random` / `np.random` as the CPU chain), but only RECORDS the
#      outcome (fire / noisy positions and value / gamma table / erased box and grey level); the Trainer applies them to the
#      batch in the chain's order (uda_field_smooth + uda_elastic_warp, uda_photometric_u8) before decoding.  Given the same
#      parameters the results equal the CPU chain's byte for byte; only the elastic displacement NOISE comes from the device
#      generator instead of numpy's (the CPU chain seeds that RandomState from OS entropy, so it has no reproducible stream).
# The PIL geometry (scale-crop, rotate, flip) stays on the workers.
DEVICE_TAIL = int(os.environ.get("UDA_CLR_DEVICE_INPUT", "0") or 0)


def _defer(sample, **rec):
    aug = dict(sample.get('_aug', {}))
    aug.update(rec)
    return aug


def _out(sample, image, label, aug=None):
    out = {'image': image, 'label': label, 'img_name': sample['img_name']}
    if aug is not None or '_aug' in sample:
        out['_aug'] = aug if aug is not None else sample['_aug']
    return out


class RandomCrop(object):
    def __init__(self, size, padding=0):
        self.size = (int(size), int(size)) if isinstance(size, numbers.Number) else size      # (h, w)
        self.padding = padding

    def __call__(self, sample):
        img, mask = sample['image'], sample['label']
        w, h = img.size
        th, tw = self.size
        if self.padding > 0 or w < tw or h < th:
            pad = int(max(self.padding, (tw - w) // 2 + 5, (th - h) // 2 + 5))
            img, mask = ImageOps.expand(img, border=pad, fill=0), ImageOps.expand(mask, border=pad, fill=255)
            w, h = img.size
        if (w, h) == (tw, th):
            return _out(sample, img, mask)
        x1, y1 = random.randint(0, w - tw), random.randint(0, h - th)
        box = (x1, y1, x1 + tw, y1 + th)
        return _out(sample, img.crop(box), mask.crop(box))


class RandomScaleCrop(object):
    """with p = 0.5 rescale each side independently by U(0.5, 1.5), then RandomCrop(size)"""
    def __init__(self, size):
        self.crop = RandomCrop(size)

    def __call__(self, sample):
        img, mask = sample['image'], sample['label']
        if random.random() > 0.5:
            w, h = int(random.uniform(0.5, 1.5) * img.size[0]), int(random.uniform(0.5, 1.5) * img.size[1])
            sample = _out(sample, img.resize((w, h), Image.BILINEAR), mask.resize((w, h), Image.NEAREST))
        return self.crop(sample)


class RandomRotate(object):
    """one multiple of 90 degrees drawn at construction, applied with p = 0.5"""
    def __init__(self, size=512):
        self.degree = random.randint(1, 4) * 90
        self.size = size

    def __call__(self, sample):
        if random.random() > 0.5:
            return _out(sample, sample['image'].rotate(self.degree, Image.BILINEAR), sample['label'].rotate(self.degree, Image.NEAREST))
        return sample


class RandomFlip(object):
    def __call__(self, sample):
        img, mask = sample['image'], sample['label']
        for op in (Image.FLIP_LEFT_RIGHT, Image.FLIP_TOP_BOTTOM):
            if random.random() < 0.5:
                img, mask = img.transpose(op), mask.transpose(op)
        return _out(sample, img, mask)


class elastic_transform(object):
    """Simard-style elastic deformation with p = 0.5 (alpha = 2*side, sigma = 0.08*side); always leaves
    numpy arrays behind (image HxWx3 uint8, label HxW uint8)."""
    def __call__(self, sample):
        image, label = np.array(sample['image']), np.array(sample['label'])
        fire = random.random() > 0.5
        if DEVICE_TAIL >= 2:
            return _out(sample, image, label, _defer(sample, elastic=fire))
        if fire:
            side = image.shape[1]
            alpha, sigma = side * 2, side * 0.08
            shape = image.shape[:2]
            rs = np.random.RandomState(None)
            dx = ndimage.gaussian_filter(rs.rand(*shape) * 2 - 1, sigma, mode='constant', cval=0) * alpha
            dy = ndimage.gaussian_filter(rs.rand(*shape) * 2 - 1, sigma, mode='constant', cval=0) * alpha
            gx, gy = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing='ij')
            idx = np.reshape(gx + dx, (-1, 1)), np.reshape(gy + dy, (-1, 1))
            warped = np.stack([ndimage.map_coordinates(image[:, :, c], idx, order=1).reshape(shape) for c in range(3)], -1)
            image = warped.astype(np.uint8)
            label = ndimage.map_coordinates(label, idx, order=1, mode='nearest').reshape(shape).astype(np.uint8)
        return _out(sample, image, label)


class add_salt_pepper_noise(object):
    """0.4 % of the pixels: salt (p = 0.25) or pepper (p = 0.25), else unchanged"""
    def __call__(self, sample):
        image = sample['image'].copy()
        amount, salt_vs_pepper = 0.004, 0.2
        seed = random.random()
        sp = None
        if seed > 0.5:
            n = int(np.ceil(amount * image.size * (salt_vs_pepper if seed > 0.75 else 1.0 - salt_vs_pepper)))
            # one draw per array axis, as the reference (custom_transforms.py:37,41): the third (channel) draw is never
            # used but consumes np.random, which the eraser reads next
            ys, xs, _ = [np.random.randint(0, i - 1, n) for i in image.shape]
            sp = (1 if seed > 0.75 else 0, ys, xs)
            if DEVICE_TAIL < 2:
                image[ys, xs, :] = sp[0]
        if DEVICE_TAIL >= 2:
            return _out(sample, image, sample['label'], _defer(sample, sp=sp))
        return _out(sample, image, sample['label'])


class adjust_light(object):
    """gamma correction with gamma ~ U(0.5, 3.5), p = 0.5 (a 256-entry lookup table)"""
    def __call__(self, sample):
        table = None
        if random.random() > 0.5:
            inv = 1.0 / (random.random() * 3 + 0.5)
            table = (((np.arange(256) / 255.0) ** inv) * 255).astype(np.uint8)
        if DEVICE_TAIL >= 2:
            return _out(sample, sample['image'], sample['label'], _defer(sample, lut=table))
        if table is not None:
            return _out(sample, table[np.asarray(sample['image']).astype(np.uint8)], sample['label'])
        return sample


class eraser(object):
    """random erasing, p = 0.5: a box of 2-6 % of the area, aspect 0.3-0.6, filled with one grey level"""
    def __call__(self, sample, s_l=0.02, s_h=0.06, r_1=0.3, r_2=0.6, v_l=0, v_h=255, pixel_level=False):
        image = sample['image']
        if random.random() > 0.5:
            return sample
        H, W, C = image.shape
        while True:
            s, r = np.random.uniform(s_l, s_h) * H * W, np.random.uniform(r_1, r_2)
            w, h = int(np.sqrt(s / r)), int(np.sqrt(s * r))
            left, top = np.random.randint(0, W), np.random.randint(0, H)
            if left + w <= W and top + h <= H:
                break
        if DEVICE_TAIL >= 2 and not pixel_level:
            return _out(sample, image, sample['label'], _defer(sample, erase=(top, left, h, w, int(np.random.uniform(v_l, v_h)))))
        image[top:top + h, left:left + w, :] = np.random.uniform(v_l, v_h, (h, w, C)) if pixel_level else np.random.uniform(v_l, v_h)
        return _out(sample, image, sample['label'])


def to_multilabel(pre_mask, classes=2):
    mask = np.zeros((pre_mask.shape[0], pre_mask.shape[1], classes))
    mask[pre_mask == 1] = [0, 1]
    mask[pre_mask == 2] = [1, 1]
    return mask


class GetBoundary(object):
    def __init__(self, width=5):
        self.width = width

    def __call__(self, mask):
        ring = np.zeros(mask.shape[:2], dtype=bool)
        for c in range(2):
            m = mask[:, :, c]
            d = ndimage.binary_dilation(m, iterations=self.width)
            e = ndimage.binary_erosion(m, iterations=self.width)
            ring |= d ^ e
        return ring.astype(np.uint8)


class Normalize_tf(object):
    """image -> [-1, 1]; grey-coded mask -> 2-channel map + soft boundary"""
    def __init__(self, mean=(0., 0., 0.), std=(1., 1., 1.)):
        self.mean, self.std = mean, std
        self.get_boundary = GetBoundary()

    def __call__(self, sample):
        if DEVICE_TAIL:
            out = {'image_u8': np.ascontiguousarray(np.array(sample['image']).astype(np.uint8)),
                   'label_u8': np.ascontiguousarray(np.array(sample['label']).astype(np.uint8)), 'img_name': sample['img_name']}
            aug = sample.get('_aug') or {}                      # no record = no transform fired
            if DEVICE_TAIL >= 2:                                # the recorded outcomes as fixed-size arrays (collate stacks them)
                maxn = int(np.ceil(0.004 * out['image_u8'].size * 0.8))
                pos = np.zeros((maxn, 2), np.int32)
                val, ys, xs = aug.get('sp') or (0, (), ())
                pos[:len(ys), 0], pos[:len(xs), 1] = ys, xs
                lut = aug.get('lut')
                out.update(aug_elastic=np.array([1 if aug.get('elastic') else 0], np.uint8), aug_sp_pos=pos,
                           aug_sp_n=np.array([len(ys)], np.int32), aug_sp_val=np.array([val], np.int32),
                           aug_lut=np.arange(256, dtype=np.uint8) if lut is None else lut.astype(np.uint8),
                           aug_erase=np.array(aug.get('erase') or (0, 0, 0, 0, 0), np.int32))
            return out
        img = np.array(sample['image']).astype(np.float32) / 127.5 - 1.0
        grey = np.array(sample['label']).astype(np.uint8)
        cls = np.full(grey.shape, 2, dtype=np.uint8)          # <= 50: cup
        cls[grey > 50] = 1                                     # 51..200: disc rim
        cls[grey > 200] = 0                                    # > 200: background
        mask = to_multilabel(cls)
        boundary = ndimage.gaussian_filter((self.get_boundary(mask) * 255).astype(np.uint8), sigma=3) / 255.0
        return {'image': img, 'map': mask, 'boundary': boundary[..., None], 'img_name': sample['img_name']}


class ToTensor(object):
    def __call__(self, sample):
        if 'image_u8' in sample:          # deferred tail: uint8 [H,W,3] + uint8 [H,W] (+ recorded augmentation outcomes)
            return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sample.items()}
        img = torch.from_numpy(np.ascontiguousarray(np.asarray(sample['image'], dtype=np.float32).transpose(2, 0, 1)))
        mp = torch.from_numpy(np.ascontiguousarray(np.asarray(sample['map']).astype(np.uint8).transpose(2, 0, 1))).float()
        bd = torch.from_numpy(np.ascontiguousarray(np.asarray(sample['boundary'], dtype=np.float64).transpose(2, 0, 1))).float()
        return {'image': img, 'map': mp, 'boundary': bd, 'img_name': sample['img_name']}
