"""The input-pipeline fixtures the REFERENCE's dataloaders/custom_transforms.py wrote (tests/golden/input_pipeline.json:
seeds + SHA-256 of every output array; input_pipeline_small.npz: the small cases in full) and helpers to re-run a case
on this repo's CPU chain (uda_clr_amd.dataloaders.custom_transforms) or on the HIP kernels."""
import hashlib
import json
import os
import random

import numpy as np

from make_golden_inputs import fundus_u8

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(GOLDEN, "input_pipeline.json")))
SMALL = np.load(os.path.join(GOLDEN, "input_pipeline_small.npz"))


def digest(a):
    a = np.ascontiguousarray(a)
    return "%s%s:%s" % (a.dtype.str, list(a.shape), hashlib.sha256(a.tobytes()).hexdigest())


def cases(prefix):
    return sorted(k for k in META if k.startswith(prefix + "."))


def expect(name, arrays):
    """Byte-for-byte: dtype, shape and SHA-256 of every array; when the fixture holds the array in full, say where it differs."""
    for k, v in arrays.items():
        v = np.ascontiguousarray(v)
        want = META[name]["sha256"][k]
        if digest(v) != want:
            full = "%s.%s" % (name, k)
            where = ""
            if full in SMALL.files and SMALL[full].shape == v.shape:
                bad = np.argwhere(SMALL[full] != v)
                where = ": %d of %d elements differ, first at %s (got %r, reference %r)" % (
                    len(bad), v.size, bad[0].tolist(), v[tuple(bad[0])], SMALL[full][tuple(bad[0])])
            raise AssertionError("%s.%s differs from the reference's output (%s vs %s)%s" % (name, k, digest(v)[:40], want[:40], where))


class seeded_noise:
    """While active, np.random.RandomState(None) (what elastic_transform seeds its displacement noise from) is RandomState(seed)."""
    def __init__(self, seed):
        self.seed = seed

    def __enter__(self):
        self.RS = np.random.RandomState
        np.random.RandomState = lambda s=None, _k=self.seed, _RS=self.RS: _RS(_k)

    def __exit__(self, *a):
        np.random.RandomState = self.RS


def elastic_noise(seed, H, W):
    """The two uniform fields elastic_transform draws (custom_transforms.py:116-117), float64 in [-1, 1)."""
    rs = np.random.RandomState(seed)
    return np.stack([rs.rand(H, W) * 2 - 1, rs.rand(H, W) * 2 - 1])


def seed_streams(info):
    random.seed(info["py_seed"])
    if "np_seed" in info:
        np.random.seed(info["np_seed"])


def sample_of(info):
    img, lab = fundus_u8(info.get("B", 1), info.get("H", 96), info.get("W", 96), info["seed"])
    b = info.get("b", 0)
    return img[b], lab[b]
