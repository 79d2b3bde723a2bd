"""Device-side input pipeline: per-sample mixup / cutmix (reference
mix_dataset.py:27-90), Normalize and constant-mode RandomErasing (reference transforms.py:321-418) applied to a batch
that is already resident on the GPU, in one HIP kernel (csrc/input.hip).

The reference runs these per sample on the CPU inside the Dataset; here every random decision is drawn on the host
with the SAME generator calls in the SAME order (``plan_batch``: partner ``randrange`` loop, mixup for even / cutmix for
odd indices, ``betavariate`` / ``uniform`` ratio, ``rand_bbox``, then RandomErasing's ``random`` / ``uniform`` /
``randint`` sequence), so a seeded ``random.Random`` reproduces the reference's outputs; only the pixel work moves to
the device.  The "dataset" a partner is drawn from is the batch.
"""
import math
import random as _random
import struct

import torch

from . import ops


def rand_bbox(size, ratio, rng):
    w, h = size                                   # the reference unpacks (H, W) as (w, h): mix_dataset.py:10-11, 77
    r = math.sqrt(1 - ratio)
    cut_w, cut_h = int(w * r), int(h * r)
    cx, cy = rng.randrange(w), rng.randrange(h)
    x1 = min(max(cx - cut_w // 2, 0), w)
    y1 = min(max(cy - cut_h // 2, 0), h)
    x2 = min(max(cx + cut_w // 2, 0), w)
    y2 = min(max(cy + cut_h // 2, 0), h)
    return x1, y1, x2, y2


class ErasePlan:
    """Parameters of transforms.RandomErasing (mode 'const'); draws rectangles in the reference's order."""

    def __init__(self, p=0.5, min_area=0.02, max_area=1 / 3, min_aspect=0.3, max_aspect=None, min_count=1,
                 max_count=None, mode="const"):
        if mode not in ("const", "", None):
            raise NotImplementedError("vtx: RandomErasing modes 'rand' / 'pixel' are not implemented on the device")
        max_aspect = max_aspect or 1 / min_aspect
        self.p, self.min_area, self.max_area = p, min_area, max_area
        self.log_aspect = (math.log(min_aspect), math.log(max_aspect))
        self.min_count, self.max_count = min_count, max_count or min_count

    def draw(self, img_h, img_w, rng):
        rects = []
        if rng.random() > self.p:
            return rects
        area = img_h * img_w
        count = self.min_count if self.min_count == self.max_count else rng.randint(self.min_count, self.max_count)
        for _ in range(count):
            for _attempt in range(10):
                target_area = rng.uniform(self.min_area, self.max_area) * area / count
                aspect = math.exp(rng.uniform(*self.log_aspect))
                h = int(round(math.sqrt(target_area * aspect)))
                w = int(round(math.sqrt(target_area / aspect)))
                if w < img_w and h < img_h:
                    rects.append((rng.randint(0, img_h - h), rng.randint(0, img_w - w), h, w))
                    break
        return rects


def plan_batch(n, height, width, mixup, cutmix, erase=None, rng=None, indices=None):
    """Per-sample plans for a batch of n images: list of dicts (partner, mode, ratio (mixup weight), box, rects,
    label_ratio).  ``indices``: the dataset index of every sample (decides mixup vs cutmix by parity like the
    reference); default 0..n-1."""
    rng = rng or _random
    plans = []
    for k in range(n):
        index = k if indices is None else indices[k]
        apply_mixup, apply_cutmix = mixup > 0, cutmix > 0
        partner, mode, wgt, box, label_ratio = k, 0, 1.0, (0, 0, 0, 0), 1
        if n == 1:                      # a one-image batch has no partner inside the batch (the reference draws its
            apply_mixup = apply_cutmix = False   # partner from the whole dataset, mix_dataset.py:43-47): leave it unmixed
        if apply_mixup or apply_cutmix:
            partner = k
            while partner == k:         # partners come from WITHIN the batch
                partner = rng.randrange(n)
        if apply_mixup and apply_cutmix:
            if index % 2 == 0:
                apply_cutmix = False
            else:
                apply_mixup = False
        if apply_mixup:
            wgt = rng.betavariate(mixup, mixup)
            mode, label_ratio = 1, wgt
        if apply_cutmix:
            r = rng.uniform(0, 1) if cutmix == 1 else rng.betavariate(cutmix, cutmix)
            x1, y1, x2, y2 = rand_bbox((height, width), r, rng)
            mode, box = 2, (x1, y1, x2, y2)
            label_ratio = 1 - ((x2 - x1) * (y2 - y1) / (height * width))
        rects = erase.draw(height, width, rng) if erase is not None else []
        plans.append(dict(partner=partner, mode=mode, ratio=wgt, box=box, rects=rects, label_ratio=label_ratio))
    return plans


class DeviceMixPipeline:
    """batch (N, C, H, W) uint8 or fp32 on the GPU + labels (N,)  ->  (fp32 normalised batch, label1, label2, ratio):
    the tuple the reference's train step consumes (train.py:270-272)."""

    def __init__(self, mixup=0.2, cutmix=1, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), erase=None, seed=None):
        self.mixup, self.cutmix, self.erase = mixup, cutmix, erase
        self.mean, self.std = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)
        self.rng = _random.Random(seed) if seed is not None else _random
        self._ring, self._slot = [], 0            # pinned staging buffers for the plan table (asynchronous upload)

    def pack(self, plans):
        maxr = ops.mix_max_rects()
        recs = []
        for p in plans:
            rects = p["rects"]
            if len(rects) > maxr:
                raise ops.VtxError(f"vtx: at most {maxr} erase rectangles per image")
            pad = [(0, 0, 0, 0)] * (maxr - len(rects))
            rr = list(rects) + pad
            x1, y1, x2, y2 = p["box"]
            recs.append(struct.pack("<iifiiiii4i4i4h4h", p["partner"], p["mode"], p["ratio"], x1, y1, x2, y2, len(rects),
                                    *[r[0] for r in rr], *[r[1] for r in rr], *[r[2] for r in rr], *[r[3] for r in rr]))
        assert len(recs[0]) == ops.mix_plan_bytes()
        return torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8)

    def upload(self, host_plan, dev):
        """Plan table -> device without stalling the host: a pageable host-to-device copy would serialise the host with
        the GPU stream every step; a ring of 4 pinned buffers (each guarded by an event) keeps the copy asynchronous."""
        n = host_plan.numel()
        if not self._ring or self._ring[0][0].numel() < n:
            self._ring = [(torch.empty(n, dtype=torch.uint8).pin_memory(), torch.cuda.Event()) for _ in range(4)]
            self._slot = 0
            for _, ev in self._ring:
                ev.record()
        buf, ev = self._ring[self._slot]
        self._slot = (self._slot + 1) % len(self._ring)
        ev.synchronize()                          # the copy issued 4 calls ago has long finished
        buf[:n].copy_(host_plan)
        out = buf[:n].to(dev, non_blocking=True)
        ev.record()
        return out

    def __call__(self, images, labels, indices=None):
        n, c, h, w = images.shape
        plans = plan_batch(n, h, w, self.mixup, self.cutmix, self.erase, self.rng, indices)
        dev = images.device
        plan = self.upload(self.pack(plans), dev)
        if self.mean.device != dev:
            self.mean, self.std = self.mean.to(dev), self.std.to(dev)
        out = ops.mix_normalize_erase(images, plan, self.mean, self.std)
        partner = torch.tensor([p["partner"] for p in plans], device=labels.device)
        ratio = torch.tensor([p["label_ratio"] for p in plans], dtype=torch.float32, device=dev)
        return out, labels, labels[partner], ratio
