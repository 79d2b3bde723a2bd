"""Device-side input pipeline: per-sample mixup / cutmix (reference
mix_dataset.py:27-90), Normalize and RandomErasing (reference transforms.py:321-418; all three colour modes: 'const',
'rand' and the 'pixel' mode factory.py:177-181 configures) applied to a batch that is already resident on the GPU, in
one HIP kernel (csrc/input.hip); output fp32 NCHW (the reference's contract) or bf16 NHWC for the models' patch gathers.

The reference runs these per sample on the CPU inside the Dataset; here every random decision is drawn on the host
with the SAME generator calls in the SAME order (``plan_batch``: partner ``randrange`` loop, mixup for even / cutmix for
odd indices, ``betavariate`` / ``uniform`` ratio, ``rand_bbox``, then RandomErasing's ``random`` / ``uniform`` /
``randint`` sequence; the erase colours of the 'rand' / 'pixel' modes with ``Tensor.normal_()`` of the same shapes from a
torch CPU generator), so a seeded ``random.Random`` (+ ``torch.Generator``) reproduces the reference's outputs; only the
pixel work moves to the device.  The "dataset" a partner is drawn from is the batch.
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
    """Parameters of transforms.RandomErasing; draws rectangles -- and, for the 'rand' / 'pixel' modes, their colours --
    in the reference's order.  ``generator``: torch CPU generator of the colour draws (None = torch's global one, which is
    what the reference's ``torch.empty(...).normal_()`` consumes, transforms.py:309-318)."""

    MODES = {"const": 0, "": 0, None: 0, "rand": 1, "pixel": 2}

    def __init__(self, p=0.5, min_area=0.02, max_area=1 / 3, min_aspect=0.3, max_aspect=None, min_count=1,
                 max_count=None, mode="const", generator=None):
        mode = mode.lower() if isinstance(mode, str) else mode
        if mode not in self.MODES:
            raise ValueError(f"RandomErasing mode {mode!r} (const | rand | pixel)")
        max_aspect = max_aspect or 1 / min_aspect
        self.p, self.min_area, self.max_area = p, min_area, max_area
        self.log_aspect = (math.log(min_aspect), math.log(max_aspect))
        self.min_count, self.max_count = min_count, max_count or min_count
        self.fmode, self.generator = self.MODES[mode], generator

    def colour(self, chan, h, w):
        """_get_pixels (transforms.py:309-318): (chan, h, w) normal draws per pixel, (chan, 1, 1) per block, or zeros."""
        if self.fmode == 2:
            return torch.empty((chan, h, w), dtype=torch.float32).normal_(generator=self.generator)
        if self.fmode == 1:
            return torch.empty((chan, 1, 1), dtype=torch.float32).normal_(generator=self.generator)
        return None

    def draw(self, img_h, img_w, rng, chan=3):
        """-> [(top, left, h, w, colour tensor or None)]"""
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
                    top, left = rng.randint(0, img_h - h), rng.randint(0, img_w - w)
                    rects.append((top, left, h, w, self.colour(chan, h, w)))
                    break
        return rects


def plan_batch(n, height, width, mixup, cutmix, erase=None, rng=None, indices=None, chan=3):
    """Per-sample plans for a batch of n images: list of dicts (partner, mode, ratio (mixup weight), box, rects =
    [(top, left, h, w, colour)], label_ratio).  ``indices``: the dataset index of every sample (decides mixup vs cutmix
    by parity like the reference); default 0..n-1."""
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
        rects = erase.draw(height, width, rng, chan) if erase is not None else []
        plans.append(dict(partner=partner, mode=mode, ratio=wgt, box=box, rects=rects, label_ratio=label_ratio))
    return plans


class DeviceMixPipeline:
    """batch (N, C, H, W) uint8 or fp32 on the GPU + labels (N,)  ->  (normalised batch, label1, label2, ratio): the tuple
    the reference's train step consumes (train.py:270-272).  ``output``: "nchw_fp32" (the reference's model input) or
    "nhwc_bf16" -- a bf16 tensor of shape (N, C, H, W) in channels-last memory that the HIP models' patch gathers read
    directly (same patch values bit for bit under bf16 autocast: the one rounding happens here instead of there)."""

    def __init__(self, mixup=0.2, cutmix=1, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), erase=None, seed=None,
                 output="nchw_fp32"):
        if output not in ("nchw_fp32", "nhwc_bf16"):
            raise ValueError(output)
        self.mixup, self.cutmix, self.erase, self.output = mixup, cutmix, erase, output
        self.mean, self.std = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)
        self.rng = _random.Random(seed) if seed is not None else _random
        self._ring, self._slot = {}, {}           # pinned staging buffers (asynchronous uploads), per table kind

    def pack(self, plans):
        """-> (plan table uint8 [N * vtx_mix_plan_bytes()], fill table fp32 or None)"""
        maxr = ops.mix_max_rects()
        fmode = self.erase.fmode if self.erase is not None else 0
        recs, fills, foff = [], [], 0
        for p in plans:
            rects = p["rects"]
            if len(rects) > maxr:
                raise ops.VtxError(f"vtx: at most {maxr} erase rectangles per image")
            rr = [r[:4] for r in rects] + [(0, 0, 0, 0)] * (maxr - len(rects))
            offs = [0] * maxr
            for i, r in enumerate(rects):
                if fmode and r[4] is not None:
                    offs[i] = foff
                    fills.append(r[4].reshape(-1))
                    foff += r[4].numel()
            x1, y1, x2, y2 = p["box"]
            recs.append(struct.pack("<iifiiiii4i4i4h4hi4i", p["partner"], p["mode"], p["ratio"], x1, y1, x2, y2, len(rects),
                                    *[r[0] for r in rr], *[r[1] for r in rr], *[r[2] for r in rr], *[r[3] for r in rr],
                                    fmode, *offs))
        assert len(recs[0]) == ops.mix_plan_bytes()
        table = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8)
        return table, (torch.cat(fills) if fills else None)

    def upload(self, host, dev, kind="plan"):
        """Host table -> device without stalling the host: a pageable host-to-device copy would serialise the host with
        the GPU stream every step; a ring of 4 pinned buffers (each guarded by an event) keeps the copy asynchronous."""
        n = host.numel()
        ring = self._ring.get(kind)
        if not ring or ring[0][0].numel() < n or ring[0][0].dtype != host.dtype:
            cap = max(n, 2 * (ring[0][0].numel() if ring else 0))
            ring = self._ring[kind] = [(torch.empty(cap, dtype=host.dtype).pin_memory(), torch.cuda.Event()) for _ in range(4)]
            self._slot[kind] = 0
            for _, ev in ring:
                ev.record()
        buf, ev = ring[self._slot[kind]]
        self._slot[kind] = (self._slot[kind] + 1) % len(ring)
        ev.synchronize()                          # the copy issued 4 calls ago has long finished
        buf[:n].copy_(host)
        out = buf[:n].to(dev, non_blocking=True)
        ev.record()
        return out

    def __call__(self, images, labels, indices=None):
        n, c, h, w = images.shape
        plans = plan_batch(n, h, w, self.mixup, self.cutmix, self.erase, self.rng, indices, chan=c)
        dev = images.device
        table, fills = self.pack(plans)
        plan = self.upload(table, dev)
        fills = self.upload(fills, dev, "fills") if fills is not None else None
        if self.mean.device != dev:
            self.mean, self.std = self.mean.to(dev), self.std.to(dev)
        out = ops.mix_normalize_erase(images, plan, self.mean, self.std, fills, nhwc_bf16=self.output == "nhwc_bf16")
        partner = torch.tensor([p["partner"] for p in plans], device=labels.device)
        ratio = torch.tensor([p["label_ratio"] for p in plans], dtype=torch.float32, device=dev)
        return out, labels, labels[partner], ratio
