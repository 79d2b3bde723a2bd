"""Pin the input-pipeline part of the oracle (oracle/ref_ops.py mix_dataset_item / random_erasing_const) against the
reference's MixDataset + RandomErasing outputs (golden G9) and the product's host planner (vtx.input_pipeline.plan_batch,
same generator calls in the same order) against the oracle -- SURVEY section 8 row F4.  CPU only."""
import random

import numpy as np
import torch

from golden_util import Golden
from oracle import ref_ops as R
from oracle.formula import fill

N, H, W = 8, 16, 20
MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
CASES = (("both", 0.2, 1, 5), ("beta_cutmix", 0.0, 0.5, 6), ("mixup_only", 0.8, 0, 7))


def dataset():
    return [fill((3, H, W), 900 + i, 0.5, 0.5) for i in range(N)], list(range(10, 10 + N))


def test_oracle_reproduces_reference_mix_and_erase():
    g = Golden("g9_input_pipeline")
    images, labels = dataset()
    for tag, mixup, cutmix, seed in CASES:
        rng = random.Random(seed)
        tf = lambda img: R.random_erasing_const((img - MEAN) / STD, rng, p=0.7, max_count=2)
        for i in range(N):
            img, l1, l2, r = R.mix_dataset_item(images, labels, i, mixup, cutmix, tf, rng)
            assert np.array_equal(img.numpy(), g.arr(f"{tag}.images")[i]), f"{tag} image {i}"
            assert (l1, l2) == (int(g.arr(f"{tag}.label1")[i]), int(g.arr(f"{tag}.label2")[i]))
            assert float(r) == float(g.arr(f"{tag}.ratio")[i])


def test_host_planner_draws_like_the_reference():
    """plan_batch consumes the generator exactly like MixDataset + RandomErasing: applying its plans with torch ops gives
    the reference's images bit for bit (cutmix / erase are copies; mixup is the same fp32 expression)."""
    from vtx.input_pipeline import ErasePlan, plan_batch
    g = Golden("g9_input_pipeline")
    images, labels = dataset()
    for tag, mixup, cutmix, seed in CASES:
        plans = plan_batch(N, H, W, mixup, cutmix, ErasePlan(p=0.7, max_count=2), random.Random(seed))
        for i, p in enumerate(plans):
            a, b = images[i].clone(), images[p["partner"]]
            if p["mode"] == 1:
                a = a.mul(p["ratio"]).add_(b, alpha=1 - p["ratio"])
            elif p["mode"] == 2:
                x1, y1, x2, y2 = p["box"]
                a[:, y1:y2, x1:x2] = b[:, y1:y2, x1:x2]
            a = (a - MEAN) / STD
            for top, left, eh, ew, _colour in p["rects"]:
                a[:, top:top + eh, left:left + ew] = 0
            assert np.array_equal(a.numpy(), g.arr(f"{tag}.images")[i]), f"{tag} image {i}"
            assert labels[p["partner"]] == int(g.arr(f"{tag}.label2")[i])
            assert float(p["label_ratio"]) == float(g.arr(f"{tag}.ratio")[i])


MODE_CASES = (("pixel", 0.2, 1, 8, "pixel"), ("rand", 0.2, 1, 9, "rand"), ("pixel_only", 0.0, 0, 10, "pixel"))


def test_oracle_reproduces_reference_erase_colour_modes():
    """RandomErasing 'pixel' (the mode factory.py:177-181 configures) and 'rand': the colour draws come from torch's CPU
    generator, interleaved with the python `random` draws of the rectangles -- golden G9b (reference outputs)."""
    g = Golden("g9b_erase_modes")
    images, labels = dataset()
    for tag, mixup, cutmix, seed, mode in MODE_CASES:
        rng = random.Random(seed)
        tgen = torch.Generator().manual_seed(1000 + seed)
        tf = lambda img: R.random_erasing_const((img - MEAN) / STD, rng, p=0.8, max_count=2, mode=mode, tgen=tgen)
        for i in range(N):
            img, l1, l2, r = R.mix_dataset_item(images, labels, i, mixup, cutmix, tf, rng)
            assert np.array_equal(img.numpy(), g.arr(f"{tag}.images")[i]), f"{tag} image {i}"
            assert l2 == int(g.arr(f"{tag}.label2")[i]) and float(r) == float(g.arr(f"{tag}.ratio")[i])


def test_host_planner_draws_erase_colours_like_the_reference():
    from vtx.input_pipeline import ErasePlan, plan_batch
    g = Golden("g9b_erase_modes")
    images, labels = dataset()
    for tag, mixup, cutmix, seed, mode in MODE_CASES:
        erase = ErasePlan(p=0.8, max_count=2, mode=mode, generator=torch.Generator().manual_seed(1000 + seed))
        plans = plan_batch(N, H, W, mixup, cutmix, erase, random.Random(seed))
        assert any(p["rects"] for p in plans)
        for i, p in enumerate(plans):
            a, b = images[i].clone(), images[p["partner"]]
            if p["mode"] == 1:
                a = a.mul(p["ratio"]).add_(b, alpha=1 - p["ratio"])
            elif p["mode"] == 2:
                x1, y1, x2, y2 = p["box"]
                a[:, y1:y2, x1:x2] = b[:, y1:y2, x1:x2]
            a = (a - MEAN) / STD
            for top, left, eh, ew, colour in p["rects"]:
                a[:, top:top + eh, left:left + ew] = colour
            assert np.array_equal(a.numpy(), g.arr(f"{tag}.images")[i]), f"{tag} image {i}"
