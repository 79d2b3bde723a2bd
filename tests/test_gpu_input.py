"""GPU parity of the device-side input pipeline: one HIP pass over the batch vs the
reference's per-sample MixDataset + Normalize + RandomErasing outputs (golden G9)."""
import random

import numpy as np
import pytest
import torch

from golden_util import Golden
from gpu_util import check, dev
from oracle.formula import fill

pytestmark = pytest.mark.gpu
N, H, W = 8, 16, 20
CASES = (("both", 0.2, 1, 5), ("beta_cutmix", 0.0, 0.5, 6), ("mixup_only", 0.8, 0, 7))


@pytest.mark.parametrize("tag,mixup,cutmix,seed", CASES)
def test_device_pipeline_vs_reference(tag, mixup, cutmix, seed):
    from vtx.input_pipeline import DeviceMixPipeline, ErasePlan
    g = Golden("g9_input_pipeline")
    images = torch.stack([fill((3, H, W), 900 + i, 0.5, 0.5) for i in range(N)]).to(dev())
    labels = torch.arange(10, 10 + N, device=dev())
    pipe = DeviceMixPipeline(mixup, cutmix, erase=ErasePlan(p=0.7, max_count=2), seed=seed)
    out, l1, l2, ratio = pipe(images, labels)
    ref = torch.from_numpy(g.arr(f"{tag}.images"))
    # cutmix / erase are copies and zeros (exact); mixup and the normalisation are fp32 expressions evaluated with a
    # fused multiply-add and a reciprocal on the device: <= 2 ulp of the reference's values
    check(f"input pipeline {tag}", out, ref, 3e-7)
    assert torch.equal((out == 0), (ref == 0).to(dev())), "erased regions must match exactly"
    assert torch.equal(l1.cpu(), torch.from_numpy(g.arr(f"{tag}.label1")))
    assert torch.equal(l2.cpu(), torch.from_numpy(g.arr(f"{tag}.label2")))
    check(f"input pipeline {tag} ratio", ratio, torch.from_numpy(g.arr(f"{tag}.ratio")), 1e-7)


def test_device_pipeline_uint8_and_train_step_contract():
    """uint8 NCHW input (ToTensor's 1/255 scaling on the device) and the (input, label1, label2, ratio) tuple feeding
    MixLoss exactly as the reference's train step does (train.py:270-283)."""
    from vtx.input_pipeline import DeviceMixPipeline, ErasePlan
    from vtx.train_step import MixLoss
    d = dev()
    gen = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (6, 3, 32, 32), generator=gen, dtype=torch.uint8)
    labels = torch.randint(0, 10, (6,), generator=gen)
    a = DeviceMixPipeline(0.2, 1, erase=ErasePlan(p=0.5), seed=3)(u8.to(d), labels.to(d))
    b = DeviceMixPipeline(0.2, 1, erase=ErasePlan(p=0.5), seed=3)((u8.float() / 255).to(d), labels.to(d))
    check("uint8 vs float input", a[0], b[0], 1e-6)
    logits = torch.randn(6, 10, generator=gen).to(d)
    loss = MixLoss(0.1)(logits, a[1], a[2], a[3])
    assert torch.isfinite(loss).item()
