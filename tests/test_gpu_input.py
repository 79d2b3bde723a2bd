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


@pytest.mark.parametrize("tag,mixup,cutmix,seed,mode", [("pixel", 0.2, 1, 8, "pixel"), ("rand", 0.2, 1, 9, "rand"),
                                                         ("pixel_only", 0.0, 0, 10, "pixel")])
def test_device_pipeline_erase_colour_modes_vs_reference(tag, mixup, cutmix, seed, mode):
    """RandomErasing 'pixel' / 'rand' on the device: the normal draws are made on the host in the reference's order and
    shipped as a table, so erased pixels equal the reference's BIT FOR BIT (golden G9b); the rest to 2 ulp."""
    from vtx.input_pipeline import DeviceMixPipeline, ErasePlan
    g = Golden("g9b_erase_modes")
    images = torch.stack([fill((3, H, W), 900 + i, 0.5, 0.5) for i in range(N)]).to(dev())
    labels = torch.arange(10, 10 + N, device=dev())
    erase = ErasePlan(p=0.8, max_count=2, mode=mode, generator=torch.Generator().manual_seed(1000 + seed))
    pipe = DeviceMixPipeline(mixup, cutmix, erase=erase, seed=seed)
    out, l1, l2, ratio = pipe(images, labels)
    ref = torch.from_numpy(g.arr(f"{tag}.images"))
    check(f"input pipeline erase mode {tag}", out, ref, 3e-7)
    if tag == "pixel_only":            # no mixing: everything outside the rectangles is (x - mean) / std, inside = the draws
        plain = ((images.cpu() - torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1))
        erased = ref != plain
        assert erased.any() and torch.equal(out.cpu()[erased], ref[erased]), "erased pixels must be the reference's draws exactly"
    assert torch.equal(l2.cpu(), torch.from_numpy(g.arr(f"{tag}.label2")))


def test_nhwc_bf16_output_feeds_the_patch_embeddings_bit_for_bit():
    """output="nhwc_bf16": the pipeline writes the batch once, in the layout and dtype the patch gathers consume (SURVEY
    F4's purpose).  The bf16 values are the fp32 output's roundings, and Swin / ViT run on that tensor give the SAME bits
    as on the fp32 NCHW batch under bf16 autocast (the single rounding just happens one kernel earlier)."""
    from models import SwinTransformer, VisionTransformer
    from vtx.input_pipeline import DeviceMixPipeline, ErasePlan
    from vtx.nn import Linear
    d = dev()
    gen = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (4, 3, 224, 224), generator=gen, dtype=torch.uint8).to(d)
    labels = torch.randint(0, 10, (4,), generator=gen).to(d)
    mk = lambda output: DeviceMixPipeline(0.2, 1, erase=ErasePlan(p=0.9, mode="pixel", generator=torch.Generator().manual_seed(5)),
                                          seed=3, output=output)
    a = mk("nchw_fp32")(u8, labels)[0]
    b = mk("nhwc_bf16")(u8, labels)[0]
    assert b.shape == a.shape and b.dtype == torch.bfloat16 and b.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(a.to(torch.bfloat16), b.contiguous())
    torch.manual_seed(0)
    swin = SwinTransformer(image_size=(224, 224), n_class=10, depths=(1, 1, 1, 1), dims=(96, 192, 384, 768), dim_head=32,
                           n_heads=(3, 6, 12, 24), dim_ffs=(384, 768, 1536, 3072), window_size=7).to(d).train()
    vit = VisionTransformer(Linear(384, 10), 224, 16, 1, 384, 6, 1536, 0.0, 0.0, 0.0, 0.0).to(d).train()
    for model in (swin, vit):
        outs, grads = [], []
        for x in (a, b):
            model.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = model(x)
            y.float().square().sum().backward()
            outs.append(y)
            grads.append(next(model.parameters()).grad.clone())
        assert torch.equal(outs[0], outs[1]) and torch.equal(grads[0], grads[1])
