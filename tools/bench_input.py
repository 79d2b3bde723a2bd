#!/usr/bin/env python3
"""Micro-benchmark of the device-side input pipeline kernel (GPU box only): 128 x 3 x 224 x 224 uint8 -> fp32."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
import torch

from vtx.input_pipeline import DeviceMixPipeline, ErasePlan

dev = torch.device("cuda")
for dtype in (torch.uint8, torch.float32):
    x = torch.randint(0, 256, (128, 3, 224, 224), device=dev).to(dtype)
    y = torch.randint(0, 1000, (128,), device=dev)
    pipe = DeviceMixPipeline(0.2, 1, erase=ErasePlan(p=0.25), seed=0)
    for _ in range(3):
        pipe(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = pipe(x, y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = x.numel() * x.element_size() * 2 + out[0].numel() * 4      # own + partner image read, fp32 written
    print(f"{dtype}: {us:7.1f} us per batch of 128 (host planning included), algorithmic {nbytes / us / 1e3:6.0f} GB/s")
