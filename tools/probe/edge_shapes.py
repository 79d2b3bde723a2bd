import sys, os
sys.path.insert(0, "vision-transformers-pytorch_amd")
import torch
from models import SwinTransformer, VisionTransformer
from vtx.nn import Linear
dev = torch.device("cuda")
def run(name, f):
    try:
        out = f()
        print(name, "OK", tuple(out.shape), bool(torch.isfinite(out.float()).all()))
    except Exception as e:
        print(name, "->", type(e).__name__, str(e)[:160])
with torch.autocast("cuda", dtype=torch.bfloat16):
    run("swin 448 win7", lambda: SwinTransformer(image_size=(448, 448), n_class=16, depths=(1,1,1,1), dims=(32,64,128,256), dim_head=32, n_heads=(1,2,4,8), dim_ffs=(64,128,256,512), window_size=7).to(dev)(torch.randn(2,3,448,448,device=dev)))
    run("swin 384 win12", lambda: SwinTransformer(image_size=(384, 384), n_class=16, depths=(1,1,1,1), dims=(32,64,128,256), dim_head=32, n_heads=(1,2,4,8), dim_ffs=(64,128,256,512), window_size=12).to(dev)(torch.randn(2,3,384,384,device=dev)))
    run("swin 224x448", lambda: SwinTransformer(image_size=(224, 448), n_class=16, depths=(1,1,1,1), dims=(32,64,128,256), dim_head=32, n_heads=(1,2,4,8), dim_ffs=(64,128,256,512), window_size=7).to(dev)(torch.randn(2,3,224,448,device=dev)))
    run("vit 384 (L=577)", lambda: VisionTransformer(Linear(128,16), 384, 16, 2, 128, 2, 256, 0.,0.,0.,0.).to(dev)(torch.randn(2,3,384,384,device=dev)))
    run("vit 160 (L=101)", lambda: VisionTransformer(Linear(128,16), 160, 16, 2, 128, 2, 256, 0.,0.,0.,0.).to(dev)(torch.randn(2,3,160,160,device=dev)))
    run("vit dim_head 32", lambda: VisionTransformer(Linear(128,16), 224, 16, 2, 128, 4, 256, 0.,0.,0.,0.).to(dev)(torch.randn(2,3,224,224,device=dev)))
    run("vit eval no_grad", lambda: VisionTransformer(Linear(128,16), 224, 16, 2, 128, 2, 256, 0.,0.,0.,0.).to(dev).eval()(torch.randn(2,3,224,224,device=dev)))
