"""vtx: MI355X-native (gfx950) runtime of the ViT / Swin training hot path.

  vtx._lib        ctypes binding of libvtx.so (C ABI in include/vtx.h) -- no fallback
  vtx.ops         tensor-level wrappers, one per C entry point
  vtx.functional  autograd.Functions (fused forward/backward kernel sequences)
  vtx.nn          nn.Linear / nn.LayerNorm parameter containers with HIP forwards
  vtx.tables      integer window tables (pos / local_mask), bit-exact vs the reference
  vtx.ddp         data-parallel gradient all-reduce over RCCL on a side stream
"""
