"""sam-textvqa_amd — MI355X-native SA-M4C hot path (spatially-aware multimodal transformer training).

Layout:
  csrc/      hand-written HIP kernels for gfx950 + the C-ABI entry points declared in include/sam_hip.h
  _capi.py   ctypes binding of libsam_hip.so (raises if the library is missing: there is no CPU fallback)
  ops.py     torch.autograd.Function wrappers: device pointers + current stream handed to the C-ABI
  modules.py host-side mirror of the reference's nn.Module surface (sam/sa_m4c.py)
  trainer.py train-step harness (loss / clip / Adam / LR schedule semantics of train.py + task_utils.py)
"""
__version__ = "0.1.0"
