"""itermvs_amd -- MI355X-native IterMVS matching hot path.

Python host code on PyTorch-ROCm (device memory, streams, hipGraph capture, torch.distributed;
MIOpen only behind the autograd convolutions of the training graph) around ``libitermvs_hip.so``: hand-written gfx950 HIP kernels
behind a C ABI (include/itermvs_hip.h).  The package mirrors the reference's
``models`` interface for this path (``Pipeline``, ``full_loss``,
``differentiable_warping`` ...), see INTEGRATION.md.

Nothing is imported eagerly: ``itermvs_amd.synthetic`` / ``.schema`` / ``.shard``
are pure host logic, while ``.ops`` / ``.net`` need the HIP library and a GPU and
fail loudly without them (there is no CPU fallback by design).
"""
__version__ = "0.1.0"
