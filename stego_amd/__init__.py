"""stego_amd - MI355X-native (gfx950) implementation of STEGO's feature-correspondence
distillation hot path (reference: mhamilton723/STEGO, src/modules.py:275-398), behind the
reference's own Python API.  Compute = hand-written HIP kernels in ``csrc/`` reached through
the C ABI of ``include/stego_corr.h`` (correspondence loss, KNN top-k, dense correspondence) and
``include/stego_vit.h`` (the frozen DINO ViT forward that produces the feature maps); PyTorch-ROCm
supplies device memory, streams, autograd plumbing and torch.distributed (RCCL)."""
from .modules import (ClusterLookup, ContrastiveCorrelationLoss, ContrastiveCRFLoss, DinoFeaturizer,  # noqa: F401
                      FeaturePyramidNet, LambdaLayer, average_norm, norm, sample, sample_nonzero_locations,
                      super_perm, tensor_correlation)

__all__ = ["ContrastiveCorrelationLoss", "DinoFeaturizer", "FeaturePyramidNet", "ClusterLookup", "ContrastiveCRFLoss",
           "LambdaLayer", "norm", "average_norm", "tensor_correlation", "sample", "super_perm",
           "sample_nonzero_locations"]
