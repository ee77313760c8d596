"""hnet — the multi-task (detection + semantic segmentation) model family of the reference (reference: hnet/), MI355X path.

What is built (BASELINE config 5): `segmentation.PanopticFeatureConnector` / `segmentation.PanopticSeg` with the
reference's constructor arguments and state_dict keys on HIP kernels (GroupNorm + ReLU, bilinear align_corners resize, Softmax2d + soft
dice: csrc/seg.hip; the 3x3 / 1x1 convolutions are the detector's conv kernels), and `hnet.HNet`: one conv backbone + pyramid shared
by a detection header and a segmentation header, mixed det + seg loss, one backward.

What is not: the reference's HNet takes its backbone from timm / a Swin transformer and its detection header from its own Mask R-CNN
on mmcv-style helpers — none of those packages exist here and the file hard-codes a three-GPU `.cuda(0)` / `.cuda(2)`
model split (hnet/hnet.py:178-180).  The backbone / pyramid / detection header of this HNet are therefore the metayolo ones."""
from .hnet import HNet  # noqa: F401
