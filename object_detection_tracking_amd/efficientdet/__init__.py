"""EfficientDet path (detector half) -- work in progress: the EfficientNet
backbone (reference efficientdet/backbone/efficientnet_{builder,model}.py) runs on the HIP kernels;
the feature network (BiFPN), class / box nets and the detection tail (top-k, decode, NMS, per-level
ROIAlign features) are built behind ``EfficientDet``; the input resize / padding of the reference's
dataloader is not (frames must have the network input size)."""
from .arch import backbone_spec, efficientnet_params, synthetic_backbone_weights  # noqa: F401
from .backbone import EfficientNetBackbone  # noqa: F401
from .model import EfficientDet, generate_anchors  # noqa: F401
