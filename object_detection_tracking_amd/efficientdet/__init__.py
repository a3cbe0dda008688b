"""EfficientDet path (detector half) -- work in progress: the EfficientNet
backbone (reference efficientdet/backbone/efficientnet_{builder,model}.py) runs on the HIP kernels;
BiFPN, the class/box nets and the detection tail are not built yet."""
from .arch import backbone_spec, efficientnet_params, synthetic_backbone_weights  # noqa: F401
from .backbone import EfficientNetBackbone  # noqa: F401
