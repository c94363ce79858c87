from .fpn import FPN, LastLevelP6, LastLevelP6P7
from .vovnet import VoVNet, build_fcos_vovnet_fpn_backbone, build_vovnet_backbone
