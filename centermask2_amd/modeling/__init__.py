"""Plugin registration happens on import (as `import centermask` does for the reference)."""
from .backbone import FPN, VoVNet, build_fcos_vovnet_fpn_backbone, build_vovnet_backbone
from .centermask import CenterROIHeads, MaskIoUHead, ROIPooler, SpatialAttentionMaskHead, build_mask_head, build_maskiou_head
from .fcos import FCOS, FCOSHead
from .meta_arch import GeneralizedRCNN, build_backbone, build_model, build_proposal_generator, build_roi_heads, flatten_to_tuple
