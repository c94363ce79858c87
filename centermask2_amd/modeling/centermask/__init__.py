from .center_heads import (CenterROIHeads, MaskIoUHead, ROIPooler, SpatialAttentionMaskHead, build_mask_head,
                           build_maskiou_head)
