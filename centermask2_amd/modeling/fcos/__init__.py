from .fcos import FCOS, FCOSHead, Scale, instances_from_padded
