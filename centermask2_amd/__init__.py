"""centermask2_amd — MI355X-native CenterMask2 inference hot path (VoVNetV2-FPN, FCOS, CenterROIHeads)."""
__version__ = "0.1.0"
