"""Config defaults of the CenterMask2 inference path, as one nested table.

Two sources: (1) detectron2's defaults for the keys the hot path reads and the yaml does not override (SURVEY Appendix B,
last row; d2's own defaults file is not in the reference tree), (2) the keys the reference adds on top of them in
centermask/config/defaults.py:9-86 — same names, same values.  Keys marked (train) only exist so that reference yaml files
merge without "non-existent key" errors; the inference path never reads them.
"""
from .cfgnode import CfgNode

_FCOS = dict(  # reference defaults.py:14-50
    NUM_CLASSES=80, IN_FEATURES=["p3", "p4", "p5", "p6", "p7"], FPN_STRIDES=[8, 16, 32, 64, 128], PRIOR_PROB=0.01,
    INFERENCE_TH_TRAIN=0.05, INFERENCE_TH_TEST=0.05, NMS_TH=0.6,
    PRE_NMS_TOPK_TRAIN=1000, PRE_NMS_TOPK_TEST=1000, POST_NMS_TOPK_TRAIN=100, POST_NMS_TOPK_TEST=100,
    TOP_LEVELS=2, NORM="GN", USE_SCALE=True, THRESH_WITH_CTR=False,
    LOSS_ALPHA=0.25, LOSS_GAMMA=2.0, SIZES_OF_INTEREST=[64, 128, 256, 512], USE_RELU=True, USE_DEFORMABLE=False,       # (train) mostly
    NUM_CLS_CONVS=4, NUM_BOX_CONVS=4, NUM_SHARE_CONVS=0, CENTER_SAMPLE=True, POS_RADIUS=1.5, LOC_LOSS_TYPE="giou",
)
_VOVNET = dict(  # reference defaults.py:56-67
    CONV_BODY="V-39-eSE", OUT_FEATURES=["stage2", "stage3", "stage4", "stage5"], NORM="FrozenBN", OUT_CHANNELS=256,
    BACKBONE_OUT_CHANNELS=256, STAGE_WITH_DCN=(False, False, False, False), WITH_MODULATED_DCN=False, DEFORMABLE_GROUPS=1,
)
_TABLE = dict(
    VERSION=2,
    OUTPUT_DIR="./output",
    MODEL=dict(
        DEVICE="cuda", META_ARCHITECTURE="GeneralizedRCNN", WEIGHTS="", LOAD_PROPOSALS=False,
        MASK_ON=False, KEYPOINT_ON=False, MASKIOU_ON=False, MASKIOU_LOSS_WEIGHT=1.0, MOBILENET=False,
        PIXEL_MEAN=[103.530, 116.280, 123.675], PIXEL_STD=[1.0, 1.0, 1.0],
        BACKBONE=dict(NAME="build_resnet_backbone", FREEZE_AT=2),
        FPN=dict(IN_FEATURES=[], OUT_CHANNELS=256, NORM="", FUSE_TYPE="sum"),
        PROPOSAL_GENERATOR=dict(NAME="RPN", MIN_SIZE=0),
        FCOS=_FCOS,
        VOVNET=_VOVNET,
        ROI_HEADS=dict(NAME="Res5ROIHeads", NUM_CLASSES=80, IN_FEATURES=["res4"], IOU_THRESHOLDS=[0.5], IOU_LABELS=[0, 1],
                       BATCH_SIZE_PER_IMAGE=512, POSITIVE_FRACTION=0.25, SCORE_THRESH_TEST=0.05, NMS_THRESH_TEST=0.5,
                       PROPOSAL_APPEND_GT=True),
        ROI_MASK_HEAD=dict(NAME="MaskRCNNConvUpsampleHead", POOLER_RESOLUTION=14, POOLER_SAMPLING_RATIO=0, POOLER_TYPE="ROIAlignV2",
                           NUM_CONV=0, CONV_DIM=256, NORM="", CLS_AGNOSTIC_MASK=False, ASSIGN_CRITERION="area"),
        ROI_MASKIOU_HEAD=dict(NAME="MaskIoUHead", CONV_DIM=256, NUM_CONV=4),
        ROI_KEYPOINT_HEAD=dict(NAME="KRCNNConvDeconvUpsampleHead", IN_FEATURES=["p2", "p3", "p4", "p5"], ASSIGN_CRITERION="ratio"),
    ),
    INPUT=dict(MIN_SIZE_TRAIN=(800,), MIN_SIZE_TEST=800, MAX_SIZE_TEST=1333, FORMAT="BGR"),
    DATASETS=dict(TRAIN=(), TEST=()),
    DATALOADER=dict(NUM_WORKERS=4),
    TEST=dict(DETECTIONS_PER_IMAGE=100),
    SOLVER=dict(CHECKPOINT_PERIOD=5000, IMS_PER_BATCH=16, BASE_LR=0.001, STEPS=(30000,), MAX_ITER=40000),
)

_C = CfgNode(_TABLE)
