"""Config system with the key paths of the reference.

The reference extends detectron2's yacs `CfgNode` (pkg/config/config.py:7-85) and merges
`config/config.yaml`.  Neither yacs nor detectron2 is available, so this is a small attribute-dict
`CfgNode` with the same surface the reference's callers use: attribute access, `merge_from_file`,
`merge_from_list`, `clone`, `freeze`/`defrost`, `dump`.  `get_cfg()` holds detectron2's defaults for the
keys that steer the detection path; `get_planercnn_cfg_defaults` adds the reference's own keys.
"""
from __future__ import annotations

import copy
from ast import literal_eval
from typing import Any

import yaml


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode.IMMUTABLE]:
            raise AttributeError(f"Attempted to set {name} on an immutable CfgNode")
        self[name] = value

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _set_immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        out.__dict__[CfgNode.IMMUTABLE] = self.__dict__[CfgNode.IMMUTABLE]
        return out

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kw):
        return yaml.safe_dump(self.to_dict(), **kw)

    def _merge(self, other: dict):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                self[k] = _coerce(self.get(k), v)

    def merge_from_other_cfg(self, other):
        if self.is_frozen():
            raise AttributeError("immutable CfgNode")
        self._merge(other)

    def merge_from_file(self, path, allow_unsafe: bool = False):
        """Accepts the reference's config/*.yaml verbatim (VERSION: 2, anchors such as `*id001`)."""
        if self.is_frozen():
            raise AttributeError("immutable CfgNode")
        with open(path, "r") as f:
            data = yaml.safe_load(f) or {}
        base = data.pop("_BASE_", None)
        if base:
            import os

            self.merge_from_file(base if os.path.isabs(base) else os.path.join(os.path.dirname(path), base))
        self._merge(data)

    def merge_from_list(self, opts):
        if self.is_frozen():
            raise AttributeError("immutable CfgNode")
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent config key: {key}")
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {key}")
            if isinstance(val, str):
                try:
                    val = literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = _coerce(node[parts[-1]], val)


def _coerce(old: Any, new: Any) -> Any:
    if isinstance(new, str) and isinstance(old, (tuple, list)):  # yacs decodes "(210000, 250000)" (step1_bbox.yaml:37) with literal_eval
        try:
            new = literal_eval(new)
        except (ValueError, SyntaxError):
            pass
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    return new


CN = CfgNode


def get_cfg() -> CfgNode:
    """detectron2 defaults (v0.6) for the keys the PlaneRCNN detection path reads."""
    _C = CN()
    _C.VERSION = 2
    _C.MODEL = CN()
    _C.MODEL.LOAD_PROPOSALS = False
    _C.MODEL.MASK_ON = False
    _C.MODEL.KEYPOINT_ON = False
    _C.MODEL.DEVICE = "cuda"
    _C.MODEL.META_ARCHITECTURE = "GeneralizedRCNN"
    _C.MODEL.WEIGHTS = ""
    _C.MODEL.PIXEL_MEAN = [103.530, 116.280, 123.675]
    _C.MODEL.PIXEL_STD = [1.0, 1.0, 1.0]
    _C.INPUT = CN()
    _C.INPUT.FORMAT = "BGR"
    _C.INPUT.MIN_SIZE_TEST = 800
    _C.INPUT.MAX_SIZE_TEST = 1333
    _C.INPUT.MASK_FORMAT = "polygon"
    _C.DATASETS = CN()
    _C.DATASETS.TRAIN = ()
    _C.DATASETS.TEST = ()
    _C.DATALOADER = CN()
    _C.DATALOADER.NUM_WORKERS = 4
    _C.DATALOADER.ASPECT_RATIO_GROUPING = True
    _C.MODEL.BACKBONE = CN()
    _C.MODEL.BACKBONE.NAME = "build_resnet_backbone"
    _C.MODEL.BACKBONE.FREEZE_AT = 2
    _C.MODEL.FPN = CN()
    _C.MODEL.FPN.IN_FEATURES = []
    _C.MODEL.FPN.OUT_CHANNELS = 256
    _C.MODEL.FPN.NORM = ""
    _C.MODEL.FPN.FUSE_TYPE = "sum"
    _C.MODEL.PROPOSAL_GENERATOR = CN()
    _C.MODEL.PROPOSAL_GENERATOR.NAME = "RPN"
    _C.MODEL.PROPOSAL_GENERATOR.MIN_SIZE = 0
    _C.MODEL.ANCHOR_GENERATOR = CN()
    _C.MODEL.ANCHOR_GENERATOR.NAME = "DefaultAnchorGenerator"
    _C.MODEL.ANCHOR_GENERATOR.SIZES = [[32, 64, 128, 256, 512]]
    _C.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS = [[0.5, 1.0, 2.0]]
    _C.MODEL.ANCHOR_GENERATOR.ANGLES = [[-90, 0, 90]]
    _C.MODEL.ANCHOR_GENERATOR.OFFSET = 0.0
    _C.MODEL.RPN = CN()
    _C.MODEL.RPN.HEAD_NAME = "StandardRPNHead"
    _C.MODEL.RPN.IN_FEATURES = ["res4"]
    _C.MODEL.RPN.BOUNDARY_THRESH = -1
    _C.MODEL.RPN.IOU_THRESHOLDS = [0.3, 0.7]
    _C.MODEL.RPN.IOU_LABELS = [0, -1, 1]
    _C.MODEL.RPN.BATCH_SIZE_PER_IMAGE = 256
    _C.MODEL.RPN.POSITIVE_FRACTION = 0.5
    _C.MODEL.RPN.BBOX_REG_LOSS_TYPE = "smooth_l1"
    _C.MODEL.RPN.BBOX_REG_LOSS_WEIGHT = 1.0
    _C.MODEL.RPN.BBOX_REG_WEIGHTS = (1.0, 1.0, 1.0, 1.0)
    _C.MODEL.RPN.SMOOTH_L1_BETA = 0.0
    _C.MODEL.RPN.LOSS_WEIGHT = 1.0
    _C.MODEL.RPN.PRE_NMS_TOPK_TRAIN = 12000
    _C.MODEL.RPN.PRE_NMS_TOPK_TEST = 6000
    _C.MODEL.RPN.POST_NMS_TOPK_TRAIN = 2000
    _C.MODEL.RPN.POST_NMS_TOPK_TEST = 1000
    _C.MODEL.RPN.NMS_THRESH = 0.7
    _C.MODEL.ROI_HEADS = CN()
    _C.MODEL.ROI_HEADS.NAME = "Res5ROIHeads"
    _C.MODEL.ROI_HEADS.NUM_CLASSES = 80
    _C.MODEL.ROI_HEADS.IN_FEATURES = ["res4"]
    _C.MODEL.ROI_HEADS.IOU_THRESHOLDS = [0.5]
    _C.MODEL.ROI_HEADS.IOU_LABELS = [0, 1]
    _C.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 512
    _C.MODEL.ROI_HEADS.POSITIVE_FRACTION = 0.25
    _C.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.05
    _C.MODEL.ROI_HEADS.NMS_THRESH_TEST = 0.5
    _C.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT = True
    _C.MODEL.ROI_BOX_HEAD = CN()
    _C.MODEL.ROI_BOX_HEAD.NAME = ""
    _C.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE = "smooth_l1"
    _C.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT = 1.0
    _C.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS = (10.0, 10.0, 5.0, 5.0)
    _C.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA = 0.0
    _C.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION = 14
    _C.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO = 0
    _C.MODEL.ROI_BOX_HEAD.POOLER_TYPE = "ROIAlignV2"
    _C.MODEL.ROI_BOX_HEAD.NUM_FC = 0
    _C.MODEL.ROI_BOX_HEAD.FC_DIM = 1024
    _C.MODEL.ROI_BOX_HEAD.NUM_CONV = 0
    _C.MODEL.ROI_BOX_HEAD.CONV_DIM = 256
    _C.MODEL.ROI_BOX_HEAD.NORM = ""
    _C.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = False
    _C.MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES = False
    _C.MODEL.ROI_MASK_HEAD = CN()
    _C.MODEL.ROI_MASK_HEAD.NAME = "MaskRCNNConvUpsampleHead"
    _C.MODEL.ROI_MASK_HEAD.POOLER_RESOLUTION = 14
    _C.MODEL.ROI_MASK_HEAD.POOLER_SAMPLING_RATIO = 0
    _C.MODEL.ROI_MASK_HEAD.NUM_CONV = 0
    _C.MODEL.ROI_MASK_HEAD.CONV_DIM = 256
    _C.MODEL.ROI_MASK_HEAD.NORM = ""
    _C.MODEL.ROI_MASK_HEAD.CLS_AGNOSTIC_MASK = False
    _C.MODEL.ROI_MASK_HEAD.POOLER_TYPE = "ROIAlignV2"
    _C.MODEL.RESNETS = CN()
    _C.MODEL.RESNETS.DEPTH = 50
    _C.MODEL.RESNETS.OUT_FEATURES = ["res4"]
    _C.MODEL.RESNETS.NUM_GROUPS = 1
    _C.MODEL.RESNETS.NORM = "FrozenBN"
    _C.MODEL.RESNETS.WIDTH_PER_GROUP = 64
    _C.MODEL.RESNETS.STRIDE_IN_1X1 = True
    _C.MODEL.RESNETS.RES5_DILATION = 1
    _C.MODEL.RESNETS.RES2_OUT_CHANNELS = 256
    _C.MODEL.RESNETS.STEM_OUT_CHANNELS = 64
    _C.MODEL.RESNETS.DEFORM_ON_PER_STAGE = [False, False, False, False]
    _C.MODEL.RESNETS.DEFORM_MODULATED = False
    _C.MODEL.RESNETS.DEFORM_NUM_GROUPS = 1
    _C.SOLVER = CN()
    _C.SOLVER.IMS_PER_BATCH = 16
    _C.SOLVER.BASE_LR = 0.001
    # detectron2 defaults of the keys the training step reads (tools/train_net.py -> DefaultTrainer.build_optimizer / build_lr_scheduler)
    _C.SOLVER.LR_SCHEDULER_NAME = "WarmupMultiStepLR"
    _C.SOLVER.MAX_ITER = 40000
    _C.SOLVER.MOMENTUM = 0.9
    _C.SOLVER.NESTEROV = False
    _C.SOLVER.WEIGHT_DECAY = 0.0001
    _C.SOLVER.GAMMA = 0.1
    _C.SOLVER.STEPS = (30000,)
    _C.SOLVER.WARMUP_FACTOR = 0.001
    _C.SOLVER.WARMUP_ITERS = 1000
    _C.SOLVER.WARMUP_METHOD = "linear"
    _C.SOLVER.CHECKPOINT_PERIOD = 5000
    _C.TEST = CN()
    _C.TEST.DETECTIONS_PER_IMAGE = 100
    _C.TEST.EVAL_PERIOD = 0
    _C.OUTPUT_DIR = "./output"
    _C.SEED = -1
    _C.CUDNN_BENCHMARK = False
    _C.VIS_PERIOD = 0
    _C.GLOBAL = CN()
    _C.GLOBAL.HACK = 1.0
    return _C


def get_planercnn_cfg_defaults(cfg: CfgNode) -> CfgNode:
    """Same keys and defaults as the reference's pkg/config/config.py:7-85."""
    cfg.MODEL.FREEZE = []
    cfg.MODEL.PLANE_ON = True
    cfg.MODEL.DEPTH_ON = False
    cfg.MODEL.REFINE_ON = False
    cfg.MODEL.AXIS_ON = False
    cfg.MODEL.VIS_MINIBATCH = False
    cfg.INPUT.IMG_HEIGHT = 480
    cfg.INPUT.IMG_WIDTH = 640
    cfg.INPUT.IMG_ROOT = "/z/syqian/articulation_data"
    cfg.DATALOADER.ASPECT_RATIO_GROUPING = False
    cfg.TEST.SAVE_VIS = False
    cfg.TEST.EVAL_GT_BOX = False
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.7
    for name in ("ROI_PLANE_HEAD", "ROI_AXIS_HEAD"):
        n = CN()
        n.NAME = "PlaneRCNNConvFCHead"
        n.POOLER_RESOLUTION = 14
        n.POOLER_SAMPLING_RATIO = 0
        n.POOLER_TYPE = "ROIAlign"
        n.EMBEDDING_DIM = 128
        n.NUM_CONV = 4
        n.CONV_DIM = 256
        n.NUM_FC = 1
        n.FC_DIM = 1024
        n.NORM = ""
        n.LOSS_WEIGHT = 1.0
        n.PARAM_DIM = 3
        cfg.MODEL[name] = n
    cfg.MODEL.ROI_PLANE_HEAD.NORMAL_ONLY = True
    cfg.MODEL.ROI_AXIS_HEAD.SMOOTH_L1_BETA = 0.0
    cfg.MODEL.DEPTH_HEAD = CN()
    cfg.MODEL.DEPTH_HEAD.NAME = "PlaneRCNNDepthHead"
    cfg.MODEL.DEPTH_HEAD.LOSS_WEIGHT = 1.0
    cfg.MODEL.REFINE_HEAD = CN()
    cfg.MODEL.REFINE_HEAD.NAME = "PlaneRCNNRefineHead"
    cfg.MODEL.REFINE_HEAD.LOSS_WEIGHT = 1.0
    cfg.MODEL.ROI_MASK_HEAD.MASK_THRESHOLD = 0.5
    cfg.MODEL.ROI_MASK_HEAD.NMS = False
    return cfg
