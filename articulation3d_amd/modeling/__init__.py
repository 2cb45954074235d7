from .meta_arch import PlaneRCNN, BatchedOutput, build_model  # noqa: F401
from .backbone import build_backbone, build_resnet_fpn_backbone  # noqa: F401
from .rpn import RPN, StandardRPNHead, DefaultAnchorGenerator, build_proposal_generator  # noqa: F401
from .roi_heads.roi_heads import PlaneRCNNROIHeads, build_roi_heads  # noqa: F401
from .depth_head import PlaneRCNNDepthHead, build_depth_head  # noqa: F401
from .postprocessing import detector_postprocess, paste_masks_in_image  # noqa: F401
