from .config import CfgNode, CN, get_cfg, get_planercnn_cfg_defaults  # noqa: F401
