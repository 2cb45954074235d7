import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import planercnn_oracle as O

    return O


@pytest.fixture(scope="session")
def oracle_params(oracle):
    """Seeded random-init + BN-calibrated weights (shared BY VALUE with the HIP path)."""
    return oracle.init_params(2020)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def make_cfg(score_thresh=0.7, device="cuda"):
    from articulation3d_amd.config import get_cfg, get_planercnn_cfg_defaults

    cfg = get_cfg()
    get_planercnn_cfg_defaults(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "planercnn_inference.yaml"))
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = score_thresh
    cfg.MODEL.DEVICE = device
    return cfg


@pytest.fixture(scope="session")
def hip_model(oracle_params):
    """The product model on cuda:0 carrying the oracle's weights; score threshold set per test."""
    import torch

    from articulation3d_amd.modeling import build_model

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    model = build_model(make_cfg(0.0)).eval()
    missing, unexpected = model.load_state_dict(oracle_params, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing)
    return model
