"""The pieces of detectron2's training engine that the reference's tools/train_net.py reaches (:23-45 Trainer, :84-104 main ->
DefaultTrainer -> SimpleTrainer.run_step), mirrored over the hand-written training step (articulation3d_amd/training.py).

    model = build_model(cfg).train()
    optimizer = build_optimizer(cfg, model)
    for data in loader:                      # SimpleTrainer.run_step
        loss_dict = model(data)              # forward + backward launches; gradients in the trainer's flat buffer
        losses = sum(loss_dict.values())
        optimizer.zero_grad()
        losses.backward()
        optimizer.step()                     # ONE RCCL all-reduce of the flat gradient buffer + ONE fused SGD launch

Data-parallel training is one process per GPU (torch.distributed, backend nccl = RCCL): no DistributedDataParallel wrapper is
needed, the optimiser's step all-reduces the flat gradient buffer (parallel.allreduce_gradients)."""
from __future__ import annotations

from .training import SolverCfg, lr_at


def solver_from_cfg(cfg) -> SolverCfg:
    """cfg.SOLVER / cfg.MODEL.* keys of config/step1_bbox.yaml -> the trainer's SolverCfg."""
    s, m = cfg.SOLVER, cfg.MODEL
    return SolverCfg(
        rpn_batch_per_image=m.RPN.BATCH_SIZE_PER_IMAGE, rpn_positive_fraction=m.RPN.POSITIVE_FRACTION,
        rpn_iou_thresholds=tuple(m.RPN.IOU_THRESHOLDS), rpn_pre_topk_train=m.RPN.PRE_NMS_TOPK_TRAIN,
        rpn_post_topk_train=m.RPN.POST_NMS_TOPK_TRAIN, rpn_nms_thresh=m.RPN.NMS_THRESH,
        roi_batch_per_image=m.ROI_HEADS.BATCH_SIZE_PER_IMAGE, roi_positive_fraction=m.ROI_HEADS.POSITIVE_FRACTION,
        roi_iou_threshold=float(m.ROI_HEADS.IOU_THRESHOLDS[0]), num_classes=m.ROI_HEADS.NUM_CLASSES,
        rpn_weights=tuple(m.RPN.BBOX_REG_WEIGHTS), box_weights=tuple(m.ROI_BOX_HEAD.BBOX_REG_WEIGHTS),
        base_lr=s.BASE_LR, momentum=s.MOMENTUM, weight_decay=s.WEIGHT_DECAY, warmup_iters=s.WARMUP_ITERS,
        warmup_factor=s.WARMUP_FACTOR, steps=tuple(s.STEPS), gamma=s.GAMMA)


class FlatSGD:
    """torch.optim.SGD + WarmupMultiStepLR of detectron2's build_optimizer / build_lr_scheduler, over the trainer's flat buffers."""

    def __init__(self, trainer):
        self.trainer = trainer

    @property
    def param_groups(self):
        return [{"lr": lr_at(self.trainer.iter, self.trainer.s), "momentum": self.trainer.s.momentum,
                 "weight_decay": self.trainer.s.weight_decay}]

    def zero_grad(self, set_to_none: bool = True):
        pass  # every step overwrites the flat gradient buffer

    def step(self):
        self.trainer.optimizer_step()

    def state_dict(self):
        return {"iteration": self.trainer.iter, "momentum": self.trainer.momentum.clone()}

    def load_state_dict(self, sd):
        self.trainer.iter = int(sd["iteration"])
        self.trainer.momentum.copy_(sd["momentum"])


def build_optimizer(cfg, model, precision: str = "bf16x3") -> FlatSGD:
    """detectron2.solver.build_optimizer(cfg, model) for this package's model: SGD momentum / weight decay / warm-up multi-step
    schedule from cfg.SOLVER, bound to the model's trainer.  precision "bf16" = the reference's autocast arithmetic."""
    return FlatSGD(model.trainer(solver_from_cfg(cfg), precision=precision))
