"""Minimal `Boxes` / `Instances` / `ImageList` / `ShapeSpec` containers with the detectron2 surface the
reference's hot path touches (pkg/modeling/meta_arch/planercnn.py:6-9,195; roi_heads.py:9,121-125;
postprocessing.py:47-60; utils/arti_vis.py:152-194).  Plain Python + tensor indexing: no arithmetic
kernels live here."""
from __future__ import annotations

import itertools
from collections import namedtuple
from typing import Any, Dict, List, Tuple, Union

import numpy as np
import torch


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    def __new__(cls, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)


def to_host(t: torch.Tensor) -> torch.Tensor:
    """Device -> host copy through PINNED host memory (torch's caching host allocator keeps the pages locked and re-uses
    them).  A plain `.cpu()` lands in pageable memory that the HIP runtime has to register with the GPU for the DMA; when
    that memory is later returned to the OS the driver must tear the registration down, which evicts and restores the
    process's GPU queues -- measured as 40-70 ms stalls of the NEXT frame every few frames of the reference-style loop."""
    if not t.is_cuda:
        return t
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t)
    return h


def _is_cpu(device) -> bool:
    return str(device) == "cpu" or (isinstance(device, torch.device) and device.type == "cpu")


class Boxes:
    """N x 4 xyxy float32 boxes."""

    def __init__(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(np.asarray(tensor), dtype=torch.float32)
        tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4))
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, device):
        return Boxes(to_host(self.tensor) if _is_cpu(device) else self.tensor.to(device=device))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size):
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

    def nonempty(self, threshold: float = 0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def get_centers(self):
        """(N, 2) box centres (x, y) -- detectron2 Boxes.get_centers, used by the optimiser (opt_utils.py:405)."""
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    def scale(self, scale_x: float, scale_y: float):
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    @property
    def device(self):
        return self.tensor.device

    @classmethod
    def cat(cls, boxes_list):
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    def __iter__(self):
        yield from self.tensor


class Instances:
    """Per-image bag of equally long fields, detectron2-style (set by attribute)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name, value):
        data_len = len(value)
        if len(self._fields):
            assert len(self) == data_len, f"Adding a field of length {data_len} to a Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, *args, **kwargs):
        ret = Instances(self._image_size)
        to_cpu = len(args) == 1 and not kwargs and _is_cpu(args[0])
        for k, v in self._fields.items():
            if to_cpu and isinstance(v, torch.Tensor):
                v = to_host(v)
            elif hasattr(v, "to"):
                v = v.to(*args, **kwargs)
            ret.set(k, v)
        return ret

    def __getitem__(self, item):
        if type(item) == int:
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    @staticmethod
    def cat(instance_lists: List["Instances"]) -> "Instances":
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        image_size = instance_lists[0].image_size
        ret = Instances(image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = list(itertools.chain(*values))
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError(f"Unsupported type {type(v0)} for concatenation")
            ret.set(k, values)
        return ret

    def __str__(self):
        s = self.__class__.__name__ + "("
        s += f"num_instances={len(self) if len(self._fields) else 0}, "
        s += f"image_height={self._image_size[0]}, image_width={self._image_size[1]}, "
        s += "fields=[{}])".format(", ".join(f"{k}: {v}" for k, v in self._fields.items()))
        return s

    __repr__ = __str__


class ImageList:
    """Batched, padded images + their un-padded sizes."""

    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors: List[torch.Tensor], size_divisibility: int = 0, pad_value: float = 0.0) -> "ImageList":
        assert len(tensors) > 0
        image_sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in tensors]
        max_h = max(s[0] for s in image_sizes)
        max_w = max(s[1] for s in image_sizes)
        if size_divisibility > 1:
            st = size_divisibility
            max_h = (max_h + st - 1) // st * st
            max_w = (max_w + st - 1) // st * st
        if all(s == (max_h, max_w) for s in image_sizes):
            batched = torch.stack(tensors)
        else:
            batched = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (max_h, max_w), pad_value)
            for img, pad_img in zip(tensors, batched):
                pad_img[..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batched.contiguous(), image_sizes)


def pairwise_iou(boxes1: Boxes, boxes2: Boxes) -> torch.Tensor:
    """Host-side IoU matrix used by the temporal tracker (pkg/utils/opt_utils.py:1156-1208)."""
    a, b = boxes1.tensor, boxes2.tensor
    area1, area2 = boxes1.area(), boxes2.area()
    wh = (torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])).clamp(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (area1[:, None] + area2 - inter), torch.zeros(1, dtype=inter.dtype, device=inter.device))
