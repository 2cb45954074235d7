"""ctypes binding of liba3d_hip.so (include/a3d.h).

This is the only place the product touches the kernel library.  There is NO fallback: if the library
is missing or a call returns a non-zero code, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "liba3d_hip.so")

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2

_ERR = {-1: "A3D_ERR_ARG", -2: "A3D_ERR_LAUNCH", -3: "A3D_ERR_UNSUPPORTED"}

fptr = C.c_void_p  # device pointers travel as integers


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", fptr), ("x2", fptr), ("w", fptr), ("scale", fptr), ("shift", fptr), ("res", fptr), ("y", fptr),
        ("workspace", fptr),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cin2", C.c_int),
        ("Ho", C.c_int), ("Wo", C.c_int), ("Cout", C.c_int),
        ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("Kpad", C.c_int), ("ups", C.c_int), ("act", C.c_int), ("res_ups", C.c_int), ("pixshuf", C.c_int),
        ("stem", C.c_int), ("splitk", C.c_int),
        ("m_dev", fptr),
        ("tune", C.c_int), ("phase", C.c_int),
        ("w_wino", fptr),
        ("gate", fptr),
        ("precision", C.c_int),
        ("w_wino_x3", fptr),
        ("w_wino_cm", fptr),
        ("w_x3", fptr),
        ("in_amax", fptr),
        ("in_amax2", fptr),
        ("y_amax", fptr),
        ("w_scale", C.c_float),
        ("wino_m", fptr),
        ("io_bf16", C.c_int),
        ("x_h2", fptr),
        ("x2_h2", fptr),
        ("dot_w", fptr),
        ("dot_y", fptr),
        ("wino_t_off", C.c_int), ("wino_t_total", C.c_int),
        ("w_bf16", fptr),
    ]


class RpnDesc(C.Structure):
    _fields_ = [
        ("head", fptr * 5),
        ("Hf", C.c_int * 5), ("Wf", C.c_int * 5), ("stride", C.c_int * 5),
        ("cell_anchors", ((C.c_float * 4) * 3) * 5),
        ("B", C.c_int), ("L", C.c_int), ("A", C.c_int), ("CH", C.c_int),
        ("img_h", C.c_int), ("img_w", C.c_int),
        ("pre_topk", C.c_int), ("post_topk", C.c_int),
        ("nms_thresh", C.c_float), ("min_size", C.c_float),
        ("weights", C.c_float * 4),
        ("scale_clamp", C.c_float),
        ("workspace", fptr),
        ("out_boxes", fptr), ("out_scores", fptr), ("out_level", fptr), ("out_pos", fptr), ("out_count", fptr),
    ]


class BoxDetDesc(C.Structure):
    _fields_ = [
        ("pred", fptr), ("prop_boxes", fptr), ("prop_count", fptr),
        ("B", C.c_int), ("R", C.c_int), ("C", C.c_int), ("CH", C.c_int),
        ("img_h", C.c_int), ("img_w", C.c_int),
        ("score_thresh", C.c_float), ("nms_thresh", C.c_float),
        ("topk", C.c_int),
        ("weights", C.c_float * 4),
        ("scale_clamp", C.c_float),
        ("workspace", fptr),
        ("out_boxes", fptr), ("out_scores", fptr), ("out_classes", fptr), ("out_pos", fptr), ("out_count", fptr),
    ]


class RoiAlignDesc(C.Structure):
    _fields_ = [
        ("feat", fptr * 4),
        ("Hf", C.c_int * 4), ("Wf", C.c_int * 4),
        ("scale", C.c_float * 4),
        ("L", C.c_int), ("C", C.c_int),
        ("boxes", fptr), ("count", fptr), ("row_offset", fptr),
        ("B", C.c_int), ("R", C.c_int),
        ("P", C.c_int), ("sampling_ratio", C.c_int), ("aligned", C.c_int),
        ("out", fptr), ("out_level", fptr), ("order_ws", fptr),
        ("out_amax", fptr), ("level_amax", fptr * 4), ("window_count", fptr),
        ("out_h2", fptr),
        ("serial", C.c_int),
    ]


class PasteDesc(C.Structure):
    _fields_ = [
        ("boxes", fptr), ("scores", fptr), ("count", fptr), ("row_offset", fptr), ("mask_prob", fptr),
        ("normals", fptr), ("depth", fptr),
        ("B", C.c_int), ("R", C.c_int), ("MS", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("post_score_thresh", C.c_float), ("mask_thresh", C.c_float),
        ("focal", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("clip_boxes", C.c_int),
        ("masks", fptr), ("planes", fptr), ("area", fptr), ("keep", fptr), ("out_boxes", fptr),
    ]


class PackDesc(C.Structure):
    _fields_ = [
        ("boxes", fptr), ("scores", fptr), ("classes", fptr), ("count", fptr), ("row_offset", fptr), ("keep", fptr),
        ("planes", fptr), ("rot_axis", fptr), ("tran_axis", fptr), ("mask_prob", fptr),
        ("B", C.c_int), ("R", C.c_int), ("MS", C.c_int),
        ("records", fptr), ("rec_count", fptr),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("x", fptr), ("dy", fptr), ("scale", fptr), ("dw", fptr), ("workspace", fptr),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int), ("Cout", C.c_int),
        ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("splitk", C.c_int), ("accumulate", C.c_int), ("precision", C.c_int), ("io_bf16", C.c_int), ("defer_reduce", C.c_int),
    ]


class TransposeItem(C.Structure):
    _fields_ = [("w", fptr), ("scale", fptr), ("wt", fptr), ("Cout", C.c_int), ("KH", C.c_int), ("KW", C.c_int), ("Cin", C.c_int),
                ("block0", C.c_int), ("pad_", C.c_int)]


class RoiAlignBwdDesc(C.Structure):
    _fields_ = [
        ("dfeat", fptr * 4),
        ("Hf", C.c_int * 4), ("Wf", C.c_int * 4),
        ("scale", C.c_float * 4),
        ("L", C.c_int), ("C", C.c_int),
        ("boxes", fptr), ("count", fptr), ("row_offset", fptr),
        ("B", C.c_int), ("R", C.c_int), ("P", C.c_int), ("sampling_ratio", C.c_int), ("aligned", C.c_int),
        ("dout", fptr),
    ]


class MatchDesc(C.Structure):
    _fields_ = [
        ("boxes", fptr), ("box_count", fptr), ("gt_boxes", fptr), ("gt_count", fptr),
        ("B", C.c_int), ("N", C.c_int), ("Gmax", C.c_int), ("box_batch_stride", C.c_int),
        ("thresholds", C.c_float * 2), ("labels", C.c_int * 3), ("n_thresholds", C.c_int), ("allow_low_quality", C.c_int),
        ("gt_best", fptr), ("matched_idx", fptr), ("label", fptr), ("matched_iou", fptr),
    ]


class RpnLossDesc(C.Structure):
    _fields_ = [
        ("head", fptr * 5), ("dhead", fptr * 5),
        ("Hf", C.c_int * 5), ("Wf", C.c_int * 5), ("stride", C.c_int * 5),
        ("cell_anchors", ((C.c_float * 4) * 3) * 5),
        ("B", C.c_int), ("L", C.c_int), ("A", C.c_int), ("CH", C.c_int), ("Atotal", C.c_int), ("Gmax", C.c_int),
        ("labels", fptr), ("matched_idx", fptr), ("gt_boxes", fptr),
        ("weights", C.c_float * 4), ("normalizer", C.c_float),
        ("workspace", fptr), ("loss", fptr),
    ]


class BoxLossDesc(C.Structure):
    _fields_ = [
        ("pred", fptr), ("dpred", fptr), ("gt_classes", fptr), ("boxes", fptr), ("gt_boxes", fptr),
        ("M", C.c_int), ("num_classes", C.c_int), ("pitch", C.c_int),
        ("weights", C.c_float * 4),
        ("workspace", fptr), ("loss", fptr),
        ("count", fptr), ("R", C.c_int),
    ]


class RoiSampleDesc(C.Structure):
    _fields_ = [
        ("boxes", fptr), ("box_count", fptr), ("gt_boxes", fptr), ("gt_classes", fptr), ("gt_count", fptr),
        ("matched_idx", fptr), ("match_label", fptr),
        ("B", C.c_int), ("N", C.c_int), ("Gmax", C.c_int), ("num_classes", C.c_int), ("num", C.c_int), ("max_fg", C.c_int),
        ("seed", C.c_ulonglong),
        ("out_boxes", fptr), ("out_gt_boxes", fptr), ("out_classes", fptr), ("out_index", fptr), ("out_count", fptr),
    ]


class SweepDesc(C.Structure):
    _fields_ = [
        ("mask", fptr), ("H", C.c_int), ("W", C.c_int),
        ("normal", C.c_float * 3), ("offset", C.c_float),
        ("focal", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("pivot", C.c_float * 3),
        ("xforms", fptr), ("A", C.c_int),
        ("out_bits", fptr),
    ]


# name -> (restype, argtypes); every symbol include/a3d.h declares
SIGNATURES = {
    "a3d_version": (C.c_int, []),
    "a3d_struct_size": (C.c_size_t, [C.c_int]),
    "a3d_preprocess_u8hwc": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), fptr]),
    "a3d_preprocess_f32chw": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), fptr]),
    "a3d_preprocess_resize_u8": (C.c_int, [fptr, fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                           C.POINTER(C.c_float), fptr]),
    "a3d_conv_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "a3d_wino_m_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "a3d_wino_gemm_levels": (C.c_int, [C.POINTER(ConvDesc), C.c_int, fptr]),
    "a3d_split_bf16x3": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_split_bf16x3_chunk": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_split_f16x2_chunk": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, fptr]),
    "a3d_absmax_rows": (C.c_int, [fptr, fptr, C.c_int, C.c_size_t, fptr]),
    "a3d_presplit_f16x2": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int, C.c_size_t, fptr]),
    "a3d_mask_rle": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr, fptr, fptr, fptr]),
    "a3d_roi_amax": (C.c_int, [C.POINTER(fptr), C.c_int, fptr, fptr, C.c_int, C.c_int, fptr, fptr]),
    "a3d_conv2d_nhwc_f32": (C.c_int, [C.POINTER(ConvDesc), fptr]),
    "a3d_conv_b2b": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), fptr]),
    "a3d_stem_conv_pool": (C.c_int, [C.POINTER(ConvDesc), fptr]),
    "a3d_tapsum9": (C.c_int, [fptr, C.c_float, fptr, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_wino_input_transform": (C.c_int, [C.POINTER(ConvDesc), fptr]),
    "a3d_wino_gemm": (C.c_int, [C.POINTER(ConvDesc), fptr]),
    "a3d_last_conv_variant": (C.c_char_p, []),
    "a3d_maxpool3x3s2_nhwc": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_subsample2_nhwc": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_resize_bilinear_nhwc": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_conv3x3_to1_nhwc": (C.c_int, [fptr, fptr, C.c_float, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_group_buffers_bytes": (C.c_size_t, [C.c_int]),
    "a3d_rpn_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "a3d_rpn_proposals": (C.c_int, [C.POINTER(RpnDesc), fptr]),
    "a3d_box_detections": (C.c_int, [C.POINTER(BoxDetDesc), fptr]),
    "a3d_group_nms": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int, C.c_float, fptr]),
    "a3d_roi_align_fpn": (C.c_int, [C.POINTER(RoiAlignDesc), fptr]),
    "a3d_count_offsets": (C.c_int, [fptr, fptr, C.c_int, C.c_int, fptr]),
    "a3d_linear_small": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_paste_lsq": (C.c_int, [C.POINTER(PasteDesc), fptr]),
    "a3d_plane_offset_dense": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, fptr]),
    "a3d_plane_offset_dense_u8": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, fptr]),
    "a3d_detections_pack": (C.c_int, [C.POINTER(PackDesc), fptr]),
    "a3d_record_floats": (C.c_int, [C.c_int]),
    # training step
    "a3d_wgrad_workspace_bytes": (C.c_size_t, [C.POINTER(WgradDesc)]),
    "a3d_conv_wgrad_nhwc_f32": (C.c_int, [C.POINTER(WgradDesc), fptr]),
    "a3d_wgrad_tiles": (C.c_int, [C.POINTER(WgradDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "a3d_weight_transpose": (C.c_int, [fptr, fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_weight_transpose_batch": (C.c_int, [fptr, C.c_int, C.c_int, fptr]),
    "a3d_wgrad_reduce_batch": (C.c_int, [fptr, C.c_int, fptr]),
    "a3d_wino_weight_transform": (C.c_int, [fptr, fptr, C.c_int, C.c_int, fptr]),
    "a3d_zero_insert2_nhwc": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_sumpool2_add_nhwc": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_colsum_workspace_bytes": (C.c_size_t, [C.c_int]),
    "a3d_colsum": (C.c_int, [fptr, fptr, fptr, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_colsum_bf16": (C.c_int, [fptr, fptr, fptr, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_roi_align_fpn_backward": (C.c_int, [C.POINTER(RoiAlignBwdDesc), fptr]),
    "a3d_roi_align_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(RoiAlignBwdDesc)]),
    "a3d_roi_align_fpn_backward_gather": (C.c_int, [C.POINTER(RoiAlignBwdDesc), fptr, fptr]),
    "a3d_match_boxes": (C.c_int, [C.POINTER(MatchDesc), fptr]),
    "a3d_loss_workspace_bytes": (C.c_size_t, []),
    "a3d_rpn_loss": (C.c_int, [C.POINTER(RpnLossDesc), fptr]),
    "a3d_box_loss": (C.c_int, [C.POINTER(BoxLossDesc), fptr]),
    "a3d_sample_labels": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ulonglong, fptr]),
    "a3d_append_gt_boxes": (C.c_int, [fptr, fptr, fptr, fptr, fptr, fptr, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_sample_rois": (C.c_int, [C.POINTER(RoiSampleDesc), fptr]),
    "a3d_masks_pack_bits": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_masks_unpack_bits": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_project_hypotheses": (C.c_int, [C.POINTER(SweepDesc), fptr]),
    "a3d_mask_iou_matrix": (C.c_int, [fptr, fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "a3d_sgd_momentum": (C.c_int, [fptr, fptr, fptr, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, fptr]),
    "a3d_sgd_momentum_bf16g": (C.c_int, [fptr, fptr, fptr, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, fptr]),
    "a3d_f32_to_bf16_scaled": (C.c_int, [fptr, fptr, C.c_size_t, C.c_float, fptr]),
    "a3d_bf16_to_f32": (C.c_int, [fptr, fptr, C.c_size_t, fptr]),
}

STRUCT_IDS = {0: ConvDesc, 1: RpnDesc, 2: BoxDetDesc, 3: RoiAlignDesc, 4: PasteDesc, 5: PackDesc, 6: WgradDesc,
              7: RoiAlignBwdDesc, 8: MatchDesc, 9: RpnLossDesc, 10: BoxLossDesc, 11: RoiSampleDesc, 12: SweepDesc, 13: TransposeItem}

_lib = None


def lib():
    """The loaded kernel library.  Raises if it has not been built (python -m articulation3d_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP kernel library has not been built. "
                "Run `python -m articulation3d_amd.build` (or __graft_entry__.build()). There is no CPU fallback."
            )
        # torch ships its own ROCm runtime (torch/lib/libamdhip64.so).  It must be the HIP runtime of the process:
        # if this library were loaded first it would pull in the system libamdhip64 and the two runtimes would not
        # share streams / code objects (every launch then fails).  Import torch before dlopen.
        import torch  # noqa: F401

        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        for sid, cls in STRUCT_IDS.items():  # a layout mismatch would silently shift every field
            if handle.a3d_struct_size(sid) != C.sizeof(cls):
                raise RuntimeError(f"descriptor layout mismatch for {cls.__name__}: C {handle.a3d_struct_size(sid)} != ctypes {C.sizeof(cls)}")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed with {_ERR.get(rc, rc)}")
