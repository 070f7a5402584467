"""ctypes binding of libumpr_hip.so (C ABI in include/umpr_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is raised
(the product path never routes through oracle/ or through torch arithmetic).
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libumpr_hip.so")

_T = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_long, "z": ctypes.c_size_t, "f": ctypes.c_float,
      "d": ctypes.c_double, "u": ctypes.c_uint64}

# name -> (argument codes, return code).  p pointer, i int, l long, z size_t, f float, d double, u uint64.
SIGNATURES = {
    "umpr_version": ("", "s"),
    "umpr_last_error": ("", "s"),
    "umpr_device_info": ("ipipp", "i"),
    "umpr_gemm_f32": ("pliplipliiipiiifpzp", "i"),
    "umpr_embed_gru_bidir_ws_bytes": ("iii", "z"),
    "umpr_embed_gru_bidir_fwd": ("ppipppppppppppiipppzp", "i"),
    "umpr_embed_gru_bidir_bwd": ("ppipppppiippppppppppppzp", "i"),
    "umpr_embed_gru_bidir_bwd_acc": ("ppipppppiipppppppppppipzp", "i"),
    "umpr_coattention_fwd_ws_bytes": ("ii", "z"),
    "umpr_coattention_fwd": ("pppiipppplplpppppzp", "i"),
    "umpr_coattention_fwd_bf16": ("pppiipppplplpppppzp", "i"),
    "umpr_coattention_bwd_ws_bytes": ("ii", "z"),
    "umpr_coattention_bwd": ("ppppppppppplplppiipppipzp", "i"),
    "umpr_snet_fwd": ("ppppiiiippppplp", "i"),
    "umpr_snet_bwd_ws_bytes": ("iii", "z"),
    "umpr_snet_bwd": ("pppppppplpiiiipppppzp", "i"),
    "umpr_review_merge_fwd": ("ppppipp", "i"),
    "umpr_review_merge_bwd_ws_bytes": ("i", "z"),
    "umpr_review_merge_bwd": ("ppppppipppppzp", "i"),
    "umpr_cnet_head_fwd_ws_bytes": ("iiii", "z"),
    "umpr_cnet_head_fwd": ("pppppfiiiiiipppppppzp", "i"),
    "umpr_cnet_head_bwd_ws_bytes": ("iiiiii", "z"),
    "umpr_cnet_head_bwd": ("pppppppppiiiiiipiipppppzp", "i"),
    "umpr_control_gate_fwd": ("pppppiiippppp", "i"),
    "umpr_control_gate_bwd_ws_bytes": ("i", "z"),
    "umpr_control_gate_bwd": ("ppppppppiiippppppzp", "i"),
    "umpr_vgg16_act_bytes": ("i", "z"),
    "umpr_vgg16_fwd_ws_bytes": ("i", "z"),
    "umpr_vgg16_ws_bytes": ("i", "z"),
    "umpr_vgg16_fwd": ("ppiiiuppppzp", "i"),
    "umpr_vgg16_bwd": ("ppiipppppzp", "i"),
    "umpr_vgg16_pool5_offset": ("i", "z"),
    "umpr_vgg16_set_block_callback": ("pp", "i"),
    "umpr_vgg16_wgrad_stream": ("", "p"),
    "umpr_vgg16_features_fwd": ("ppippzp", "i"),
    "umpr_vgg16_classifier_fwd": ("piiiuppppzp", "i"),
    "umpr_vgg16_classifier_bwd_ws_bytes": ("i", "z"),
    "umpr_vgg16_classifier_bwd": ("piippppppzp", "i"),
    "umpr_vgg16_features_bwd_ws_bytes": ("i", "z"),
    "umpr_vgg16_features_bwd": ("ppippppzp", "i"),
    "umpr_conv3x3_pack_bytes": ("iiiii", "z"),
    "umpr_conv3x3_fwd": ("ppppiiiiiipzp", "i"),
    "umpr_conv3x3_bwd_data": ("ppppiiiiipzp", "i"),
    "umpr_conv3x3_bwd_weight_ws_bytes": ("iiiii", "z"),
    "umpr_conv3x3_bwd_weight": ("ppppiiiiipzp", "i"),
    "umpr_maxpool2_fwd": ("ppliip", "i"),
    "umpr_maxpool2_bwd_relu": ("pppliip", "i"),
    "umpr_bf16_tensor_bytes": ("iiii", "z"),
    "umpr_bf16_from_nchw_f32": ("ppiiiip", "i"),
    "umpr_bf16_to_nchw_f32": ("ppiiiip", "i"),
    "umpr_conv3x3_bf16_ws_bytes": ("iiiii", "z"),
    "umpr_conv3x3_bf16_fwd": ("ppppiiiiiipzp", "i"),
    "umpr_conv3x3_bf16_bwd_data": ("ppppiiiiipzp", "i"),
    "umpr_conv3x3_bf16_bwd_weight": ("ppppiiiiipzp", "i"),
    "umpr_maxpool2_bf16_fwd": ("ppiiiip", "i"),
    "umpr_maxpool2_bf16_bwd_relu": ("pppiiiip", "i"),
    "umpr_vgg16_bf16_act_bytes": ("i", "z"),
    "umpr_vgg16_bf16_fwd_ws_bytes": ("i", "z"),
    "umpr_vgg16_bf16_bwd_ws_bytes": ("i", "z"),
    "umpr_vgg16_bf16_features_fwd": ("ppipppzp", "i"),
    "umpr_vgg16_bf16_features_bwd": ("ppippppzp", "i"),
    "umpr_vgg16_cls_arena_bytes": ("i", "z"),
    "umpr_vgg16_classifier_fwd_compact": ("piiiuppppzp", "i"),
    "umpr_vgg16_classifier_bwd_compact": ("piippppppzp", "i"),
    "umpr_vgg16_classifier_fwd_compact_bf16": ("piiiuppppzp", "i"),
    "umpr_vgg16_classifier_bwd_compact_bf16": ("piippppppzp", "i"),
    "umpr_concat_ids": ("pplpp", "i"),
    "umpr_review_net_arena_bytes": ("iii", "z"),
    "umpr_review_net_ws_bytes": ("iiii", "z"),
    "umpr_review_net_fwd": ("ppipppiiiiiipppzp", "i"),
    "umpr_review_net_bwd": ("ppipppiiiippppzp", "i"),
    "umpr_control_net_arena_bytes": ("iiiiiii", "z"),
    "umpr_control_net_ws_bytes": ("iiiiiiiii", "z"),
    "umpr_control_net_fwd": ("pppipppppiiiiiiiifiippppppzp", "i"),
    "umpr_control_net_bwd": ("pppipppppiiiiiiiiipppppppzp", "i"),
    "umpr_head_fwd": ("pppppppppppppfiiipppppppp", "i"),
    "umpr_head_bwd": ("pppppppppppfiiippppppppppppppppppppp", "i"),
    "umpr_bce_head_fwd": ("plpppiipppzp", "i"),
    "umpr_bce_head_bwd": ("plpppppiiplpppzp", "i"),
    "umpr_adam_step": ("ppppldddddldp", "i"),
    "umpr_sq_err_accumulate": ("pplpp", "i"),
    "umpr_adam_step_dev": ("pppplddd" "pp", "i"),
    "umpr_debug_poison_lds": ("pp", "i"),
    "umpr_set_gemm_bf16": ("i", "i"),
    "umpr_set_conv_inference": ("i", "i"),
    "umpr_set_conv_pool_follows": ("i", "i"),
    "umpr_debug_wino_fix_count": ("", "l"),
    "umpr_profile_enable": ("i", "i"),
    "umpr_profile_reset": ("", "i"),
    "umpr_profile_read": ("ippp", "i"),
}


class UmprHipError(RuntimeError):
    pass


class _Tls(threading.local):
    gemm_b16 = False   # set by umpr_amd.model around the text path's calls in bf16 mode (umpr_set_gemm_bf16); per thread,
                       # like the library's own switch


TLS = _Tls()


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise UmprHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C umpr_amd/csrc`).  There is no CPU fallback.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.fn = {}
        for name, (args, ret) in SIGNATURES.items():
            f = getattr(self.cdll, name)
            f.argtypes = [_T[c] for c in args]
            f.restype = {"i": ctypes.c_int, "z": ctypes.c_size_t, "s": ctypes.c_char_p, "p": ctypes.c_void_p,
                         "l": ctypes.c_long}[ret]
            self.fn[name] = f

    def last_error(self) -> str:
        return self.fn["umpr_last_error"]().decode()

    def call(self, name, *args):
        """Call an int-returning entry point; tensors become device pointers; raises on error."""
        conv = []
        for a in args:
            if isinstance(a, torch.Tensor):
                conv.append(a.data_ptr())
            elif a is None:
                conv.append(None)
            else:
                conv.append(a)
        if TLS.gemm_b16:
            self.fn["umpr_set_gemm_bf16"](1)
            try:
                rc = self.fn[name](*conv)
            finally:
                self.fn["umpr_set_gemm_bf16"](0)
        else:
            rc = self.fn[name](*conv)
        if rc != 0:
            raise UmprHipError(f"{name} failed (rc={rc}): {self.last_error()}")

    def size(self, name, *args) -> int:
        return int(self.fn[name](*args))


_LIB = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


class Workspace:
    """One growing scratch buffer per (device, stream).  Ops on one stream are serialised, so they can share it;
    nothing in it is live between two C-ABI calls.  Two streams never share a buffer (the text path and the VGG path of
    UMPR.forward run concurrently)."""
    _bufs = {}

    @classmethod
    def get(cls, nbytes: int, device) -> torch.Tensor:
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() * 4 < nbytes:
            buf = None
            cls._bufs[key] = None
            n = (int(nbytes * 1.1) + 1023) // 4
            buf = torch.empty(n, dtype=torch.float32, device=device)
            cls._bufs[key] = buf
        return buf
