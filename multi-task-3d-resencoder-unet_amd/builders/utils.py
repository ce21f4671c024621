"""Topology planner and small helpers with the reference's names and argument meaning
(reference: builders/utils.py:268-285 maybe_convert_scalar_to_list, :334-402
get_pool_and_conv_props, :405-426 pad_shape, :428-445 get_n_blocks_per_stage).  Host-side integer
logic only; written from the behaviour, not from the reference text."""
import math

import torch.nn as nn

_CONV_DIM = {nn.Conv1d: 1, nn.Conv2d: 2, nn.Conv3d: 3}


def convert_conv_op_to_dim(conv_op):
    if conv_op not in _CONV_DIM:
        raise ValueError("Unknown dimension. Only 1d 2d and 3d conv are supported. got %s" % str(conv_op))
    return _CONV_DIM[conv_op]


def maybe_convert_scalar_to_list(conv_op, scalar):
    """kernel_size=3 -> [3, 3, 3] for nn.Conv3d; sequences pass through untouched."""
    if isinstance(scalar, (tuple, list)) or hasattr(scalar, "__len__"):
        return scalar
    if conv_op not in _CONV_DIM:
        raise RuntimeError("Invalid conv op: %s" % str(conv_op))
    return [scalar] * _CONV_DIM[conv_op]


def get_matching_convtransp(conv_op=None, dimension=None):
    assert not (conv_op is not None and dimension is not None), \
        "You MUST set EITHER conv_op OR dimension. Do not set both!"
    if conv_op is not None:
        dimension = convert_conv_op_to_dim(conv_op)
    assert dimension in (1, 2, 3), "Dimension must be 1, 2 or 3"
    return {1: nn.ConvTranspose1d, 2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}[dimension]


def get_matching_pool_op(conv_op=None, dimension=None, adaptive=False, pool_type="avg"):
    assert not (conv_op is not None and dimension is not None), \
        "You MUST set EITHER conv_op OR dimension. Do not set both!"
    assert pool_type in ("avg", "max"), "pool_type must be either avg or max"
    if conv_op is not None:
        dimension = convert_conv_op_to_dim(conv_op)
    assert dimension in (1, 2, 3), "Dimension must be 1, 2 or 3"
    name = ("Adaptive" if adaptive else "") + ("Avg" if pool_type == "avg" else "Max") + f"Pool{dimension}d"
    return getattr(nn, name)


def pad_shape(shape, must_be_divisible_by):
    """Round every axis up to the next multiple of its divisor."""
    if not isinstance(must_be_divisible_by, (tuple, list)) and not hasattr(must_be_divisible_by, "__len__"):
        must_be_divisible_by = [must_be_divisible_by] * len(shape)
    assert len(must_be_divisible_by) == len(shape)
    return tuple(int(math.ceil(s / int(d)) * int(d)) for s, d in zip(shape, must_be_divisible_by))


def get_pool_and_conv_props(spacing, patch_size, min_feature_map_size, max_numpool):
    """nnU-Net pooling planner.  Returns (num_pool_per_axis, pool_op_kernel_sizes, conv_kernel_sizes,
    padded_patch_size, must_be_divisible_by) like the reference: the pool kernel list starts with an
    all-ones entry (stage 0), the conv kernel list ends with an all-3 entry (bottleneck)."""
    dim = len(spacing)
    cur_spacing = [float(s) for s in spacing]
    cur_size = [int(s) for s in patch_size]
    pool_kernels = [tuple([1] * dim)]
    conv_kernels = []
    num_pool = [0] * dim
    ksize = [1] * dim
    while True:
        axes = [a for a in range(dim) if cur_size[a] >= 2 * min_feature_map_size]
        if not axes:
            break
        finest = min(cur_spacing[a] for a in axes)
        axes = [a for a in axes if cur_spacing[a] / finest < 2 and num_pool[a] < max_numpool]
        if not axes:
            break
        smallest = min(cur_spacing)
        for a in range(dim):
            if ksize[a] != 3 and cur_spacing[a] / smallest < 2:
                ksize[a] = 3
        step = [1] * dim
        for a in axes:
            step[a] = 2
            num_pool[a] += 1
            cur_spacing[a] *= 2
            cur_size[a] = int(math.ceil(cur_size[a] / 2))
        pool_kernels.append(tuple(step))
        conv_kernels.append(tuple(ksize))
    must_div = [2 ** p for p in num_pool]
    conv_kernels.append(tuple([3] * dim))
    return num_pool, tuple(pool_kernels), tuple(conv_kernels), pad_shape(tuple(patch_size), must_div), must_div


def get_n_blocks_per_stage(num_stages):
    """1, 3, 4, then 6 residual blocks per stage."""
    head = (1, 3, 4)
    return [head[i] if i < len(head) else 6 for i in range(num_stages)]
