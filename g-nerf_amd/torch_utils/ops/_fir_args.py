"""Argument normalisation shared by upfirdn2d.py and filtered_lrelu.py (scaling factors, paddings, filter sizes).
Same accepted forms and the same AssertionError-on-misuse behaviour as the reference's private helpers
(torch_utils/ops/upfirdn2d.py:36-69, filtered_lrelu.py:36-54)."""

import numpy as np
import torch


def as_xy(value, kinds=(int,)):
    """int -> (v, v); [x, y] -> (x, y).  Elements must be instances of `kinds`."""
    if isinstance(value, kinds) and not isinstance(value, bool):
        return value, value
    assert isinstance(value, (list, tuple))
    assert all(isinstance(v, kinds) for v in value)
    x, y = value                        # any other length: ValueError from the unpacking, as upstream (upfirdn2d.py:36-43)
    return x, y


def parse_scaling(scaling):
    sx, sy = as_xy(scaling)
    assert sx >= 1 and sy >= 1
    return sx, sy


def parse_padding(padding, kinds=(int,)):
    """int | [x, y] | [x0, x1, y0, y1] -> (x0, x1, y0, y1)."""
    if isinstance(padding, kinds) and not isinstance(padding, bool):
        padding = [padding] * 4
    assert isinstance(padding, (list, tuple))
    assert all(isinstance(v, kinds) for v in padding)
    values = [int(v) for v in padding]
    if len(values) == 2:
        values = [values[0], values[0], values[1], values[1]]
    x0, x1, y0, y1 = values             # any other length: ValueError from the unpacking, as upstream (upfirdn2d.py:45-54)
    return x0, x1, y0, y1


def filter_size(f, strict=True):
    """(width, height) of a FIR filter given as None, [taps] or [fh, fw]."""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in (1, 2)
    fw, fh = int(f.shape[-1]), int(f.shape[0])
    if strict:
        assert fw >= 1 and fh >= 1
    return fw, fh


NUMPY_INTS = (int, np.integer)
