"""Host side of the fused 3x3 convolution (csrc/conv3x3.hip) that needs no GPU: the tap-major weight packing the kernel reads, the shape
gate the generator consults before it leaves MIOpen, the failure without a GPU, and the LDS bank model the kernel's swizzles were
chosen with (tools/lds_bank_model.py: MI355X serves a ds_read_b128 in four groups of sixteen NON-contiguous lanes)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), os.path.join(ROOT, 'tools')]
import gnerf_hip


def test_weight_packing_is_tap_major_with_the_taps_of_the_correlation():
    o, c = 5, 7
    w = torch.arange(o * c * 9, dtype=torch.float32).reshape(o, c, 3, 3)
    pk = gnerf_hip.pack_conv3x3_weights(w)
    assert pk.shape == (9, o, c) and pk.dtype == torch.float16 and pk.is_contiguous()
    for ky in range(3):
        for kx in range(3):
            assert torch.equal(pk[ky * 3 + kx].float(), w[:, :, ky, kx].half().float())
    # ... i.e. conv2d(x, w, padding=1)[n, o, y, x] = sum over taps t and channels c of pk[t, o, c] * x[n, c, y + t // 3 - 1, x + t % 3 - 1]
    x = torch.randn(1, c, 6, 9)
    want = torch.nn.functional.conv2d(x, w.half().float(), padding=1)
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    got = sum(torch.einsum('oc,nchw->nohw', pk[t].float(), xp[:, :, t // 3:t // 3 + 6, t % 3:t % 3 + 9]) for t in range(9))
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-2)


def test_shape_gate_takes_only_what_the_kernel_tiles():
    x = torch.zeros(2, 64, 8, 32, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    assert not gnerf_hip.conv3x3_epilogue_supported(x, 128)                     # a CPU tensor: never
    with pytest.raises(RuntimeError):
        gnerf_hip.conv3x3_epilogue(x, gnerf_hip.pack_conv3x3_weights(torch.zeros(128, 64, 3, 3)))     # the product path has no CPU fallback


def test_swizzles_of_the_kernel_are_conflict_free_for_the_real_lane_groups():
    import lds_bank_model as M
    # the model itself: sixteen lanes of one group on sixteen distinct 16-byte slots of a 256-byte window cost one cycle per group ...
    g0 = M.GROUPS['ds_read_b128'][0][0]
    assert g0 == [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
    addr = [0] * 64
    for gi, g in enumerate(M.GROUPS['ds_read_b128'][0]):
        for k, lane in enumerate(g):
            addr[lane] = 1024 * gi + 16 * k
    assert M.cycles('ds_read_b128', addr) == (4, 4)
    # ... and all of them on one slot of different rows cost sixteen
    assert M.cycles('ds_read_b128', [256 * lane for lane in range(64)])[0] == 64
    pats = M.conv3x3_patterns()
    reads = {k: v for k, v in pats.items() if 'fragment reads' in k and not k.startswith('(')}
    assert len(reads) == 2 and all(set(v) == {4} for v in reads.values()), reads
    old = next(v for k, v in pats.items() if k.startswith('('))
    assert 8 in old                                                              # the formula the first build used on the input tile was two-way
    assert set(pats['epilogue staging reads (ds_read_b128)']) == {4}
