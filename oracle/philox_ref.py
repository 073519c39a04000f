"""TEST INFRASTRUCTURE (CPU oracle; not part of the product path): torch's device uniform draws, restated in numpy.

`torch.rand(..., device='cuda')` (the reference's two draws, renderer.py:190 `torch.rand_like` and :241 `torch.rand`) is ATen's
`distribution_elementwise_grid_stride_kernel` over a Philox4x32-10 counter-based generator (ATen/native/cuda/DistributionTemplates.h
in the installed torch 2.10; rocRAND's philox engine underneath on ROCm).  What it does, element by element:

  * launch shape: block 256; grid = min(ceil(numel / 256), multiProcessorCount * (maxThreadsPerMultiProcessor / 256));
    G = 256 * grid threads; unroll 4
  * thread `tid` owns the Philox stream (key = seed, subsequence = tid, offset = the generator's philox offset): its j-th call
    returns the 128-bit block philox4x32_10(counter = (offset / 4 + j) as 64-bit low half | tid as 64-bit high half, key)
  * element `li` is written by thread li % G from call j = (li / G) / 4, component (li / G) % 4, as
    u = float(x) * 2^-32 + 2^-32  (rocRAND's uniform: (0, 1]), then 1.0 -> 0.0 (ATen's bound reversal)
  * afterwards the generator's offset has advanced by 4 * ceil(numel / (4 G))

Pinned against the device generator itself: tests/test_gpu_parity.py::test_philox_restatement_matches_torch_rand (a GPU test, the
only place a device generator exists); the in-kernel draws of csrc/render*.inl are then held to this file AND to torch.rand.
"""

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Ten Philox rounds on arrays of uint32 counters (c0 lowest word) with key (k0, k1): four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32).copy() for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * M0
            p1 = c2.astype(np.uint64) * M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


def grid_threads(numel, multi_processor_count, max_threads_per_multi_processor):
    blocks = min((numel + 255) // 256, multi_processor_count * (max_threads_per_multi_processor // 256))
    return 256 * blocks


def offset_increment(numel, g_threads):
    return ((numel - 1) // (g_threads * 4) + 1) * 4


def uniform_from_bits(x):
    """rocRAND's float uniform on a uint32 array, then ATen's (0,1] -> [0,1) reversal."""
    u = (x.astype(np.float32) * np.float32(2.0 ** -32) + np.float32(2.0 ** -32)).astype(np.float32)
    return np.where(u == np.float32(1.0), np.float32(0.0), u)


def torch_rand(numel, seed, offset, multi_processor_count, max_threads_per_multi_processor, index=None):
    """The `numel` floats torch.rand(numel, device) yields with the device generator at (seed, offset) -- or only the elements
    `index` (an int array) of them.  Returns (values float32, new offset)."""
    g = grid_threads(numel, multi_processor_count, max_threads_per_multi_processor)
    li = np.arange(numel, dtype=np.int64) if index is None else np.asarray(index, dtype=np.int64)
    tid, m = li % g, li // g
    ctr = np.uint64(offset // 4) + (m // 4).astype(np.uint64)
    out = philox4x32_10((ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32),
                        (tid & 0xFFFFFFFF).astype(np.uint32), (tid >> 32).astype(np.uint32),
                        seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    comp = (m % 4).astype(np.int64)
    bits = np.choose(comp, out)
    return uniform_from_bits(bits), offset + offset_increment(numel, g)
