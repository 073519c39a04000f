#!/usr/bin/env python3
"""Does oracle/philox_ref.py reproduce torch.rand on this device?  (Run on the GPU box; prints one JSON line per case.)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import numpy as np
import torch
from oracle import philox_ref as P
dev = torch.device('cuda', 0)
pr = torch.cuda.get_device_properties(dev)
mp, mt = pr.multi_processor_count, pr.max_threads_per_multi_processor
print(json.dumps({'device': pr.name, 'multi_processor_count': mp, 'max_threads_per_multi_processor': mt}))
gen = torch.cuda.default_generators[0]
for seed, shape in [(0, (7,)), (123, (1000,)), (5, (4, 16384, 48, 1)), (5, (65536, 48)), (2 ** 40 + 17, (3, 333, 12)), (9, (4 * 4096 * 96,)), (9, (1, 4096, 96, 1))]:
    torch.manual_seed(seed)
    # two draws in a row, like the renderer's: the second starts at the offset the first left
    o0 = gen.get_offset()
    a = torch.rand(shape, device=dev)
    o1 = gen.get_offset()
    b = torch.rand(shape, device=dev)
    o2 = gen.get_offset()
    n = a.numel()
    wa, p1 = P.torch_rand(n, gen.initial_seed(), o0, mp, mt)
    wb, p2 = P.torch_rand(n, gen.initial_seed(), o1, mp, mt)
    ea = bool(np.array_equal(a.cpu().numpy().reshape(-1), wa))
    eb = bool(np.array_equal(b.cpu().numpy().reshape(-1), wb))
    print(json.dumps({'seed': seed, 'shape': shape, 'offsets': [o0, o1, o2], 'predicted_offsets': [o0, p1, p2], 'first_equal': ea, 'second_equal': eb,
                      'max_abs_diff': float(np.abs(a.cpu().numpy().reshape(-1) - wa).max())}))
