"""CPU: this repo's generator / discriminator (PyTorch-op forms of the renderer and the ops, fp32) against the fixtures that
tests/golden/make_golden.py made from the REFERENCE's TriPlaneGenerator and Discriminator on the CPU: BASELINE config 3 at N=4
and one G + D training step (config 5: loss terms and every parameter's gradient norm)."""

import numpy as np
import torch

import gen_cases as C


def test_config3_n4_matches_reference_fixture(golden):
    g = golden('generator_n4.npz')
    G, _ = C.build(torch.device('cpu'))
    ws, out = C.run_config3(G, torch.device('cpu'))
    np.testing.assert_allclose(ws[:, 0, :8].numpy(), g['ws_first'], atol=1e-5)
    assert out['image'].shape == (4, 3, 512, 512)
    for k, ref in (('image_raw', g['image_raw']), ('image_depth', g['image_depth'])):
        assert float(((out[k].numpy() - ref) ** 2).mean()) < 1e-8, k
    assert float(((out['image'][:, :, 4::8, 4::8].numpy() - g['image_sub']) ** 2).mean()) < 1e-8
    np.testing.assert_allclose(out['image'].mean((1, 2, 3)).numpy(), g['image_mean'], atol=1e-4)
    np.testing.assert_allclose(out['image'].std((1, 2, 3)).numpy(), g['image_std'], atol=1e-4)
    # only_depth (triplane.py:83-84): the depth image three times, no superresolution pass
    b = C.batch_on(torch.device('cpu'))
    with torch.no_grad(), C.DI.DetNoise('config3'):
        d = G.synthesis(G.mapping(b['z'], b['c']), b['c'], noise_mode='const', neural_rendering_resolution=64, only_depth=True)
    assert d['image'] is d['image_raw'] is d['image_depth'] and torch.equal(d['image'], out['image_depth'])


def test_config5_training_step_matches_reference_fixture(golden):
    g = golden('train_step.npz')
    G, D = C.build(torch.device('cpu'))
    parts, gen, g_norms, d_norms = C.run_config5(G, D, torch.device('cpu'), force_fp32=True)
    assert abs(parts['loss'] - float(g['loss'])) < 1e-4 * abs(float(g['loss']))
    assert abs(parts['gan'] - float(g['loss_gan'])) < 1e-4
    assert abs(parts['l1'] - float(g['l1'].mean())) < 1e-5 and abs(parts['l1_raw'] - float(g['l1_raw'].mean())) < 1e-5
    assert abs(parts['d_gen'] - float(g['loss_dgen'])) < 1e-4 and abs(parts['d_real'] - float(g['loss_dreal'])) < 1e-4
    assert abs(parts['d_r1'] - float(g['loss_r1'].mean())) < 1e-3 * float(g['loss_r1'].mean())
    np.testing.assert_allclose(gen['image_raw'].detach().numpy(), g['image_raw'], atol=2e-4)
    np.testing.assert_allclose(gen['image_depth'].detach().numpy(), g['image_depth'], atol=2e-4)
    C.compare_norms(g_norms, g['g_names'], g['g_grad_norms'], 5e-3, 'G')
    C.compare_norms(d_norms, g['d_names'], g['d_grad_norms'], 5e-3, 'D')
    np.testing.assert_allclose(G.decoder.net[0].weight.grad.numpy(), g['g_grad_decoder_w1'], rtol=5e-3, atol=1e-6)
    np.testing.assert_allclose(D.b4.out.weight.grad[:4, :64].numpy(), g['d_grad_out_w'], rtol=5e-3, atol=1e-7)
    C.compare_whole_gradients(G, D, golden('train_step_grads.npz'), 1e-3, 'cpu fp32')
