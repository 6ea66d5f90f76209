"""f3: MaterialNet (DINOv2 ViT-B + two DPT heads) against the reference module run under the stub imports, with identical
name-seeded weights (tests/golden/materialnet.npz); fp64 on the CPU."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def net():
    from materialist_amd.materialnet import MaterialNet, init_from_names

    return init_from_names(MaterialNet().double().eval())


def test_state_dict_names_and_size_match_the_reference(golden_dir, net):
    g = np.load(os.path.join(golden_dir, "materialnet.npz"))
    assert sorted(net.state_dict().keys()) == list(g["names"])          # matnet_weights.pth would load unchanged
    assert sum(p.numel() for p in net.parameters()) == int(g["n_params"])   # 108.36 M


def test_forward_matches_the_reference(golden_dir, net):
    g = np.load(os.path.join(golden_dir, "materialnet.npz"))
    x = torch.from_numpy(g["x"])                                           # 70 x 98: exercises the position-grid resampling
    with torch.no_grad():
        taps = net.pretrained.taps(x)
        out = net(x)
    np.testing.assert_allclose(taps[3][0].numpy(), g["feat3_patch"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(taps[0][1].numpy(), g["feat0_cls"], rtol=1e-8, atol=1e-10)
    for k in ("depth", "albedo", "roughness", "metallic", "normal"):
        assert out[k].shape == g[k].shape
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=1e-7, atol=1e-9, err_msg=k)
    assert out["depth"].shape == (1, 1, 70, 98) and out["normal"].shape == (1, 3, 70, 98)
    np.testing.assert_allclose(out["normal"].norm(dim=1).numpy(), 1.0, atol=1e-6)


def test_network_input_size_matches_resize(golden_dir):
    from materialist_amd.materialnet import network_input_size

    g = np.load(os.path.join(golden_dir, "materialnet.npz"))
    for (w, h), (nw, nh) in zip(g["sizes_in"], g["sizes_out"]):
        assert network_input_size(int(w), int(h)) == (int(nw), int(nh))


def test_infer_image_shapes():
    from materialist_amd.materialnet import MaterialNet

    m = MaterialNet().eval()
    img = (np.random.default_rng(0).random((40, 56, 3)) * 255).astype(np.uint8)
    out = m.infer_image(img, input_size=70)
    assert out["albedo"].shape == (40, 56, 3) and out["depth"].shape == (40, 56) and out["roughness"].shape == (40, 56)
    assert np.isfinite(out["normal"]).all()
