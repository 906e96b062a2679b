"""CPU: the LPIPS restatement (oracle/lpips_oracle.py, torch CPU) behaves like the published metric and the host side of the
HIP path validates caller-supplied weights.  PARITY UNPINNED (no `lpips` package / pretrained weights offline): these
tests pin the restatement's structure -- feature map sizes against torch's own convolution / pooling shapes, identity,
symmetry, the input transform of quality_assessment_module.py:197-224 -- not the package's numbers."""
import numpy as np
import pytest

import _native
from oracle import lpips_oracle as lo


def _img(rng, h, w, cn=3):
    return rng.integers(0, 256, (h, w, cn) if cn else (h, w), dtype=np.uint8)


@pytest.mark.parametrize("net", ["alex", "vgg"])
def test_layer_sizes_follow_torch_shapes(net):
    import torch
    import torch.nn.functional as F
    w = lo.synthetic_weights(net)
    h, wd = 77, 101
    x = torch.zeros(1, 3, h, wd)
    sizes = []
    for layer in lo.ARCH[net]:
        if layer[0] == "conv":
            x = F.conv2d(x, torch.from_numpy(w[layer[6] + ".weight"]), stride=layer[4], padding=layer[5])
        elif layer[0] == "pool":
            x = F.max_pool2d(x, layer[1], layer[2])
        else:
            sizes.append(tuple(x.shape[2:]))
    assert sizes == lo.layer_sizes(net, h, wd)
    hw = (_native.C.c_int * 10)()
    _native.check(_native.load().sr_lpips_layer_sizes(_native.LPIPS_NETS[net], h, wd, hw))
    assert [(hw[2 * i], hw[2 * i + 1]) for i in range(5)] == sizes
    assert [wgt.shape[0] for k, wgt in w.items() if k.startswith("lin")] == [1] * 5
    assert tuple(w[f"lin{i}.model.1.weight"].shape[1] for i in range(5)) == lo.TAP_CHANNELS[net] == _native.LPIPS_TAP_CHANNELS[net]


@pytest.mark.parametrize("net", ["alex", "vgg"])
def test_oracle_identity_symmetry_and_crop(rng, net):
    w = lo.synthetic_weights(net)
    a, b = _img(rng, 64, 80), _img(rng, 64, 80)
    assert lo.lpips(a, a, net, w) == 0.0
    v, layers = lo.lpips(a, b, net, w, per_layer=True)
    assert v > 0 and len(layers) == 5 and abs(v - sum(layers)) < 1e-12
    assert abs(lo.lpips(b, a, net, w) - v) <= 1e-6 * v
    assert lo.lpips(a, b[:60, :70], net, w) == lo.lpips(a[:60, :70], b[:60, :70], net, w)


def test_input_transform_matches_reference_rule(rng):
    """_to_lpips_tensor: /255, x2-1, HWC->NCHW; gray repeated, alpha dropped."""
    g = _img(rng, 5, 7, cn=0)
    t = lo.to_lpips_tensor(g).numpy()
    assert t.shape == (1, 3, 5, 7) and np.array_equal(t[0, 0], t[0, 2])
    assert np.array_equal(t[0, 0], (g.astype(np.float32) / 255.0) * 2.0 - 1.0)
    rgba = _img(rng, 5, 7, cn=4)
    assert np.array_equal(lo.to_lpips_tensor(rgba).numpy(), lo.to_lpips_tensor(rgba[:, :, :3]).numpy())
    from quality_assessment_module import QualityAssessmentModule
    q = QualityAssessmentModule()
    assert q.lpips_model_vgg is None and q.lpips_model_alex is None          # nothing is fetched, nothing is loaded
    assert np.array_equal(q._to_lpips_tensor(rgba), lo.to_lpips_tensor(rgba).numpy())


def test_weight_validation_and_loader(tmp_path):
    w = lo.synthetic_weights("vgg")
    cw, cb, lins, shift, scale = _native.lpips_pack_weights("vgg", w)
    assert len(cw) == len(cb) == 13 and len(lins) == 5 and shift.tolist() == pytest.approx(list(lo.SHIFT))
    bad = dict(w)
    bad["net.slice2.5.weight"] = bad["net.slice2.5.weight"][:, :32]
    with pytest.raises(ValueError):
        _native.lpips_pack_weights("vgg", bad)
    missing = {k: v for k, v in w.items() if not k.startswith("lin3")}
    with pytest.raises(ValueError):
        _native.lpips_pack_weights("vgg", missing)
    with pytest.raises(ValueError):
        _native.lpips_pack_weights("resnet", w)
    p = tmp_path / "w.npz"
    np.savez(p, **w)
    back = _native.load_lpips_weights(str(p))
    assert sorted(back) == sorted(w) and all(np.array_equal(back[k], w[k]) for k in w)
    obj = tmp_path / "pickled.npz"
    np.savez(obj, evil=np.array([{"a": 1}], dtype=object))
    with pytest.raises(ValueError):                       # allow_pickle=False: object arrays are refused
        _native.load_lpips_weights(str(obj))
    n = _native.C.c_int(0)
    _native.check(_native.load().sr_lpips_tile_count(300, 420, 128, _native.C.byref(n)))
    assert n.value == 12
    with pytest.raises(ValueError):
        _native.check(_native.load().sr_lpips_tile_count(300, 420, 100, _native.C.byref(n)))
