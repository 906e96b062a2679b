"""CPU restatement of LPIPS as the reference uses it -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's
cpu_baseline); the product path never imports this.

PARITY UNPINNED.  The reference calls the third-party package ``lpips`` (requirements.txt:16 ``lpips>=0.1.4``; call
sites quality_assessment_module.py:135-146 ``lpips.LPIPS(net='vgg'|'alex')`` and :419-465, :197-224
``_to_lpips_tensor``).  Neither the package nor its torchvision / linear-layer weights are available offline and the
reference holds no LPIPS fixture, so this file restates the *published* forward of lpips 0.1.4
(richzhang/PerceptualSimilarity, ``lpips/lpips.py`` + ``lpips/pretrained_networks.py``) with torch CPU ops and is
exercised with seeded SYNTHETIC weights.  What is restated:

  input       u8 HWC -> float32 / 255 -> x 2 - 1 -> NCHW                       (quality_assessment_module.py:197-224;
              gray is repeated to 3 channels, alpha is dropped)
  scaling     (x - shift) / scale, shift = [-.030, -.088, -.188], scale = [.458, .448, .450]   (ScalingLayer)
  backbone    'alex': torchvision AlexNet.features, taps after ReLU 1..5 (chns 64, 192, 384, 256, 256)
              'vgg' : torchvision VGG16.features, taps relu1_2, relu2_2, relu3_3, relu4_3, relu5_3
                      (chns 64, 128, 256, 512, 512)
  per tap     f = feat / (sqrt(sum_c feat^2) + 1e-10);  d = (f0 - f1)^2;  1x1 conv ``lin`` (no bias);  spatial mean
  value       sum over the five taps

Weight container (what ``QualityAssessmentModule(lpips_weights=...)`` loads with numpy.load(allow_pickle=False)):
a flat .npz whose keys are the ``lpips.LPIPS(...).state_dict()`` names -- ``net.slice<S>.<I>.weight|bias`` for the
convolutions (I = index in torchvision's ``features``), ``lin<K>.model.1.weight`` (shape [1, C, 1, 1]) -- one file
per net.  A user with the package writes it with
``np.savez(path, **{k: v.cpu().numpy() for k, v in lpips.LPIPS(net=net).state_dict().items()})``.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

SHIFT = (-0.030, -0.088, -0.188)
SCALE = (0.458, 0.448, 0.450)

# (kind, ...) per layer.  conv: (cout, cin, k, stride, pad, state-dict key); pool: (k, stride); tap: LPIPS layer index
ARCH: Dict[str, List[Tuple]] = {
    "alex": [
        ("conv", 64, 3, 11, 4, 2, "net.slice1.0"), ("tap", 0),
        ("pool", 3, 2), ("conv", 192, 64, 5, 1, 2, "net.slice2.3"), ("tap", 1),
        ("pool", 3, 2), ("conv", 384, 192, 3, 1, 1, "net.slice3.6"), ("tap", 2),
        ("conv", 256, 384, 3, 1, 1, "net.slice4.8"), ("tap", 3),
        ("conv", 256, 256, 3, 1, 1, "net.slice5.10"), ("tap", 4),
    ],
    "vgg": [
        ("conv", 64, 3, 3, 1, 1, "net.slice1.0"), ("conv", 64, 64, 3, 1, 1, "net.slice1.2"), ("tap", 0),
        ("pool", 2, 2), ("conv", 128, 64, 3, 1, 1, "net.slice2.5"), ("conv", 128, 128, 3, 1, 1, "net.slice2.7"), ("tap", 1),
        ("pool", 2, 2), ("conv", 256, 128, 3, 1, 1, "net.slice3.10"), ("conv", 256, 256, 3, 1, 1, "net.slice3.12"),
        ("conv", 256, 256, 3, 1, 1, "net.slice3.14"), ("tap", 2),
        ("pool", 2, 2), ("conv", 512, 256, 3, 1, 1, "net.slice4.17"), ("conv", 512, 512, 3, 1, 1, "net.slice4.19"),
        ("conv", 512, 512, 3, 1, 1, "net.slice4.21"), ("tap", 3),
        ("pool", 2, 2), ("conv", 512, 512, 3, 1, 1, "net.slice5.24"), ("conv", 512, 512, 3, 1, 1, "net.slice5.26"),
        ("conv", 512, 512, 3, 1, 1, "net.slice5.28"), ("tap", 4),
    ],
}
TAP_CHANNELS = {"alex": (64, 192, 384, 256, 256), "vgg": (64, 128, 256, 512, 512)}


def synthetic_weights(net: str, seed: int = 20260313) -> Dict[str, np.ndarray]:
    """Seeded stand-in for the pretrained weights (He-scaled normal convolutions, small positive biases, positive
    ``lin`` weights like the trained ones): same shapes and key names as ``lpips.LPIPS(net).state_dict()``."""
    rng = np.random.default_rng(seed + (0 if net == "alex" else 1))
    out: Dict[str, np.ndarray] = {}
    for layer in ARCH[net]:
        if layer[0] != "conv":
            continue
        _, cout, cin, k, _, _, key = layer
        std = np.sqrt(2.0 / (cin * k * k))
        out[key + ".weight"] = (rng.standard_normal((cout, cin, k, k)) * std).astype(np.float32)
        out[key + ".bias"] = (rng.uniform(0.0, 0.1, cout)).astype(np.float32)
    for i, c in enumerate(TAP_CHANNELS[net]):
        out[f"lin{i}.model.1.weight"] = rng.uniform(0.0, 2.0 / c, (1, c, 1, 1)).astype(np.float32)
    return out


def to_lpips_tensor(image: np.ndarray):
    """quality_assessment_module.py:197-224."""
    import torch
    img = image.astype(np.float32) / 255.0
    if img.ndim == 2:
        img = np.stack([img, img, img], axis=-1)
    elif img.shape[2] == 1:
        img = np.repeat(img, 3, axis=-1)
    elif img.shape[2] == 4:
        img = img[:, :, :3]
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).unsqueeze(0)
    return t * 2.0 - 1.0


def lpips(img1: np.ndarray, img2: np.ndarray, net: str, weights: Dict[str, np.ndarray], per_layer: bool = False):
    """LPIPS(img1, img2) of two u8 images (cropped to the common top-left rectangle like :449-453)."""
    import torch
    import torch.nn.functional as F
    if img1.shape != img2.shape:
        mh, mw = min(img1.shape[0], img2.shape[0]), min(img1.shape[1], img2.shape[1])
        img1, img2 = img1[:mh, :mw], img2[:mh, :mw]
    shift = torch.tensor(SHIFT, dtype=torch.float32).view(1, 3, 1, 1)
    scale = torch.tensor(SCALE, dtype=torch.float32).view(1, 3, 1, 1)
    with torch.no_grad():
        xs = [(to_lpips_tensor(im) - shift) / scale for im in (img1, img2)]
        vals = []
        for layer in ARCH[net]:
            if layer[0] == "conv":
                _, _, _, _, stride, pad, key = layer
                w = torch.from_numpy(weights[key + ".weight"])
                b = torch.from_numpy(weights[key + ".bias"])
                xs = [F.relu(F.conv2d(x, w, b, stride=stride, padding=pad)) for x in xs]
            elif layer[0] == "pool":
                xs = [F.max_pool2d(x, kernel_size=layer[1], stride=layer[2]) for x in xs]
            else:
                lin = torch.from_numpy(weights[f"lin{layer[1]}.model.1.weight"])
                f0, f1 = [x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True)) + 1e-10) for x in xs]
                d = (f0 - f1) ** 2
                vals.append(float(F.conv2d(d, lin).double().mean()))
    total = float(sum(vals))
    return (total, vals) if per_layer else total


def layer_sizes(net: str, h: int, w: int) -> List[Tuple[int, int]]:
    """(H, W) of the five tap layers for an h x w input (floor rules of Conv2d / MaxPool2d)."""
    out = []
    for layer in ARCH[net]:
        if layer[0] == "conv":
            _, _, _, k, s, p, _ = layer
            h, w = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        elif layer[0] == "pool":
            k, s = layer[1], layer[2]
            h, w = (h - k) // s + 1, (w - k) // s + 1
        else:
            out.append((h, w))
    return out
