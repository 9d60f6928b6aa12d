"""Small host-side helpers shared by bench, smoke and the tests (pure torch, no HIP).

The reference ships no weights offline and ``zero_module`` zeroes every
ResBlock's second conv, every attention ``proj_out`` and the output conv
(reference ``guided_diffusion/nn.py:68-74``), so a freshly built ADM returns
exactly 0.  ``seeded_state_dict`` re-randomises every parameter with a
per-tensor seed derived from its *name*, so the same weights can be rebuilt on
any box from (names, shapes, seed) alone - that is how the golden vectors of
``tests/golden`` and the random-init weights of ``bench.py`` are defined.
"""
import argparse
import math
import os
import random
import zlib

import numpy as np
import torch
import yaml

DATASET_IMAGE_SIZE = {"imagenet64": 64, "imagenet128": 128, "imagenet256": 256, "lsun-churches": 256, "imagenet512": 512,
                      "cifar10": 32}                                   # reference utils.py:23-30


def load_config(folder_path, config_name):
    """Flat key -> value YAML file as an ``argparse.Namespace`` (reference utils.py:42-62: a leading ``configs/`` folder is
    stripped from the name, ``.yaml`` appended when missing; the callers feed the result to ``argparser.set_defaults``)."""
    config_name = str(config_name)
    if config_name.startswith("configs/"):
        config_name = os.path.basename(config_name)
    if not config_name.endswith(".yaml"):
        config_name = config_name + ".yaml"
    with open(os.path.join(str(folder_path), config_name)) as f:
        config = yaml.safe_load(f)
    return argparse.Namespace(**config)


def get_generation_configs(folder_name):
    """Names (without extension) of the YAML files of ``CONFIG/<folder_name>`` (reference utils.py:65-68)."""
    from .paths import CONFIG
    return [x.basename().replace(".yaml", "") for x in CONFIG.joinpath(folder_name).files("*.yaml")]


def seeded_tensor(name, shape, seed=0):
    g = torch.Generator(device="cpu").manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 62))
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if "label_emb" in name or "pos_embed" in name:
        return t * 0.3
    if t.ndim >= 2:                                   # conv / linear weight: unit-gain fan-in init
        return t / math.sqrt(max(1, int(np.prod(shape[1:]))))
    if name.endswith("weight"):                       # norm gamma
        return 1.0 + 0.1 * t
    return 0.1 * t                                    # biases / norm beta


def seeded_state_dict(shapes, seed=0):
    """shapes: mapping name -> shape (e.g. ``{k: v.shape for k, v in model.state_dict().items()}``)."""
    return {k: seeded_tensor(k, tuple(s), seed) for k, s in shapes.items()}


def randomize_parameters_(module, seed=0):
    sd = module.state_dict()
    module.load_state_dict(seeded_state_dict({k: v.shape for k, v in sd.items()}, seed))
    return module


def seed_everything(seed):
    """Counterpart of pytorch_lightning.seed_everything used by the reference script
    (scripts/generate_dataset_score_uncertainty_imagenet.py:39)."""
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    return seed


# The datasets of reference scripts/generate_diffusion_starting_data.py:33-40, in its order: name -> (channels, height, width, classes).
# The script seeds the generator of the k-th dataset with 49394 + k (:31,67).
STARTING_POINT_DATASETS = {"imagenet64": (3, 64, 64, 1000), "imagenet128": (3, 128, 128, 1000), "imagenet128_uvit": (4, 16, 16, 1000),
                           "imagenet256": (4, 32, 32, 1000), "imagenet512": (4, 64, 64, 1000), "cifar10": (3, 32, 32, 10)}


def starting_point_dataset_index(name):
    return list(STARTING_POINT_DATASETS).index(name)


def starting_points(n, shape, num_classes=1000, dataset_index=0):
    """n starting points X_T and labels y drawn the way reference scripts/generate_diffusion_starting_data.py:43-47 draws them: one CPU
    generator seeded 49394 + dataset_index (k-th dataset of the script: imagenet64 0, imagenet128 1, imagenet128_uvit 2, imagenet256 3,
    imagenet512 4), ``randn(n, *shape)`` then ``randint(0, num_classes, (n,))``.  X_T is a prefix of the reference's
    61 000-sample file; its y follows the FULL draw, so labels equal the file's only when n is the script's own count
    (scripts/generate_diffusion_starting_data.py here writes those files)."""
    g = torch.Generator(device="cpu").manual_seed(49394 + int(dataset_index))
    x = torch.randn(n, *shape, generator=g)
    y = torch.randint(0, num_classes, (n,), generator=g)
    return x, y


def params_generation(module):
    """Sum of the in-place version counters of a module's parameters: moves whenever a weight is edited in place through the
    parameter itself (``with torch.no_grad(): p.copy_(...)`` / ``p.add_(...)``, an optimizer step, a LoRA merge), which the
    ``_apply`` / ``load_state_dict`` hooks do not see.  The packed-weight caches of the HIP models compare it at every forward
    and repack when it moved.  Edits through ``p.data`` bypass torch's version counter and cannot be seen: call
    ``model.invalidate_packed_weights()`` after those."""
    return sum(p._version for p in module.parameters())
