#!/usr/bin/env python3
"""Time-step wise uncertainty threshold for each pixel (counterpart of the reference's scripts/compute_threshold_pixel_wise.py on the
MI355X engine; SURVEY.md 8f rank 2: generate maps -> thresholds -> guided sampling).

Same flags and the same flow as the reference's ``main`` (:44-116): the dataset folders are outputs of
scripts/generate_dataset_score_uncertainty_imagenet.py (``args.yaml``, ``uncertainty_*.pth``, ``gen_images_*.pth``; a name that is not a path
is looked up under ``RESULTS/score-uncertainty``); per uncertainty file with at least 100 samples the value at rank ``int(N * perc)`` over
the N samples, per timestep and pixel, in fp16 (``load_uncertainty_datasets`` :118-165); several files: the same order statistic over the
per-file thresholds.  Saved to ``RESULTS/thresholds/<dataset>/thresholds_<scheduler_type>_perc=<perc>.pth`` (``[S, 3, H, W]`` fp16) with
``config_<scheduler_type>_perc=<perc>.yaml`` beside it (the arguments + the dataset's own ``args.yaml`` as ``dataset_config``), which
scripts/generate_images_with_uncertainty_threshold.py reads.  The selection runs on the GPU (dua_pixelwise_order_statistic, bit-identical
to the reference's argsort + gather: tests/golden G14); ``--on-cpu`` is refused - the engine has no CPU path.
"""
import argparse
import os
import sys
from pathlib import Path

import torch
import yaml

ROOT_DIR = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT_DIR))

from diffusion_uncertainty_amd.paths import CONFIG, RESULTS, ROOT  # noqa: E402
from diffusion_uncertainty_amd.thresholds import thresholds_from_datasets  # noqa: E402
from diffusion_uncertainty_amd.utils import load_config  # noqa: E402


def assert_validity_folders_(args):
    """reference :21-40: every folder has an args.yaml, the folders agree on dataset and scheduler type, and hold both kinds of files."""
    scheduler_type = dataset = None
    for folder in args.dataset_folders:
        folder = Path(folder)
        assert folder.exists(), f"dataset folder {folder} does not exist"
        assert (folder / "args.yaml").exists(), f"args.yaml not found in {folder}"
        with open(folder / "args.yaml") as f:
            config = yaml.safe_load(f)
        if dataset is None:
            dataset = config["dataset"]
        else:
            assert dataset == config["dataset"], f"datasets do not match: {dataset=} {config['dataset']=}"
        if scheduler_type is None:
            scheduler_type = config["scheduler_type"]
        else:
            assert scheduler_type == config["scheduler_type"], f"scheduler types do not match: {scheduler_type=} {config['scheduler_type']=}"
        assert list(folder.glob("gen_images*.pth")), f"no gen_images files found in {folder}"
        assert list(folder.glob("uncertainty*.pth")), f"no uncertainty files found in {folder}"


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--dataset-folders", "--datasets", type=str, nargs="+", required=False, dest="dataset_folders",
                   help="dataset folders holding the uncertainty and gen_images files")
    p.add_argument("--batch-size", type=int, default=128)
    p.add_argument("--on-cpu", action="store_true", dest="on_cpu")
    p.add_argument("--perc", type=float, default=0.15, help="percentage of samples to estimate the threshold from")
    options = [x.basename().replace(".yaml", "") for x in CONFIG.joinpath("threshold").files("*.yaml")]
    p.add_argument("--config", type=str, help="name of a config/threshold/*.yaml file", choices=options or None)
    args = p.parse_args(argv)
    if args.config is not None:
        print("Loading config file - ignoring other arguments")
        args = load_config(CONFIG.joinpath("threshold"), args.config)
    assert args.dataset_folders, "--dataset-folders (or --config) must be given"
    return args


def main(argv=None):
    args = parse_args(argv)
    folders = []
    for folder in args.dataset_folders:                                  # canonical paths (:56-67)
        folder = Path(folder)
        if not folder.exists():
            folder = Path(RESULTS) / "score-uncertainty" / folder
            if not folder.exists():
                raise ValueError(f"dataset folder {folder} does not exist")
        folders.append(folder)
    args.dataset_folders = folders
    if args.on_cpu:
        raise SystemExit("--on-cpu: the engine has no CPU path (the selection runs in libdua_hip.so)")
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: the engine has no CPU path")
    assert_validity_folders_(args)
    with open(folders[0] / "args.yaml") as f:
        config = yaml.safe_load(f)
    print("Config:")
    print(config)
    timestep_thresholds = thresholds_from_datasets(folders, args.perc, device="cuda")
    print(f"{timestep_thresholds.shape=}")
    threshold_folder = Path(RESULTS) / "thresholds" / config["dataset"]
    threshold_folder.mkdir(parents=True, exist_ok=True)
    out = threshold_folder / f'thresholds_{config["scheduler_type"]}_perc={args.perc}.pth'
    torch.save(timestep_thresholds, out)
    print("Saved thresholds to", out)
    args_dict = dict(args.__dict__)
    args_dict["dataset_config"] = config
    args_dict["dataset_folders"] = [os.path.relpath(str(x), str(ROOT)) for x in folders]
    with open(threshold_folder / f'config_{config["scheduler_type"]}_perc={args.perc}.yaml', "w") as f:
        yaml.safe_dump(args_dict, f)
    return out


if __name__ == "__main__":
    main()
