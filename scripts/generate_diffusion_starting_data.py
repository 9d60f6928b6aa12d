#!/usr/bin/env python3
"""X_T.pth / y.pth starting points (counterpart of the reference's scripts/generate_diffusion_starting_data.py): the files
scripts/generate_dataset_score_uncertainty_imagenet.py and scripts/generate_images_with_uncertainty_threshold.py read.

Same draws as the reference (:25-69): per dataset, in its order, ONE CPU generator seeded 49394 + k, ``randn(60 000 + 1 000, C, H, W)`` then
``randint(0, classes, (61 000,))``, saved to ``DIFFUSION_STARTING_POINTS/<name>/{X_T,y}.pth``.  Host-only (no GPU work).  ``--datasets`` picks a
subset (the seeds keep following the reference's order), ``--num-samples`` a smaller draw: X_T is then a prefix of the full file but y is
NOT (the labels are drawn after the noise from the same generator), so use the default count for files identical to the reference's.  The
full imagenet128 draw holds 12 GB of fp32 noise, imagenet64 3 GB.
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from diffusion_uncertainty_amd.paths import DIFFUSION_STARTING_POINTS  # noqa: E402
from diffusion_uncertainty_amd.utils import STARTING_POINT_DATASETS, starting_point_dataset_index, starting_points  # noqa: E402


@torch.no_grad()
def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--datasets", nargs="+", default=list(STARTING_POINT_DATASETS), choices=list(STARTING_POINT_DATASETS))
    p.add_argument("--num-samples", type=int, default=60_000 + 1_000, help="the reference's count: 60 000 + 1 000 extra")
    a = p.parse_args(argv)
    for name in a.datasets:
        c, h, w, classes = STARTING_POINT_DATASETS[name]
        k = starting_point_dataset_index(name)
        print("Generating for", name)
        dest = Path(DIFFUSION_STARTING_POINTS) / name
        dest.mkdir(parents=True, exist_ok=True)
        x, y = starting_points(a.num_samples, (c, h, w), num_classes=classes, dataset_index=k)
        print("Stats of gen_data:\\n\\tMean:", x.mean().item(), "\\n\\tStandard Deviation:", x.std().item(), "\\n\\tMin:", x.min().item(), "\\n\\tMax:", x.max().item())
        torch.save(x, dest / "X_T.pth")
        torch.save(y, dest / "y.pth")
        print("Using seed:", 49394 + k)
        print(f"Saved {name} to {dest / 'X_T.pth'} and {dest / 'y.pth'}")
    print("Done!")


if __name__ == "__main__":
    main()
