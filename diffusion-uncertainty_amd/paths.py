"""Path constants (reference paths.py:6-58): ROOT is the PARENT of the repository, models live in ROOT/models.
Unlike the reference nothing is created at import time."""
import os
from pathlib import Path

ROOT = Path(os.environ.get("DIFFUSION_UNCERTAINTY_ROOT", Path(__file__).resolve().parent.parent.parent))
MODELS = ROOT / "models"
RESULTS = ROOT / "results"
DIFFUSION_STARTING_POINTS = ROOT / "X_T"
