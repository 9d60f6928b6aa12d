"""Path constants of the hot path's callers (reference paths.py:6-40,172-203).

Same names and the same values relative to ROOT as the reference: ``ROOT`` is the PARENT of the repository
(reference paths.py:6 takes three ``.parent`` steps from the module file), ``CONFIG = ROOT/'config'`` (:8),
``RESULTS = ROOT/'results'`` (:12), ``DIFFUSION_STARTING_POINTS = RESULTS/'diffusion_starting_points'`` (:36),
``MODELS = ROOT/'models'`` (:172), ``SCORE_UNCERTAINTY`` (:185), ``THRESHOLD`` (:189).  The objects answer the few
``path.Path`` calls the reference's callers make on them (``.files(pattern)``, ``.basename()``, ``.joinpath``, ``/``).

Two deliberate differences: (1) nothing is created at import time (the reference ``mkdir``s about twenty folders
when the module is imported); a writer creates its own destination.  (2) ``DIFFUSION_UNCERTAINTY_ROOT`` overrides
ROOT, and when ``ROOT/config`` does not exist the repository's own ``config/`` is used, so a fresh checkout finds its
``config/generation/*.yaml`` (the reference's layout resolves CONFIG outside its own checkout).
"""
import fnmatch
import os
import pathlib


class Path(type(pathlib.Path())):
    """pathlib path with the ``path.Path`` methods the reference's scripts call on these constants."""

    def files(self, pattern="*"):
        """path.Path.files: the regular files of this directory matching ``pattern`` (script :174, utils.py:65)."""
        if not self.is_dir():
            return []
        return sorted(p for p in self.iterdir() if p.is_file() and fnmatch.fnmatch(p.name, pattern))

    def basename(self):
        """path.Path.basename: the last component, as a str (callers chain ``.replace('.yaml', '')``, script :175)."""
        return self.name

    def makedirs_p(self):
        self.mkdir(parents=True, exist_ok=True)
        return self


_REPO = Path(__file__).resolve().parent.parent
ROOT = Path(os.environ.get("DIFFUSION_UNCERTAINTY_ROOT", _REPO.parent))

CONFIG = ROOT / "config"
if not CONFIG.exists():
    CONFIG = _REPO / "config"
RESULTS = ROOT / "results"
DIFFUSION_STARTING_POINTS = RESULTS / "diffusion_starting_points"
SCORE_UNCERTAINTY = RESULTS / "score-uncertainty"
THRESHOLD = RESULTS / "thresholds"
MODELS = ROOT / "models"
