"""Path constants of the hot path's callers (reference paths.py:6-40,172-203).

Same names and the same values relative to ROOT as the reference: ``ROOT`` is the PARENT of the repository
(reference paths.py:6 takes three ``.parent`` steps from the module file), ``CONFIG = ROOT/'config'`` (:8),
``RESULTS = ROOT/'results'`` (:12), ``DIFFUSION_STARTING_POINTS = RESULTS/'diffusion_starting_points'`` (:36),
``MODELS = ROOT/'models'`` (:172), ``SCORE_UNCERTAINTY`` (:185), ``THRESHOLD`` (:189).  The objects answer the few
``path.Path`` calls the reference's callers make on them (``.files(pattern)``, ``.basename()``, ``.joinpath``, ``/``).

Two deliberate differences: (1) nothing is created at import time (the reference ``mkdir``s about twenty folders
when the module is imported); a writer creates its own destination, and ``make_dirs()`` - ``alias_as_reference(create_dirs=True)`` - creates the
ones a reference script relies on.  (2) ``DIFFUSION_UNCERTAINTY_ROOT`` overrides
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
FID = RESULTS / "fid"                                                            # (:40; imported by the threshold script, unused on this path)
STABLE_DIFFUSION_UNCERTAINTY_GUIDANCE = RESULTS / "stable_diffusion_uncertainty_guidance"    # (:24; the SD guidance script's output folder)
INTERMEDIATES = RESULTS / "intermediates"                                        # (:178)
GENERATIONS = RESULTS / "generations"                                            # (:48)
MODELS = ROOT / "models"
DATA = ROOT / "data"


def make_dirs():
    """Create the folders the reference's paths.py creates when it is imported (:9-40,172-190) - the ones on this path: a reference script
    ``mkdir()``s its output folder non-recursively and relies on them.  This package's own scripts create what they write to;
    ``alias_as_reference(create_dirs=True)`` calls this, so that a reference script finds the layout it expects."""
    for d in (RESULTS, DIFFUSION_STARTING_POINTS, SCORE_UNCERTAINTY, THRESHOLD, STABLE_DIFFUSION_UNCERTAINTY_GUIDANCE, MODELS):
        d.mkdir(parents=True, exist_ok=True)
