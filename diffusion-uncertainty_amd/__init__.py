"""MI355X-native uncertainty-guided diffusion sampling engine (hot path only).

Python host side mirroring the reference's duck-typed API
(``init_model`` / ``generate_samples`` / ``get_uncertainty_scheduler`` /
scheduler ``.step()``) on top of a C-ABI HIP library (``include/dua_hip.h``).
There is no CPU fallback: every compute entry point raises if the HIP library
or a GPU tensor is missing.
"""
__version__ = "0.1.0"

import importlib as _importlib
import importlib.abc as _abc
import importlib.machinery as _machinery
import sys as _sys


class _AliasFinder(_abc.MetaPathFinder, _abc.Loader):
    """Resolves ``<alias>[.sub.module]`` to the SAME module object as ``<this package>[.sub.module]`` (one copy of every
    class, so ``isinstance`` checks hold whichever name a caller imported it under)."""

    def __init__(self, alias, real):
        self.alias, self.real = alias, real

    def find_spec(self, fullname, path=None, target=None):
        if fullname != self.alias and not fullname.startswith(self.alias + "."):
            return None
        real = self.real + fullname[len(self.alias):]
        try:
            module = _importlib.import_module(real)
        except ModuleNotFoundError as e:
            if e.name == real:
                return None                                   # -> ModuleNotFoundError under the alias name
            raise
        spec = _machinery.ModuleSpec(fullname, self, is_package=hasattr(module, "__path__"))
        spec._dua_module = module
        return spec

    def create_module(self, spec):
        return spec._dua_module

    def exec_module(self, module):
        pass


def alias_as_reference(name="diffusion_uncertainty", create_dirs=False):
    """Make ``import diffusion_uncertainty.<module>`` resolve to this package: the reference's
    scripts keep their import blocks.  ``create_dirs=True`` also creates the results / models folders under ``paths.ROOT`` that the
    reference creates when its ``paths`` module is imported (``paths.make_dirs``): its scripts ``mkdir()`` their output folders
    non-recursively and rely on them.  Off by default - importing or aliasing this package writes nothing.  Idempotent; returns the package."""
    this = _sys.modules[__name__]
    if not any(isinstance(f, _AliasFinder) and f.alias == name for f in _sys.meta_path):
        _sys.meta_path.insert(0, _AliasFinder(name, __name__))
    _sys.modules[name] = this
    if create_dirs:                                   # the reference's paths.py creates its folders at import; its scripts rely on them
        from . import paths as _paths
        _paths.make_dirs()
    return this
