"""Import alias for the package directory ``mrphy.py_amd/``.

The package directory carries a dot in its name (the repo contract), which the
normal import system cannot spell.  ``import mrphy_amd`` executes this shim,
which loads ``mrphy.py_amd/__init__.py`` as the package ``mrphy_amd`` and
replaces itself in ``sys.modules``; after that ``import mrphy_amd.sims`` etc.
resolve inside the package directory as usual.
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        'mrphy.py_amd')
_spec = importlib.util.spec_from_file_location(
    'mrphy_amd', os.path.join(_pkg_dir, '__init__.py'),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['mrphy_amd'] = _mod
_spec.loader.exec_module(_mod)
