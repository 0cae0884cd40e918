"""Importable alias of the ``panoptic-forecasting_amd/`` package directory.

The package directory carries the reference project's hyphenated name, which
Python cannot import directly; this one-file alias points ``__path__`` at it so
``import panoptic_forecasting_amd`` (and its submodules) resolve there.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      'panoptic-forecasting_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
