"""Importable alias of the product package.

The product lives in `open-solution-mapping-challenge_amd/` (the directory name the project is
specified under, which is not a legal Python identifier); this alias package points its
`__path__` there, so `import mapping_challenge_amd.postprocessing` loads
`open-solution-mapping-challenge_amd/postprocessing.py`.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      'open-solution-mapping-challenge_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _f
