"""Import alias: ``import rcmarl_amd`` loads the package that lives in the
directory ``resilient-consensus-based-marl_amd/`` (a hyphenated directory name
cannot be imported directly)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "resilient-consensus-based-marl_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
