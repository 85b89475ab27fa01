"""Three-line `gym` stub: the reference only subclasses gym.Env
(environments/grid_world.py:5).  TEST INFRASTRUCTURE."""


class Env(object):
    pass


class _Spaces:
    pass


spaces = _Spaces()
