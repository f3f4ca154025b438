"""TEST INFRASTRUCTURE (oracle shim): permissive stand-in for third-party modules that the
reference imports at module scope but that are outside the hot path (neptune, imgaug, cv2,
pycocotools, lightgbm, xgboost, imageio, pydot_ng, IPython).  Attribute access yields a
class that can be instantiated, subclassed and called with anything."""
import sys
import types


class Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return Any()

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return Any()

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        full = self.__name__ + '.' + name
        if full in sys.modules:
            return sys.modules[full]
        return type(name, (Any,), {})


def install(*names):
    for name in names:
        parts = name.split('.')
        for i in range(1, len(parts) + 1):
            sub = '.'.join(parts[:i])
            if sub not in sys.modules:
                m = _StubModule(sub)
                m.__path__ = []
                sys.modules[sub] = m
                if i > 1:
                    setattr(sys.modules['.'.join(parts[:i - 1])], parts[i - 1], m)
