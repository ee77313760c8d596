"""`import metayolo` resolves to the MI355X-native implementation (hd_yolo_amd.metayolo), so the reference's entry points
and user code keep their import lines (`from metayolo.models.yolo import Model`, `from metayolo.common import ModelEMA`, ...).
Every `metayolo.X` is the very same module object as `hd_yolo_amd.metayolo.X` (no second copy, relative imports intact)."""
import importlib
import importlib.abc
import importlib.util
import sys

_REAL = 'hd_yolo_amd.'


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != 'metayolo' and not fullname.startswith('metayolo.'):
            return None
        try:
            importlib.import_module(_REAL + fullname)
        except ModuleNotFoundError as e:
            if e.name == _REAL + fullname:
                return None
            raise
        return importlib.util.spec_from_loader(fullname, self)

    def create_module(self, spec):
        return sys.modules[_REAL + spec.name]

    def exec_module(self, module):
        pass


if not any(isinstance(f, _Alias) for f in sys.meta_path):
    sys.meta_path.insert(0, _Alias())
_impl = importlib.import_module(_REAL + 'metayolo')
sys.modules[__name__] = _impl
