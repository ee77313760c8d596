"""`import hnet` resolves to the MI355X-native implementation (hd_yolo_amd.hnet): `from hnet.segmentation import PanopticSeg`,
`from hnet.hnet import HNet` keep the reference's import lines (same aliasing as the top-level `metayolo` package)."""
import importlib
import importlib.abc
import importlib.util
import sys

_REAL = 'hd_yolo_amd.'


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != 'hnet' and not fullname.startswith('hnet.'):
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


if not any(type(f).__name__ == '_Alias' and f.__module__ == __name__ for f in sys.meta_path):
    sys.meta_path.insert(0, _Alias())
_impl = importlib.import_module(_REAL + 'hnet')
sys.modules[__name__] = _impl
