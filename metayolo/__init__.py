"""`import metayolo` resolves to the MI355X-native implementation (hd_yolo_amd.metayolo), so the reference's
entry points and user code keep their import lines (`from metayolo.models.yolo import Model`, ...)."""
import importlib
import sys

_impl = importlib.import_module('hd_yolo_amd.metayolo')
sys.modules[__name__] = _impl
for _name in ('models', 'models.yolo', 'models.yolov5', 'models.layers', 'models.yolo_head', 'models.loss',
              'models.utils_general', 'models.utils_torch', 'models.activations', 'models.metrics', 'common',
              'engines', 'engines.torch_utils', 'engines.general'):
    try:
        sys.modules[f'{__name__}.{_name}'] = importlib.import_module(f'hd_yolo_amd.metayolo.{_name}')
    except ModuleNotFoundError:
        pass
