"""Drop-in surface of the reference's `metayolo` package for the detection hot path, backed by the MI355X
HIP kernels of hd_yolo_amd.  Only what the hot-path modules and the two entry points import is provided:
LOGGER, check_version, load_cfg (reference: metayolo/__init__.py:62-75, :90, :93-102, :135-145) — without
the reference's cv2 / matplotlib / star-import side effects.
"""
import logging
import os

RANK = int(os.getenv('RANK', -1))
VERBOSE = str(os.getenv('YOLOv5_VERBOSE', True)).lower() == 'true'


def _make_logger():
    log = logging.getLogger('yolov5')
    if not log.handlers:
        level = logging.INFO if (VERBOSE and RANK in (-1, 0)) else logging.WARNING
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter('%(message)s'))
        h.setLevel(level)
        log.addHandler(h)
        log.setLevel(level)
        log.propagate = False
    return log


LOGGER = _make_logger()


def _vtuple(v):
    out = []
    for part in str(v).split('+')[0].split('.')[:4]:
        digits = ''.join(ch for ch in part if ch.isdigit())
        out.append(int(digits) if digits else 0)
    return tuple(out)


def check_version(current='0.0.0', minimum='0.0.0', name='version ', pinned=False, hard=False, verbose=False):
    cur, req = _vtuple(current), _vtuple(minimum)
    ok = cur == req if pinned else cur >= req
    msg = f'{name}{minimum} required by YOLOv5, but {name}{current} is currently installed'
    if hard:
        assert ok, msg
    if verbose and not ok:
        LOGGER.warning(msg)
    return ok


def load_cfg(cfg):
    """A dict is returned as is; a path is read as YAML."""
    if isinstance(cfg, dict):
        return cfg
    import yaml
    with open(cfg, encoding='ascii', errors='ignore') as f:
        return yaml.safe_load(f)
