from .yolo import *  # noqa: F401,F403  (same re-export chain as the reference: metayolo/models/__init__.py:1)
