"""`utils` package face for the Level-1 drop-in: ``utils.util`` is this build's (same names as
codes/utils/util.py, no cv2 / torchvision needed); any other ``utils.*`` module of a ``codes/`` tree that is
also on the path still resolves."""
import pkgutil as _pkgutil

__path__ = _pkgutil.extend_path(__path__, __name__)
