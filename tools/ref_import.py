"""Import the read-only reference (/root/reference) in THIS container with stubs for the
third-party packages that are absent (cv2, skimage, numba, torchvision, ...).

Development-container tool only: used by tools/make_golden_forward.py to generate the
fixtures under tests/golden/.  Nothing under tests/ -m gpu, bench.py or smoke() imports it.
"""
import sys
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    if "cv2" not in sys.modules:
        _stub("cv2")
    if "skimage" not in sys.modules:
        sk = _stub("skimage")
        sk.segmentation = _stub("skimage.segmentation", watershed=None)
        sk.draw = _stub("skimage.draw", polygon=None)
    if "numba" not in sys.modules:
        def njit(*a, **k):
            if a and callable(a[0]):
                return a[0]
            return lambda f: f
        _stub("numba", njit=njit, prange=range)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms")
    if REF not in sys.path:
        sys.path.insert(0, REF)


def import_cellvit():
    install_stubs()
    from models.segmentation.cell_segmentation import cellvit  # noqa
    return cellvit
