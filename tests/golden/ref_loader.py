"""Loader for the upstream reference backend -- GOLDEN-GENERATION ONLY.

Runs only in the build container (where /root/reference is mounted).  It is
never imported by the product, by the `-m gpu` tests, by smoke() or by
bench.py: the reference does not exist on the GPU box.

Recipe: the reference's PIVbackend.py imports cv2,
imageio.v3 and torchPIV.PlotterFunctions (which needs PyQt5).  None of those
touch the hot-path arithmetic, so they are replaced by tiny stand-in modules
in sys.modules and PIVbackend.py is loaded by path.
"""
import importlib.util
import os
import re
import sys
import types

import numpy as np

REF_BACKEND = "/root/reference/src/torchPIV/PIVbackend.py"


def _stub_cv2():
    cv2 = types.ModuleType("cv2")
    cv2.IMREAD_GRAYSCALE = 0
    cv2.MORPH_ELLIPSE = 2
    cv2.BORDER_CONSTANT = 0

    def imdecode(buf, flag):
        import io
        from PIL import Image
        try:
            return np.array(Image.open(io.BytesIO(buf.tobytes())).convert("L"))
        except Exception:
            return None

    def getStructuringElement(shape, ksize):
        # OpenCV's 3x3 MORPH_ELLIPSE is the 4-connected cross
        assert tuple(ksize) == (3, 3)
        return np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=np.uint8)

    def dilate(img, kernel, borderType=None, borderValue=0):
        from scipy import ndimage
        return ndimage.binary_dilation(img.astype(bool), structure=kernel.astype(bool),
                                       border_value=0).astype(np.uint8)

    cv2.imdecode = imdecode
    cv2.getStructuringElement = getStructuringElement
    cv2.dilate = dilate
    return cv2


def load_reference():
    if not os.path.exists(REF_BACKEND):
        raise RuntimeError("reference not mounted; goldens can only be made in the build container")
    if "torchPIV_ref_backend" in sys.modules:
        return sys.modules["torchPIV_ref_backend"]
    sys.modules.setdefault("cv2", _stub_cv2())
    iio = types.ModuleType("imageio")
    iio3 = types.ModuleType("imageio.v3")
    iio.v3 = iio3
    sys.modules.setdefault("imageio", iio)
    sys.modules.setdefault("imageio.v3", iio3)
    pkg = types.ModuleType("torchPIV")
    pkg.__path__ = []
    pf = types.ModuleType("torchPIV.PlotterFunctions")

    def natural_keys(text):
        return [int(c) if c.isdigit() else c for c in re.split(r"(\d+)", text)]

    pf.natural_keys = natural_keys
    sys.modules.setdefault("torchPIV", pkg)
    sys.modules.setdefault("torchPIV.PlotterFunctions", pf)
    spec = importlib.util.spec_from_file_location("torchPIV_ref_backend", REF_BACKEND)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["torchPIV_ref_backend"] = mod
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    m = load_reference()
    print("loaded", m.__name__, sorted(m.DeviceMap.devicies))
