"""Golden vectors for SURVEY.md 8(f-3): the ensemble statistics and the export files, made by the
REFERENCE'S OWN CODE -- PIVWorker.run (workers.py:29-124: statistics block :85-118) and save_table /
save_binary (PlotterFunctions.py:48-65) -- run here, in the build container only:

    python tests/golden/make_golden_stats.py

PyQt5 is absent, so QObject / pyqtSignal / QMessageBox are replaced by inert stand-ins (they carry no
arithmetic: the signals just record what is emitted); cv2 / imageio as in ref_loader.py.  The fixture
holds data only: the per-pair fields the worker emitted, the final table, and the BYTES of every file the
reference wrote.
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
REF = "/root/reference/src/torchPIV"


class _Signal:
    def __init__(self, *types_):
        self.sent = []

    def emit(self, *payload):
        self.sent.append(payload[0] if len(payload) == 1 else payload)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_worker_modules():
    qt = types.ModuleType("PyQt5")
    qtw = types.ModuleType("PyQt5.QtWidgets")
    qtc = types.ModuleType("PyQt5.QtCore")
    qtw.QMessageBox = type("QMessageBox", (), {})

    class QObject:
        def __init__(self, *a, parent=None, **k):
            # pyqtSignal class attributes become per-instance recorders
            for klass in type(self).__mro__:
                for key, val in vars(klass).items():
                    if isinstance(val, _Signal):
                        setattr(self, key, _Signal())

    qtc.QObject = QObject
    qtc.pyqtSignal = _Signal
    qtc.QThread = type("QThread", (), {})
    qtc.QTimer = type("QTimer", (), {})
    qt.QtWidgets, qt.QtCore = qtw, qtc
    sys.modules.update({"PyQt5": qt, "PyQt5.QtWidgets": qtw, "PyQt5.QtCore": qtc})
    pkg = types.ModuleType("torchPIV")
    pkg.__path__ = []
    sys.modules["torchPIV"] = pkg
    pf = _load("torchPIV.PlotterFunctions", os.path.join(REF, "PlotterFunctions.py"))       # the real module (pandas is here)
    from ref_loader import load_reference
    backend = load_reference()            # PIVbackend.py by path (keeps the real PlotterFunctions registered above)
    sys.modules["torchPIV.PIVbackend"] = backend
    workers = _load("torchPIV.workers", os.path.join(REF, "workers.py"))
    return pf, backend, workers


def main():
    from PIL import Image
    from make_golden import make_frames
    pf, backend, workers = load_worker_modules()
    out = {}
    H, W = 192, 256
    kinds = ["wavy", "vortex", "shear", "wavy", "uniform", "vortex"]
    with tempfile.TemporaryDirectory() as root:
        folder = os.path.join(root, "run A")               # (a name with a blank, as users have them)
        os.mkdir(folder)
        frames = []
        for i, kind in enumerate(kinds):
            a, b = make_frames(H, W, kind, 2.0 + i, True, 60 + i)       # special regions: every pair has invalid vectors
            frames.append((a.numpy(), b.numpy()))
            Image.fromarray(a.numpy(), "L").save(os.path.join(folder, f"img{i + 8}_a.bmp"))
            Image.fromarray(b.numpy(), "L").save(os.path.join(folder, f"img{i + 8}_b.bmp"))
        out["frames_a"] = np.stack([f[0] for f in frames])
        out["frames_b"] = np.stack([f[1] for f in frames])
        kw = dict(wind_size=32, overlap=16, multipass=2, multipass_mode="CWS", dt=2, scale=0.5, multipass_scale=2.0,
                  folder_mode="pairs", device="cpu", file_fmt="bmp")
        out["kw"] = np.array([32, 16, 2, 1, 2])
        out["scale"] = np.array([0.5])
        for opt, tag in (("Save all text", "txt"), ("Save all binary", "bin")):
            save_dir = os.path.join(root, "Out_" + tag)
            P = pf.PIVparams()
            for k, v in dict(kw, folder=folder, save_opt=opt, save_dir=save_dir).items():
                setattr(P, k, v)
            w = workers.PIVWorker(P)
            w.run()
            table = w.finished.sent[-1]
            pairs = w.output.sent
            if tag == "txt":
                out["n_pairs"] = np.array([len(pairs)])
                for j, o in enumerate(pairs):
                    for key, short in (("x[mm]", "x"), ("y[mm]", "y"), ("Vx[m/s]", "u"), ("Vy[m/s]", "v")):
                        out[f"pair{j}_{short}"] = np.asarray(o[key])
                out["table_keys"] = np.array(list(table.keys()))
                for j, (key, val) in enumerate(table.items()):
                    out[f"table_{j}"] = np.asarray(val)
            names = sorted(os.listdir(save_dir))
            out[f"{tag}_names"] = np.array(names)
            for j, nm in enumerate(names):
                out[f"{tag}_file{j}"] = np.frombuffer(open(os.path.join(save_dir, nm), "rb").read(), dtype=np.uint8)
            print(f"{opt}: {len(pairs)} pairs, files {names}")
    path = os.path.join(HERE, "g9_stats.npz")
    np.savez_compressed(path, **out)
    print(f"g9_stats: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
