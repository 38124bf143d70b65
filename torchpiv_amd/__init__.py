"""torchpiv_amd -- MI355X-native PIV cross-correlation engine, a drop-in for the
OfflinePIV(...) generator API of NikNazarov/TorchPIV.

    from torchpiv_amd import OfflinePIV
    for x, y, u, v in OfflinePIV(folder, device="cuda:0", file_fmt="bmp",
                                 wind_size=64, overlap=32, multipass=2)():
        ...

Importing the API loads torchpiv_amd/libtorchpiv_hip.so and raises if it is missing
(build it with `python -c "import __graft_entry__ as g; g.build()"`): there is no CPU
fallback.  `torchpiv_amd.synth` (synthetic frames) is importable without the library.
"""
__version__ = "0.1.0"

_API = {
    "OfflinePIV", "DeviceMap", "IterModMap", "PIVDataset", "ToTensor", "natural_keys",
    "extended_search_area_piv", "piv_iteration_CWS", "piv_iteration_DWS",
    "get_field_shape", "get_coordinates", "moving_window_array",
    "interpolate_boarders", "fillMissingValues", "getPixelsForInterp", "nan_helper",
    "post_validate", "free_cuda_memory", "ResidentPIV", "fill_holes_host", "piv_iteration_CWS_Fast", "OnlinePIV",
}


def __getattr__(name):
    if name in _API:
        from . import backend
        return getattr(backend, name)
    if name in ("engine", "backend", "dist", "io", "synth", "_lib"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)


def __dir__():
    return sorted(_API | {"engine", "backend", "dist", "io", "synth"})
