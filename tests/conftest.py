import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Results containers: the tests that look INTO a container read the .npz stand-in (hdf.load_npz), so the suite pins that type unless a
# test asks for "hdf5" itself (tests/test_hdf5_file.py: real HDF5 files, opened with the real h5py of another interpreter).  The
# product's own default is "hdf5" wherever an HDF5 library can be loaded (hdf.container_type).
os.environ.setdefault("GBP_CONTAINER", "npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def native_library():
    """The in-tree gfx950 library, built once if it is missing or older than its sources (hipcc cross-compiles without a GPU; the
    built .so is not in the history).  A no-op when it is up to date."""
    from geobipy_amd.build import build_native
    try:
        return build_native()
    except (RuntimeError, OSError, Exception) as e:      # (no hipcc here: the tests that need the library say so themselves)
        return None


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def golden_npz():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "fdem_golden.npz"))


# Parity tolerances (hard part 1): the reference itself is only reproducible to
# ~1.1e-8 ppm between its Numba build and the interpreter, so bit equality is not attainable.
PRED_ATOL, PRED_RTOL = 1e-7, 1e-9        # ppm
LIKE_ATOL, LIKE_RTOL = 1e-6, 1e-9        # chi^2, logL


def oracle_system(name):
    from oracle import fdem_oracle as fo
    return fo.OracleSystem.read(os.path.join(GOLDEN, f"{name}.stm"))


WEDGE_CONDUCTIVITY = {          # Model.create_synthetic_model conductivities (model/Model.py:902-908)
    "glacial": [1e-2, 1e-1, 0.03333333],
    "saline_clay": [1e-2, 1e-1, 1.0],
    "resistive_dolomites": [2e-2, 2e-3, 2e-2],
    "resistive_basement": [1e-2, 1e-1, 1e-4],
    "coastal_salt_water": [1.0, 1e-2, 5e-2],
    "ice_over_salt_water": [1e-4, 1e-2, 1.0],
}


def wedge_models():
    """79-sounding wedge of tests/test_synthetic_data.py:16-30 (interfaces / 10 for Resolve)."""
    import numpy as np
    zw = np.linspace(50.0, 1.0, 79) / 10.0
    zd = np.linspace(75.0, 500.0, 79) / 10.0
    thk = np.stack([zw, zd - zw, np.zeros(79)], axis=1)
    return thk


def read_clean_csv(path):
    import numpy as np
    return np.loadtxt(path, delimiter=",", skiprows=1)[:, 6:]
