import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(ROOT / "tests") not in sys.path:
    sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "paths(*names): also run the test with each named path switch of the engine forced (tests/conftest.py)")


# ---- the engine's other paths under the same tests --------------------------------------------------------------------------
# @pytest.mark.paths("general", "never_lean", ...) runs a parity test once per listed path besides the default one: the `sa_path`
# fixture ORs the path's switch (a bit of sa_config.flags) into every config abi.make_config builds while the test runs.  Same
# process, same oracle — the switches are per engine, not per process.
import pytest  # noqa: E402


def _path_flags():
    from similari_amd import abi

    return {"default": 0, "general": abi.SA_FLAG_GENERAL_TAIL, "never_lean": abi.SA_FLAG_NEVER_LEAN, "bestfit_tile": abi.SA_FLAG_BESTFIT_TILE,
            "separate_resolve": abi.SA_FLAG_SEPARATE_RESOLVE, "euclid_valu": abi.SA_FLAG_EUCLID_VALU, "euclid_mfma": abi.SA_FLAG_EUCLID_MFMA,
            "row_tiles": abi.SA_FLAG_ROW_TILES, "xcd_tiles": abi.SA_FLAG_XCD_TILES, "signal_completion": abi.SA_FLAG_SIGNAL_COMPLETION,
            "staged_loop": abi.SA_FLAG_STAGED_LOOP, "no_yield": abi.SA_FLAG_NO_YIELD}


def pytest_generate_tests(metafunc):
    m = metafunc.definition.get_closest_marker("paths")
    if m and "sa_path" in metafunc.fixturenames:
        metafunc.parametrize("sa_path", ["default", *m.args], indirect=True)


@pytest.fixture(autouse=True)
def sa_path(request):
    from similari_amd import abi

    mode = getattr(request, "param", "default")
    old = abi.EXTRA_FLAGS
    abi.EXTRA_FLAGS = _path_flags()[mode]
    yield mode
    abi.EXTRA_FLAGS = old
