"""The drop-in boundary as a library: libsimilari_assoc.so loads without a GPU and exports every function that
include/similari_assoc.h and include/similari_tracker.h declare; without a gfx950 device the product path fails loudly
(SA_ERR_NO_DEVICE) instead of falling back to any CPU code."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

from similari_amd import abi, build

ROOT = Path(__file__).resolve().parent.parent
DECL = re.compile(r"^\s*(?:const\s+)?(?:int|void|uint32_t|uint64_t|double|sa_engine\s*\*|const char\s*\*)\s*\*?\s*(sa_[a-z0-9_]+)\s*\(", re.M)


def declared(header: str):
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(DECL.findall(text)))


@pytest.fixture(scope="module")
def lib():
    return abi.load_library(build.build_lib())


@pytest.mark.parametrize("header", ["similari_assoc.h", "similari_tracker.h"])
def test_every_declared_function_is_exported(lib, header):
    names = declared(header)
    assert len(names) == (68 if header == "similari_assoc.h" else 21), names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"{header} declares functions the library does not export: {missing}"


def test_no_oracle_or_cpu_fallback_linked():
    so = build.build_lib()
    out = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True, check=True).stdout
    assert " or_" not in out, "the product library must not contain oracle symbols"
    deps = subprocess.run(["ldd", str(so)], capture_output=True, text=True).stdout
    assert "liboracle" not in deps


def test_struct_sizes_match_the_headers(lib):
    assert lib.sa_api_version() == 2
    assert C.sizeof(abi.sa_box) == 32 and abi.BOX_DTYPE.itemsize == 32
    cfg = abi.sa_config()
    lib.sa_config_default(C.byref(cfg))
    assert cfg.struct_size == C.sizeof(abi.sa_config)  # the C side wrote its own sizeof
    assert abs(cfg.positional_threshold - 0.3) < 1e-7 and abs(cfg.positional_min_confidence - 0.05) < 1e-7
    o = abi.sa_tracker_options()
    lib.sa_tracker_options_default(C.byref(o), 1)
    assert o.struct_size == C.sizeof(abi.sa_tracker_options)
    assert o.visual_max_observations == 5 and o.visual_minimal_track_length == 3 and o.max_idle_epochs == 2
    assert o.spin_us == -1 and o.n_devices == 0   # (0 would mean "no thread of the tracker ever polls": the defaults must ask for the facade's own times)


def gpu_visible() -> bool:
    """A ROCm GPU is present when its kernel driver node is (asking torch would start a second HIP context in this process)."""
    import os

    return os.path.exists("/dev/kfd")


def test_engine_refuses_to_run_without_a_gpu(lib):
    if gpu_visible():
        pytest.skip("a GPU is visible")
    cfg = abi.make_config()
    h = abi.ENGINE()
    rc = lib.sa_engine_create(C.byref(cfg), C.byref(h))
    assert rc == abi.SA_ERR_NO_DEVICE, rc
    assert b"no CPU fallback" in lib.sa_last_error(None)


def test_pinned_blocks_need_a_device_too(lib):
    """sa_host_alloc hands out pinned host memory for zero-staging uploads: NULL without a GPU (no silent malloc fallback), and
    sa_host_free(NULL) is a no-op."""
    p = lib.sa_host_alloc(4096)
    if gpu_visible():
        assert p
        lib.sa_host_free(p)
    else:
        assert not p
    lib.sa_host_free(None)


def test_cluster_refuses_to_run_without_a_gpu(lib):
    """The multi-GPU dispatcher is a router over engines: no device, no cluster (and no CPU fallback behind it either)."""
    if gpu_visible():
        pytest.skip("a GPU is visible")
    cfg = abi.make_config()
    h = C.c_void_p()
    rc = lib.sa_cluster_create(C.byref(cfg), 2, None, C.byref(h))
    assert rc == abi.SA_ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in lib.sa_cluster_last_error(None)
    lib.sa_cluster_destroy(None)


def test_device_block_registry_is_plain_bookkeeping(lib):
    """sa_device_block_register / _unregister keep a process-wide list of address ranges: no device call is made for an explicit
    device ordinal, bad arguments are refused, unregistering an unknown pointer is a no-op."""
    assert lib.sa_device_block_register(None, 4096, 0) == abi.SA_ERR_BAD_ARG
    assert lib.sa_device_block_register(C.c_void_p(0x7f0000000000), 0, 0) == abi.SA_ERR_BAD_ARG
    assert lib.sa_device_block_register(C.c_void_p(0x7f0000000000), 4096, 0) == abi.SA_OK
    assert lib.sa_device_block_register(C.c_void_p(0x7f0000000000), 8192, 0) == abi.SA_OK  # same base again: the size is updated
    lib.sa_device_block_unregister(C.c_void_p(0x7f0000000000))
    lib.sa_device_block_unregister(C.c_void_p(0x7f0000000000))
    lib.sa_device_block_unregister(None)


def test_every_prototype_of_the_python_binding_is_declared_in_a_header():
    """abi.PROTOTYPES (what tests and bench.py call) and the two headers name the same functions: nothing is bound that a C
    caller could not see, nothing declared is left unbound."""
    bound = set(abi.PROTOTYPES)
    decl = set(declared("similari_assoc.h")) | set(declared("similari_tracker.h"))
    assert bound == decl, (sorted(bound - decl), sorted(decl - bound))


@pytest.mark.parametrize("order", ["engine_first", "torch_first"])
def test_one_hip_runtime_in_the_process_whichever_is_loaded_first(order):
    """abi.load_library and PyTorch-ROCm end up on ONE libamdhip64 image in either order (abi.share_torchs_hip_runtime: a second runtime
    in the process would start without devices — "No HIP GPUs are available" — and could not read the first one's feature buffers)."""
    import subprocess
    import sys

    pytest.importorskip("torch")
    first, second = ("lib = abi.load_library()", "import torch") if order == "engine_first" else ("import torch", "lib = abi.load_library()")
    code = f"from similari_amd import abi\n{first}\n{second}\nm = abi.hip_runtimes_mapped()\nprint(len(m), m)"
    root = str(Path(__file__).resolve().parent.parent)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split()[0] == "1", r.stdout
