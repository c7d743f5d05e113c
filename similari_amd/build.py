"""In-tree build of libsimilari_assoc.so (hipcc, gfx950 only). `python -m similari_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "lib" / "libsimilari_assoc.so"
SOURCES = ["sa_kernels.hip", "sa_gemm.hip", "sa_upkeep.hip", "sa_engine.hip", "sa_tracker.cpp", "sa_cluster.cpp"]
HEADERS = sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "similari_assoc.h", PKG.parent / "include" / "similari_tracker.h"]
# -ffp-contract=off: the reference (rustc) never fuses a*b+c; the bit-exact IoU / assignment gates rely on it.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
         "-Wno-unused-value", "-Wno-unused-result", "-pthread"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [CSRC / s for s in SOURCES] + HEADERS)


def build_lib(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    LIB.parent.mkdir(parents=True, exist_ok=True)
    extra = os.environ.get("SA_EXTRA_FLAGS", "").split()   # e.g. -DSA_GEMM_TRACE / -DSA_POS_TRACE (in-kernel timelines)
    cmd = [hipcc(), *FLAGS, *extra, *[str(CSRC / s) for s in SOURCES], "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
