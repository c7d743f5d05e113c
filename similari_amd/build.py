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
    """One object per source (compiled side by side, kept under similari_amd/lib/obj and reused while neither the source nor any
    header is newer), then one link."""
    if not force and not needs_build():
        return LIB
    LIB.parent.mkdir(parents=True, exist_ok=True)
    obj_dir = LIB.parent / "obj"
    obj_dir.mkdir(exist_ok=True)
    extra = os.environ.get("SA_EXTRA_FLAGS", "").split()   # e.g. -DSA_GEMM_TRACE / -DSA_POS_TRACE (in-kernel timelines)
    cflags = [f for f in FLAGS if f != "-shared"]
    stamp = obj_dir / "flags.txt"
    flag_text = " ".join(cflags + extra)
    if force or not stamp.exists() or stamp.read_text() != flag_text:
        for o in obj_dir.glob("*.o"):
            o.unlink()
        stamp.write_text(flag_text)
    newest_header = max(p.stat().st_mtime for p in HEADERS)
    jobs = []
    objs = []
    for src in SOURCES:
        o = obj_dir / (src.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if o.exists() and o.stat().st_mtime > max((CSRC / src).stat().st_mtime, newest_header):
            continue
        cmd = [hipcc(), *cflags, *extra, "-c", str(CSRC / src), "-o", str(o)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((src, subprocess.Popen(cmd, cwd=str(CSRC))))
    failed = [src for src, p in jobs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed on " + ", ".join(failed))
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", *[str(o) for o in objs], "-o", str(LIB)]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.run(link, check=True, cwd=str(CSRC))
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
