"""Build libhippomm_hip.so (gfx950) in-tree with hipcc.

    python -m hippomm_amd.build [--force] [--jobs N]

hipcc cross-compiles without a GPU.  The .so is written next to this file
(hippomm_amd/libhippomm_hip.so) so that it travels with the source tree to the
GPU box; objects go to build/obj.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
OBJ = ROOT / "build" / "obj"
LIB = PKG / "libhippomm_hip.so"
ARCH = "gfx950"

COMMON = ["-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
          f"--offload-arch={ARCH}", "-I", str(ROOT / "include")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (Path(cand).exists() or cand == "hipcc"):
            return cand
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _stale(src: Path, obj: Path) -> bool:
    if not obj.exists():
        return True
    newest_dep = max([src.stat().st_mtime] + [h.stat().st_mtime for h in CSRC.glob("*.h")]
                     + [h.stat().st_mtime for h in (ROOT / "tools" / "csrc").glob("*.h")]
                     + [h.stat().st_mtime for h in (ROOT / "include").glob("*.h")])
    return obj.stat().st_mtime < newest_dep


def _compile(src: Path, obj: Path, extra):
    cmd = [_hipcc(), *COMMON, *extra, "-x", "hip", "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def _build(srcs, obj_dir: Path, lib: Path, force: bool, jobs: int, extra) -> Path:
    obj_dir.mkdir(parents=True, exist_ok=True)
    todo, objs = [], []
    for src in srcs:
        obj = obj_dir / (src.stem + ".o")
        objs.append(obj)
        if force or _stale(src, obj):
            todo.append((src, obj))
    if todo:
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(lambda so: _compile(so[0], so[1], list(extra)), todo))
    if todo or not lib.exists():
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(lib)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


def build(force: bool = False, jobs: int = 4, extra=()) -> Path:
    """The product library: hippomm_amd/libhippomm_hip.so."""
    return _build(sources(), OBJ, LIB, force, jobs, extra)


PROBE_LIB = ROOT / "tools" / "libhippomm_probe.so"


def build_probe(force: bool = False, jobs: int = 4) -> Path:
    """The probe library used by tools/*.py only: the same sources compiled with -DHMM_PROBE (tuning knobs become
    run-time variables with hmm_probe_set_* setters, experiment kernels are instantiated) plus tools/csrc/*.hip
    (micro-benchmarks).  Never loaded by the package, the tests or bench.py."""
    srcs = sources() + sorted((ROOT / "tools" / "csrc").glob("*.hip"))
    return _build(srcs, ROOT / "build" / "obj_probe", PROBE_LIB, force, jobs,
                  ["-DHMM_PROBE=1", "-I", str(CSRC)])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--probe", action="store_true", help="also build tools/libhippomm_probe.so (-DHMM_PROBE)")
    a = ap.parse_args()
    print(build(a.force, a.jobs))
    if a.probe:
        print(build_probe(a.force, a.jobs))
