"""Build the gfx950 shared library in-tree (hipcc cross-compiles without a GPU).

Every source is compiled to its own object (in parallel, only when it or a header changed), then linked: a one-file edit
rebuilds in seconds instead of re-compiling all kernels."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsefd_hip.so")
SOURCES = ["api.hip", "kernels.hip", "rungemm.hip", "cgemm256.hip", "lstm_bf16.hip", "lstm_cluster.hip", "lstm_rows.hip", "bn.hip", "cbn.hip", "lms.hip", "pmsqe.hip", "mix.hip", "fsn.hip", "stft_fft.hip", "thin.hip", "plan.cpp"]


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "sefd.h"))
    return [h for h in hs if os.path.exists(h)]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    return _stale(LIB, [os.path.join(CSRC, s) for s in SOURCES] + _headers())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    heads = _headers()
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s + ".o")
        if force or _stale(obj, [src] + heads):
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-o", obj, src])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=int(os.environ.get("SEFD_BUILD_JOBS", "6"))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + [os.path.join(OBJ, s + ".o") for s in SOURCES])
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
