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
SOURCES = ["api.hip", "kernels.hip", "rungemm.hip", "cgemm256.hip", "enc0.hip", "lstm_bf16.hip", "lstm_cluster.hip", "lstm_rows.hip", "bn.hip", "cbn.hip", "lms.hip", "pmsqe.hip", "mix.hip", "fsn.hip", "stft_fft.hip", "thin.hip", "plan.cpp", "tuning.cpp"]


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "sefd.h"))
    return [h for h in hs if os.path.exists(h)]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_digest(paths, extra="") -> str:
    """sha256 over the CONTENTS of `paths` (sorted by name) + `extra` (compiler flags).  "Is the built library current?" is decided by this
    digest, stored beside the library as <lib>.stamp, not by modification times: a snapshot of the tree on another machine (gpurun, the
    driver's GPU box) has arbitrary mtimes, and a library that merely LOOKS older than its sources was rebuilt there for 90 seconds."""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def stamp_current(lib, digest) -> bool:
    try:
        return os.path.exists(lib) and open(lib + ".stamp").read().strip() == digest
    except OSError:
        return False


def write_stamp(lib, digest):
    with open(lib + ".stamp", "w") as f:
        f.write(digest + "\n")


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _digest():
    return source_digest([os.path.join(CSRC, s) for s in SOURCES] + _headers(), " ".join(FLAGS + SOURCES))


def needs_build() -> bool:
    return not stamp_current(LIB, _digest())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    # one builder at a time across processes (two ranks of a test that both find the library stale compiled into the same object files and broke each
    # other's translation units): the second one waits, then finds the stamp current
    import fcntl
    with open(os.path.join(OBJ, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    heads = _headers()
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s + ".o")
        if force or _stale(obj, [src] + heads):
            jobs.append([hipcc] + FLAGS + ["-c", "-o", obj, src])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=int(os.environ.get("SEFD_BUILD_JOBS", "6"))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + [os.path.join(OBJ, s + ".o") for s in SOURCES])
    write_stamp(LIB, _digest())
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
