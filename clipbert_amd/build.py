"""Build libclipbert_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the tree."""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libclipbert_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wno-unused-result"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def variant_path(variant: str) -> str:
    return os.path.join(LIB_DIR, f"libclipbert_hip_{variant}.so")


def build(force: bool = False, verbose: bool = False, variant: str = "", defines=(), csrc: str = "") -> str:
    """variant / defines: a DIAGNOSTIC copy of the library (e.g. variant="stamps", defines=("CB_STAMPS",): in-kernel time stamps,
    tools/stamps_run.py) in its own object directory; the product library is variant ""."""
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj" + (f"_{variant}" if variant else ""))
    os.makedirs(obj_dir, exist_ok=True)
    lib_path = variant_path(variant) if variant else LIB_PATH
    flags = FLAGS + [f"-D{d}" for d in defines]
    if csrc:                                       # a diagnostic copy built from ANOTHER source tree (e.g. a git worktree of an older commit)
        flags = [f if f != CSRC else csrc for f in flags]
    src_dir = csrc or CSRC
    headers = glob.glob(os.path.join(src_dir, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    jobs = []
    objs = []
    for src in sorted(glob.glob(os.path.join(src_dir, "*.hip"))):
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(lib_path, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs)
    return lib_path


if __name__ == "__main__":
    if "--variant" in sys.argv:                    # python -m clipbert_amd.build --variant NAME --csrc DIR
        i = sys.argv.index("--variant")
        defs = tuple(sys.argv[j + 1] for j, a in enumerate(sys.argv) if a == "--define")        # ... [--define MACRO]...
        print(build(verbose=True, variant=sys.argv[i + 1], defines=defs, csrc=sys.argv[sys.argv.index("--csrc") + 1] if "--csrc" in sys.argv else ""))
    elif "--stamps" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, variant="stamps", defines=("CB_STAMPS",)))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
