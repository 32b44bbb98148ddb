"""Compile the product's HIP sources against the host lane-level emulator (tests/emul/hip/hip_runtime.h)
into tests/emul/_build/libclipbert_emul.so.  TEST INFRASTRUCTURE ONLY -- see that header."""
import concurrent.futures
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "clipbert_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB_PATH = os.path.join(OUT_DIR, "libclipbert_emul.so")
CXX = os.environ.get("EMUL_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-O2", "-std=c++17", "-fPIC", "-x", "c++", "-I", HERE, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
         "-Wno-unused-value", "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-Wno-pass-failed"]


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(HERE, "emul_rt.cpp")]
    deps = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + \
        [os.path.join(HERE, "hip", "hip_runtime.h")]
    newest_dep = max(os.path.getmtime(d) for d in deps)
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OUT_DIR, os.path.basename(s).rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_dep):
            jobs.append([CXX] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emulator build failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB_PATH):
        run([CXX, "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-lpthread"])
    return LIB_PATH


if __name__ == "__main__":
    print(build())
