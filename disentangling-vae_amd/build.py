"""Build libdvae_hip.so (gfx950) in-tree: hipcc cross-compiles without a GPU.

    python disentangling-vae_amd/build.py [--force] [--debug]

The .so lands in disentangling-vae_amd/lib/ (git-ignored, but it travels with gpurun
snapshots).  Objects are rebuilt only when a source / header is newer or the flags changed.
--debug (or DVAE_BUILD_DEBUG=1) adds -DDVAE_DEBUG_SWITCHES: the A/B / timing-ablation environment
switches and the experimental kernel variants they select (tools/README.md); the default (shipped)
library has none of them.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libdvae_hip.so")
HEADERS = [os.path.join(SRC, "common.h"), os.path.join(SRC, "conv_mfma_common.h"), os.path.join(SRC, "wgrad_reduce.h"),
           os.path.join(HERE, "..", "include", "dvae_hip.h")]
SOURCES = ["conv_generic", "conv_mfma", "conv_down_dma", "conv_up_ws", "conv_wgrad_ws", "conv_thin", "conv_thin_ws", "conv_up_thin_mm", "linear", "linear_narrow", "gemm_dma", "linear_grouped", "fc_chain", "stage", "loss", "latent_wide",
           "metrics", "adam", "comm", "plan", "capi"]
DEBUG_SOURCES = []          # experimental kernel files: only in --debug builds
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True, debug=None):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if debug is None:
        debug = os.environ.get("DVAE_BUILD_DEBUG", "0") == "1"
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    flags = FLAGS + (["-DDVAE_DEBUG_SWITCHES"] if debug else [])
    sources = SOURCES + (DEBUG_SOURCES if debug else [])
    stamp = os.path.join(OBJ, "flags.txt")
    want = " ".join(flags + sources)
    if not os.path.exists(stamp) or open(stamp).read() != want:
        force = True
    jobs = []
    for s in sources:
        src, obj = os.path.join(SRC, s + ".hip"), os.path.join(OBJ, s + ".o")
        if force or _newer(src, obj) or any(_newer(h, obj) for h in HEADERS):
            jobs.append([hipcc] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=8) as ex:
        for err in ex.map(run, jobs):
            if verbose and err.strip():
                sys.stderr.write(err)
    objs = [os.path.join(OBJ, s + ".o") for s in sources]
    if force or jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    with open(stamp, "w") as f:
        f.write(want)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, debug=True if "--debug" in sys.argv else None))
