"""A/B builds: python tools/variant.py NAME SOURCE[,SOURCE..] [-DFLAG ...] compiles the named csrc/ files with the extra flags and links
them with the standard objects of disentangling-vae_amd/build/ into disentangling-vae_amd/lib/libdvae_hip_NAME.so (select it with
DVAE_HIP_LIB=...).  The standard library must have been built first (python disentangling-vae_amd/build.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "disentangling-vae_amd")
sys.path.insert(0, PKG)
import build as B  # noqa: E402


def main():
    name, srcs, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in B.SOURCES:
        o = os.path.join(B.OBJ, s + ".o")
        if s in srcs:
            o = os.path.join(B.OBJ, "variant_%s_%s.o" % (name, s))
            subprocess.run([hipcc] + B.FLAGS + flags + ["-c", os.path.join(B.SRC, s + ".hip"), "-o", o], check=True)
        objs.append(o)
    out = os.path.join(PKG, "lib", "libdvae_hip_%s.so" % name)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"], check=True)
    print(out)


if __name__ == "__main__":
    main()
