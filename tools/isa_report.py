"""Static report on the gfx950 code of one .hip source (no GPU needed): per kernel the register / LDS budget,
the instruction mix, and the ISSUE ORDER of its hot region as a compact string

    M = v_mfma   r = ds_read   W = ds_write   G = global/buffer load   S = global/buffer store
    w = s_waitcnt   | = s_barrier   . = other VALU   (SALU and the rest are dropped)

e.g. `rrr w MMMMMMMM rrr w MMMMMMMM` = operand reads one step ahead of 8 MFMAs.  This is the view that showed, in
round 1, selects on freshly loaded registers forcing `s_waitcnt vmcnt` in front of the MFMA phase, operand reads
serialised behind the MFMAs of the same step, and which `sched_group_barrier` hints the compiler ignored.

    python tools/isa_report.py disentangling-vae_amd/csrc/linear.hip [kernel-name-substring] [--full]
"""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only"]


def demangle(names):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            out = subprocess.run([tool] + names, capture_output=True, text=True, check=True).stdout.split("\n")
            return dict(zip(names, out))
        except Exception:
            continue
    return {n: n for n in names}


def classify(ins):
    if ins.startswith("v_mfma"): return "M"
    if ins.startswith("ds_read") or ins.startswith("ds_load"): return "r"
    if ins.startswith("ds_write") or ins.startswith("ds_store"): return "W"
    if re.match(r"(global|buffer|flat)_load", ins): return "G"
    if re.match(r"(global|buffer|flat)_store", ins): return "S"
    if ins.startswith("s_waitcnt"): return "w"
    if ins.startswith("s_barrier"): return "|"
    if ins.startswith("v_"): return "."
    return ""


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    full = "--full" in sys.argv
    src = args[0]
    want = args[1] if len(args) > 1 else ""
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC] + FLAGS + [src, "-o", asm], capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr)
        lines = open(asm).read().split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):", l)] if m]
    names = demangle([n for _, n in starts])
    meta = {}
    for i, l in enumerate(lines):                      # .amdhsa_ metadata blocks follow each kernel
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", l)
        if m:
            cur = meta.setdefault(m.group(1), {})
            for l2 in lines[i:i + 60]:
                m2 = re.match(r"\s*\.amdhsa_(next_free_vgpr|accum_offset|group_segment_fixed_size|next_free_sgpr)\s+(\S+)", l2)
                if m2:
                    cur[m2.group(1)] = m2.group(2)
    for idx, (i0, mangled) in enumerate(starts):
        name = re.sub(r"\(.*", "", names.get(mangled, mangled)).replace("void ", "").replace("dvae::", "")
        if want and want not in name:
            continue
        # the function's end label (a kernel may hold several s_endpgm: wave roles that return early); data symbols have none
        nxt_start = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
        i1 = next((i for i in range(i0, nxt_start) if lines[i].startswith(".Lfunc_end")), None)
        if i1 is None:
            continue
        body = [re.sub(r"\s*;.*$", "", l.strip()) for l in lines[i0 + 1:i1]]
        body = [l for l in body if l and not l.startswith(".") and not l.startswith(";") and not l.endswith(":")]
        seq = "".join(classify(l) for l in body)
        mix = {k: seq.count(k) for k in "MrWGSw|."}
        md = meta.get(mangled, {})
        print("== %s" % name)
        print("   vgpr+agpr %s (accum offset %s)  sgpr %s  static LDS %s B   instructions %d" % (
            md.get("next_free_vgpr", "?"), md.get("accum_offset", "?"), md.get("next_free_sgpr", "?"),
            md.get("group_segment_fixed_size", "?"), len(body)))
        print("   mix: mfma %(M)d  ds_read %(r)d  ds_write %(W)d  vmem load %(G)d  vmem store %(S)d  waitcnt %(w)d  barrier %(|)d  other valu %(.)d" % mix)
        if mix["M"]:
            a, b = seq.index("M"), seq.rindex("M")
            hot = seq[max(0, a - 24):b + 25]
            hot = re.sub(r"\.{4,}", lambda m: ".%d." % len(m.group(0)), hot)          # compress VALU runs
            if not full and len(hot) > 900:
                hot = hot[:600] + " ... " + hot[-280:]
            print("   issue order around the MFMAs:\n   " + hot)
        print()


if __name__ == "__main__":
    main()
