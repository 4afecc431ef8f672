#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch figures from a `hipcc -save-temps` .s file (gfx950).

usage: tools/isa_report.py path/to/*-gfx950.s [substring ...]
Exit code 1 when any kernel reserves private (scratch) memory: every dispatch of such a
kernel sets up scratch even if no instruction touches it.
"""
import re
import subprocess
import sys


def kernels(text):
    out = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(2)

        def g(k):
            mm = re.search(r"\.amdhsa_" + k + r" (\S+)", body)
            return int(mm.group(1)) if mm else 0

        out.append({"mangled": m.group(1), "vgpr": g("next_free_vgpr"), "agpr_offset": g("accum_offset"),
                    "sgpr": g("next_free_sgpr"), "lds": g("group_segment_fixed_size"),
                    "scratch": g("private_segment_fixed_size")})
    names = subprocess.run(["c++filt"], input="\n".join(k["mangled"] for k in out), capture_output=True,
                           text=True).stdout.split("\n")
    for k, n in zip(out, names):
        k["name"] = n
    return out


def main():
    ks = kernels(open(sys.argv[1]).read())
    want = sys.argv[2:]
    bad = 0
    for k in ks:
        if want and not any(w in k["name"] for w in want):
            continue
        print(f'{k["vgpr"]:4d} vgpr {k["sgpr"]:4d} sgpr {k["lds"]:6d} lds {k["scratch"]:4d} scratch  {k["name"][:150]}')
    for k in ks:
        if k["scratch"]:
            bad = 1
            print("SCRATCH:", k["name"][:200], file=sys.stderr)
    print(f"{len(ks)} kernels", file=sys.stderr)
    return bad


if __name__ == "__main__":
    sys.exit(main())
