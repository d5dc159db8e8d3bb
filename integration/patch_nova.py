#!/usr/bin/env python3
"""Adds the `#elif defined(BACKEND_HIP)` branch to Nova's three backend dispatch headers.

    patch_nova.py --nova /path/to/Nova --out DIR     writes patched copies under DIR/src/backend/... (an include
                                                     overlay: put -I DIR/src before -I Nova/src), Nova untouched
    patch_nova.py --nova /path/to/Nova --in-place    edits the three headers of the tree itself

Only these three files name BACKEND_* under src/ (checked by the script).  Nothing else in Nova changes."""
import argparse
import os
import re
import sys

BRANCHES = {
    "src/backend/accelerator.hpp": ['#include "backend/hip/accelerator/accelerator.hpp"'],
    "src/backend/types.hpp": ['#include "backend/common/types/types.hpp"', '#include "backend/hip/types/types.hpp"'],
    "src/backend/common/utils/utils.hpp": ['#include "backend/hip/types/vector_types.hpp"'],
}
CUDA_BRANCH = re.compile(r"(#elif defined\(BACKEND_CUDA\)\n(?:[ \t]+#include [^\n]+\n)+)")


def patched(text: str, includes) -> str:
    if "BACKEND_HIP" in text:
        return text
    m = CUDA_BRANCH.search(text)
    if not m:
        raise SystemExit("dispatch block not found: has this header changed upstream?")
    branch = "#elif defined(BACKEND_HIP)\n" + "".join("  %s\n" % i for i in includes)
    return text[: m.end()] + branch + text[m.end():]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nova", required=True)
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("--out")
    g.add_argument("--in-place", action="store_true")
    a = ap.parse_args()
    named = []
    for dp, _, fs in os.walk(os.path.join(a.nova, "src")):
        for f in fs:
            p = os.path.join(dp, f)
            try:
                if re.search(r"BACKEND_(OPENCL|CUDA)", open(p, errors="replace").read()):
                    named.append(os.path.relpath(p, a.nova))
            except OSError:
                pass
    extra = sorted(set(named) - set(BRANCHES))
    if extra:
        print("note: BACKEND_* is also named in", extra, file=sys.stderr)
    for rel, inc in BRANCHES.items():
        text = patched(open(os.path.join(a.nova, rel)).read(), inc)
        dst = os.path.join(a.nova if a.in_place else a.out, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as f:
            f.write(text)
        print("wrote", dst)


if __name__ == "__main__":
    main()
