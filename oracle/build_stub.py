#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: compile the binding of INTEGRATION.md for real.

Takes the reference's tntblast_local.cpp where it lies (/root/reference), replaces -- in a scratch directory outside the
repository, deleted afterwards -- the `while(true){ ... }` work loop inside its `#pragma omp parallel` region
(tntblast_local.cpp:388-791) with oracle/integration_stub.inc (written here: tnt_ctx_create / tnt_db_create / tnt_search /
tnt_hit -> hybrid_sig), compiles that one file with the flags of oracle/Makefile and links it with the reference's other
objects (oracle/_ref/*.o, built by `make -C oracle ref`) and thermonucleotideblast_amd/libtnt_mi355x.so into
oracle/_ref/tntblast_stub (git-ignored; travels to the GPU box like the other built files).  Nothing of the reference is
copied into the repository.  tests/test_gpu_parity.py::test_integration_stub_drops_in runs it beside oracle/_ref/tntblast.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("TNT_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")
PKG = os.path.join(ROOT, "thermonucleotideblast_amd")


def main():
    src = os.path.join(REF, "tntblast_local.cpp")
    if not os.path.exists(src):
        print("no reference sources under %s: nothing to build" % REF)
        return 0
    text = open(src).read()
    lines = text.split("\n")
    # the work loop: from the first `while(true){` behind `#pragma omp parallel` to the line in front of the string-table merge
    par = next(i for i, l in enumerate(lines) if l.strip() == "#pragma omp parallel")
    start = next(i for i in range(par, len(lines)) if re.match(r"\s*while\(true\)\{\s*$", lines[i]))
    merge = next(i for i in range(start, len(lines)) if "// Merge the string tables" in lines[i])
    end = merge  # exclusive; the blank line in front of the comment goes too
    indent = len(lines[start]) - len(lines[start].lstrip("\t"))
    assert indent == 2, "unexpected layout of tntblast_local.cpp"
    # the loop's closing brace is the last line of that indentation in front of the merge
    close = max(i for i in range(start, end) if lines[i].rstrip() == "\t\t}")
    assert 380 <= start + 1 <= 395 and 785 <= close + 1 <= 800, (start + 1, close + 1)
    stub = open(os.path.join(HERE, "integration_stub.inc")).read()
    patched = lines[:start] + ["#include \"integration_stub.inc\""] + lines[close + 1:]
    head = "#include <map>\n#include \"tnt_mi355x.h\"\n"
    tmp = tempfile.mkdtemp(prefix="tnt_stub_")
    try:
        with open(os.path.join(tmp, "tntblast_local_stub.cpp"), "w") as f:
            f.write(head + "\n".join(patched))
        shutil.copy(os.path.join(HERE, "integration_stub.inc"), tmp)
        obj = os.path.join(tmp, "tntblast_local_stub.o")
        subprocess.run(["g++", "-O3", "-w", "-fopenmp", "-std=c++14", "-I" + REF, "-I" + os.path.join(ROOT, "include"), "-I" + tmp,
                        "-c", os.path.join(tmp, "tntblast_local_stub.cpp"), "-o", obj], check=True)
        # the reference binary's objects, tntblast_local.o replaced
        mk = open(os.path.join(REF, "Makefile")).read()
        objs = re.search(r"OBJS\s*=\s*((?:.*\\\n)*.*)", mk).group(1).replace("\\\n", " ").split()
        link = [os.path.join(OUT, o) for o in objs if o != "tntblast_local.o"]
        missing = [o for o in link if not os.path.exists(o)]
        if missing:
            print("missing reference objects (run `make -C oracle ref` first): %s" % " ".join(os.path.basename(m) for m in missing))
            return 1
        subprocess.run(["g++", "-O3", "-fopenmp", "-o", os.path.join(OUT, "tntblast_stub"), obj] + link +
                       ["-L" + PKG, "-ltnt_mi355x", "-Wl,-rpath," + PKG, "-Wl,-rpath,$ORIGIN/../../thermonucleotideblast_amd", "-lm", "-lz"], check=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("built %s (work loop %d-%d of tntblast_local.cpp replaced by %d lines of integration_stub.inc)" % (
        os.path.join(OUT, "tntblast_stub"), start + 1, close + 1, len(stub.split("\n"))))
    return 0


if __name__ == "__main__":
    sys.exit(main())
