#!/usr/bin/env python3
"""Identity of a kernel symbol's MACHINE CODE in a built library: sha256 over the symbol's disassembled instructions
(mnemonics and operands, no addresses). A PMC traffic figure recorded in profiles/pmc_traffic.json belongs to the machine
code it was measured on; an edit elsewhere in search_kernel.inc (another template instantiation) changes the source id of
the file but not that code - bench.py accepts a recorded figure when either matches.
usage: kernel_code_id.py [library] [regex over demangled names]   -> one line per symbol"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_coop_asm as cca  # noqa: E402


def code_ids(lib=None, pattern=r"seismic_search_kernel<"):
    """{demangled symbol (template arguments only): 16 hex digits} for the search-kernel symbols matching `pattern`."""
    lib = lib or os.environ.get("SGPU_LIB") or os.path.join(cca.ROOT, "seismic_amd", "libseismic_hip.so")
    rx, out = re.compile(pattern), {}
    with tempfile.TemporaryDirectory() as td:
        for k, co in enumerate(cca.code_objects(lib)):
            path = os.path.join(td, "co%d.o" % k)
            open(path, "wb").write(co)
            syms = subprocess.run([cca.LLVM + "/llvm-readelf", "-s", "-W", path], capture_output=True, text=True).stdout.split("\n")
            names = sorted({l.split()[-1] for l in syms if l.strip() and " FUNC " in l and l.split()[-1].startswith("_ZN4sgpu21seismic_search_kernel")})
            if not names:
                continue
            dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
            pick = [(m, d) for m, d in zip(names, dem) if rx.search(d)]
            if not pick:
                continue
            asm = subprocess.run([cca.LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + ",".join(m for m, _ in pick), path],
                                 capture_output=True, text=True).stdout
            cur, h = None, {}
            for l in asm.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
                if m:
                    cur = m.group(1)
                    h[cur] = hashlib.sha256()
                elif cur and (l.startswith("\t") or l.startswith(" ")):
                    h[cur].update((l.split("//")[0].strip() + "\n").encode())
            for m, d in pick:
                if m in h:
                    out[re.search(r"seismic_search_kernel(<.*?>)\(", d).group(1)] = h[m].hexdigest()[:16]
    return out


if __name__ == "__main__":
    ids = code_ids(sys.argv[1] if len(sys.argv) > 1 else None, sys.argv[2] if len(sys.argv) > 2 else r"seismic_search_kernel<")
    for k in sorted(ids):
        print(ids[k], k)
