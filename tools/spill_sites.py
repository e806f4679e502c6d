#!/usr/bin/env python3
"""Where do a kernel symbol's scratch (spill) instructions sit? Disassembles one symbol of the built library and, for each
scratch_load / scratch_store, names the innermost loop around it (a backward branch's span) with what that loop contains:
a loop with workgroup barriers is a per-round / per-list / per-query loop, a loop with LDS byte reads and global loads and
no barrier is a scoring loop. usage: spill_sites.py "<unsigned short, 1024, 1, 1, false, 0, true>"  [library]"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_coop_asm as cca  # noqa: E402  (code_objects: the gfx950 code objects of the library's offload bundles)

want = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(cca.ROOT, "seismic_amd", "libseismic_hip.so")
import tempfile  # noqa: E402
with tempfile.TemporaryDirectory() as td:
    asm = None
    for k, co in enumerate(cca.code_objects(lib)):
        path = os.path.join(td, "co%d.o" % k)
        open(path, "wb").write(co)
        syms = subprocess.run([cca.LLVM + "/llvm-readelf", "-s", "-W", path], capture_output=True, text=True).stdout.split("\n")
        names = sorted({l.split()[-1] for l in syms if l.strip() and " FUNC " in l and "seismic_search_kernel" in l.split()[-1]})
        if not names:
            continue
        dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
        for m, d in zip(names, dem):
            if want in d:
                asm = subprocess.run([cca.LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + m, path],
                                     capture_output=True, text=True).stdout.split("\n")
                print(d)
if asm is None:
    sys.exit("no symbol matching %s in %s" % (want, lib))
ins = []
for i, l in enumerate(asm):
    m = re.match(r"\s+(\S+).*//\s*([0-9A-F]{12}):", l)
    if m:
        ins.append((i, int(m.group(2), 16), l))
at = {a: i for i, a, _ in ins}
loops = []
for i, a, l in ins:
    m = re.match(r"\s+s_c?branch\S*\s+(\d+)", l)
    if m:
        off = int(m.group(1))
        off -= 65536 if off >= 32768 else 0
        tgt = a + 4 + off * 4
        if tgt <= a and tgt in at:
            loops.append((at[tgt], i))
scr = [i for i, _, l in ins if "scratch_" in l]
print("%d instructions, %d loops, %d scratch instructions" % (len(ins), len(loops), len(scr)))
kinds = {}
for s in scr:
    inn = [(b, e) for b, e in loops if b <= s <= e]
    if not inn:
        kinds.setdefault("outside any loop (prologue / epilogue)", []).append(s)
        continue
    b, e = min(inn, key=lambda x: x[1] - x[0])
    body = asm[b:e + 1]
    nb, nu, ng = sum("s_barrier" in x for x in body), sum("ds_read_u8" in x for x in body), sum("global_load" in x for x in body)
    kind = ("loop with %d barriers (%d instructions)" % (nb, e - b)) if nb else \
           ("SCORING-LIKE loop: no barrier, %d LDS byte reads, %d global loads (%d instructions)" % (nu, ng, e - b)) if nu and ng else \
           ("small loop without barrier, %d LDS byte reads, %d global loads (%d instructions)" % (nu, ng, e - b))
    kinds.setdefault(kind, []).append(s)
for k, v in sorted(kinds.items(), key=lambda kv: -len(kv[1])):
    print("%3d  in %s" % (len(v), k))
