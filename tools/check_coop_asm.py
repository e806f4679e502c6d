#!/usr/bin/env python3
"""Tripwire for the cooperative kernel variant's machine code (VERDICT r03 "weak" #3).

The cooperative protocol (search_kernel.inc: coop_work / wide_round / coop_help) orders its publishes with
`s_waitcnt vmcnt(0)` in every storing wave + a workgroup barrier + ONE relaxed agent-scope store, and its source keeps
every thread-0 section between two barriers because ROCm 7.2 was seen fusing adjacent ones and breaking the pairing of
barriers. Neither is something the HIP memory model or the compiler promises to keep. This tool disassembles the
cooperative kernel symbols of the BUILT library and reports, per symbol: s_barrier count, explicit vmcnt(0) drains,
agent-scope (sc1) stores and loads, atomics, scratch use. `--check` compares with profiles/coop_asm_golden.json and fails on any
difference: a compiler upgrade or an edit of the kernel then fails a CPU test (tests/test_coop_asm.py) instead of a launch,
and the golden file is only regenerated (`--update`) after the forced-cooperative GPU suite has passed on the new build.

  python tools/check_coop_asm.py            # print
  python tools/check_coop_asm.py --check    # compare with the golden file
  python tools/check_coop_asm.py --update   # rewrite the golden file (after tests/test_gpu_coop.py is green)
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
GOLD = os.path.join(ROOT, "profiles", "coop_asm_golden.json")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
# the variants the product launches for the benchmark shapes: u16 components, dense lookup, k <= 64, f16 / fixed-u8 /
# DotVByte values, 512- and 1024-thread workgroups, cooperative
# (r06: the kernel template has one more parameter, STREAM, always false for the cooperative variants; the golden file keeps
# the r05 names)
WANT = re.compile(r"seismic_search_kernel<unsigned short, (512|1024), 1, 1, false, [012], true(?:, false)?>")


def code_objects(lib):
    """The gfx950 code objects of every offload bundle in the library's .hip_fatbin section."""
    data = open(lib, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(data[i + off:i + off + size])
        pos = i + len(MAGIC)
    return out


def analyse(lib):
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for k, co in enumerate(code_objects(lib)):
            path = os.path.join(td, "co%d.o" % k)
            open(path, "wb").write(co)
            syms = subprocess.run([LLVM + "/llvm-readelf", "-s", "-W", path], capture_output=True, text=True).stdout.split("\n")
            names = sorted({l.split()[-1] for l in syms if l.strip() and " FUNC " in l and l.split()[-1].startswith("_ZN4sgpu21seismic_search_kernel")})
            if not names:
                continue
            dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
            for m, d in zip(names, dem):
                if not WANT.search(d):
                    continue
                asm = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + m, path],
                                     capture_output=True, text=True).stdout
                ins = [l.split("//")[0].strip() for l in asm.split("\n") if l.startswith("\t") or l.startswith(" ")]
                ins = [i for i in ins if i]
                key = WANT.search(d).group(0).replace("true, false>", "true>")
                res[key] = {
                    "instructions": len(ins),
                    "s_barrier": sum(i.startswith("s_barrier") for i in ins),
                    "waitcnt_vmcnt0": sum(bool(re.match(r"s_waitcnt\b.*\bvmcnt\(0\)", i)) for i in ins),
                    "sc1_stores": sum(i.startswith("global_store") and " sc1" in i for i in ins),
                    "global_atomics": sum(i.startswith("global_atomic") for i in ins),   # (RMWs execute at the L2: no sc1 marker)
                    "sc1_loads": sum(i.startswith("global_load") and " sc1" in i for i in ins),
                    "scratch_ops": sum(i.startswith("scratch_") for i in ins),
                }
    return res


def main():
    lib = os.environ.get("SGPU_LIB") or os.path.join(ROOT, "seismic_amd", "libseismic_hip.so")
    res = analyse(lib)
    if not res:
        print("no cooperative kernel symbol found in %s" % lib)
        return 2
    if "--update" in sys.argv:
        ver = subprocess.run([LLVM + "/clang", "--version"], capture_output=True, text=True).stdout.split("\n")[0]
        json.dump({"compiler": ver, "kernels": res}, open(GOLD, "w"), indent=1, sort_keys=True)
        print("wrote", GOLD)
        return 0
    if "--check" in sys.argv:
        gold = json.load(open(GOLD))["kernels"]
        bad = [(k, f, gold.get(k, {}).get(f), v[f]) for k, v in res.items() for f in v
               if f != "instructions" and gold.get(k, {}).get(f) != v[f]]
        missing = [k for k in gold if k not in res]
        for b in bad:
            print("DIFFERS  %s  %s: golden %s, built %s" % b)
        for k in missing:
            print("MISSING  %s" % k)
        if bad or missing:
            print("the cooperative variant's machine code changed: rerun `pytest tests/test_gpu_coop.py -m gpu` on an MI355X, "
                  "then `python tools/check_coop_asm.py --update`")
            return 1
        print("cooperative kernel symbols match the golden counts (%d symbols)" % len(res))
        return 0
    print(json.dumps(res, indent=1, sort_keys=True))
    return 0


if __name__ == "__main__":
    sys.exit(main())
