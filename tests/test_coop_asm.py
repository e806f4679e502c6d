"""The cooperative kernel variant's machine code is what the forced-cooperative GPU suite last passed on.

tools/check_coop_asm.py counts, in the built library's cooperative kernel symbols, the workgroup barriers, the explicit
`s_waitcnt vmcnt(0)` drains and the agent-scope accesses its protocol is made of, and compares them with
profiles/coop_asm_golden.json. The protocol's publishes ("drain in every storing wave, barrier, one relaxed agent-scope
store") and the placement of its thread-0 sections between barriers are conventions of the source that neither the HIP
memory model nor the compiler promises to keep: after a compiler upgrade or an edit of the kernel this
test fails until tests/test_gpu_coop.py has been rerun on an MI355X and the golden file regenerated."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cooperative_kernel_symbols_match_the_verified_build():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_coop_asm.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_recorded_traffic_is_keyed_by_the_machine_code_it_was_measured_on():
    """profiles/pmc_traffic.json: every entry of the current round names the symbol its counter passes ran and that symbol's
    machine-code id (tools/kernel_code_id.py); bench.recorded_traffic() reports an entry only for that code (or that kernel
    source). The ids of the built library are well formed, and an entry whose id differs is refused with the reason."""
    import json
    import re
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import kernel_code_id
    # (r06: the timed variants are the streamed ones - a last template argument `true`)
    ids = kernel_code_id.code_ids(pattern=r"seismic_search_kernel<unsigned short, 512, 1, 1, false, [012], false, true>")
    assert len(ids) == 3 and all(re.fullmatch(r"[0-9a-f]{16}", v) for v in ids.values()), ids
    pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["workloads"]
    keyed = {k: e for k, e in pm.items() if e.get("symbol_code_id")}
    assert keyed, "no entry carries a machine-code id"
    for k, e in keyed.items():
        got, note = bench.recorded_traffic(k)
        have = bench.loaded_code_ids().get(e["symbol"].replace("seismic_search_kernel", ""))
        if e.get("kernel_source_id") == bench.kernel_source_id() or have == e["symbol_code_id"]:
            assert got == float(e["traffic_bytes"])
        else:   # the kernel was edited since: no figure, and the line says why
            assert got is None and e["symbol_code_id"] in note
