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
