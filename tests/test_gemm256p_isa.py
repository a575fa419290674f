"""The persistent 256 x 256 GEMM (csrc/gemm256.hpp, gemm256p_kernel) counts its vector-memory operations BY HAND: when a tile starts, a wave's
queue holds, oldest first, the 16 LDS-DMA requests of the next tile's first two K-tiles, the epilogue's 16 stores and one bias request, and
`s_waitcnt vmcnt(17)` is read as "the 16 requests have landed".  One store fewer than assumed and the wait lets a request pass that has not
landed: wrong tiles that come and go with timing.  hipcc decides how many instructions the epilogue's C++ becomes, so the count is checked
here, in the gfx950 ISA of every instantiation the library launches (hipcc cross-compiles without a GPU; ~20 s)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "meshanything_amd", "csrc")
VMEM = re.compile(r"^\s+(global_|buffer_|scratch_|flat_)(load|store|atomic)")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not available")
def test_persistent_gemm_epilogue_issues_exactly_the_counted_operations():
    src = '#include "%s/gemm256.hpp"\nusing namespace ma;\n' % CSRC
    src += "void inst(GemmTArgs g) {\n" + "".join(
        "    hipLaunchKernelGGL((gemm256p_kernel<%s, %d, 0>), dim3(8), dim3(512), G256P_LDS, 0, g, 1, 1, (unsigned long long*)nullptr);\n" % (ht, act)
        for ht in ("bf16_t", "f16_t") for act in (0, 1, 2)) + "".join(
        "    hipLaunchKernelGGL((gemm256p_kernel<%s, 0, 0, true>), dim3(8), dim3(512), G256P_LDS, 0, g, 1, 1, (unsigned long long*)nullptr);\n" % ht
        for ht in ("bf16_t", "f16_t")) + "}\n"
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "inst.hip")
        with open(f, "w") as fh:
            fh.write(src)
        out = os.path.join(d, "inst.s")
        subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-DNDEBUG", "-std=c++17", "-S", "--cuda-device-only", "-o", out, f], check=True, capture_output=True)
        text = open(out).read()
    kernels = re.findall(r"^(_ZN2ma15gemm256p_kernel\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
    assert len(kernels) == 8        # 2 formats x {3 activations, the K / V -> cache form}, [k for k, _ in kernels]
    for name, body in kernels:
        assert "scratch_" not in body, name + ": spills"
        ops = [ln.split()[0] for ln in body.splitlines() if VMEM.match(ln)]
        # the LAST run of 16 x global_load_lds_dwordx4 is the next tile's prefetch (the first run is the first tile's prologue)
        runs = [i for i in range(len(ops) - 15) if all(o == "global_load_lds_dwordx4" for o in ops[i:i + 16]) and (i + 16 == len(ops) or ops[i + 16] != "global_load_lds_dwordx4")
                and (i == 0 or ops[i - 1] != "global_load_lds_dwordx4")]
        assert len(runs) == 2, (name, runs, ops)
        tail = ops[runs[-1] + 16:]
        assert "global_load_lds_dword" in tail, (name, tail)
        between = tail[:tail.index("global_load_lds_dword")]
        assert between == ["global_store_dwordx4"] * 16, (name, between)
        # and the waits that lean on the count
        assert re.search(r"s_waitcnt vmcnt\(17\)", body) and re.search(r"s_waitcnt vmcnt\(1\)\n", body) and re.search(r"s_waitcnt vmcnt\(4\)", body), name
