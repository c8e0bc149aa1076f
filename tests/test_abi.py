"""CPU tests: the C-ABI library loads and exports every symbol include/pcr.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pcr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcr_[a-z0-9_]+)\s*\(", text)) - {"pcr_allreduce_fn"})


def test_header_symbols_are_all_exported(pcr):
    if not os.path.exists(pcr.LIB_PATH):
        pytest.fail("libpcr_hip.so not built: run __graft_entry__.build()")
    L = pcr.lib()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/pcr.h but not exported"
    assert sorted(pcr.ABI_SYMBOLS) == syms


def test_host_logic_shard_range(pcr):
    for n in (0, 1, 7, 120000, 10_000_000):
        for nr in (1, 2, 3, 8):
            edges = [pcr.shard_range(n, nr, r) for r in range(nr)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1
            for (b0, e0), (b1, e1) in zip(edges, edges[1:]):
                assert e0 == b1


def test_host_kabsch_solve_matches_oracle(pcr, orc, synth):
    import numpy as np
    src, tgt = synth.kitti_like_pair(1500)
    idx, d2 = orc.nn1_f32(tgt, src)
    sums, _ = orc.kabsch_accumulate(src, tgt, idx, d2, 1.0)
    rc, R, t = pcr.kabsch_solve(sums)
    orc_rc, oR, ot = orc.kabsch_solve(sums)
    assert rc == orc_rc == 0
    assert np.array_equal(R, oR) and np.array_equal(t, ot)
    assert pcr.kabsch_solve(np.zeros(16))[0] != 0          # empty pair set is reported, not divided by zero


def test_no_gpu_means_loud_failure(pcr):
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    with pytest.raises(pcr.PcrError):
        pcr.Context(0)


def test_header_is_plain_c_and_links_from_c(pcr, tmp_path):
    """The boundary is a C ABI: include/pcr.h must compile as strict C11 (no C++-isms) and link from a C program."""
    import subprocess
    src = tmp_path / "abi_c.c"
    src.write_text('#include "pcr.h"\n#include <stdio.h>\n'
                   'int main(void) { pcr_iss_params p = {0}; pcr_icp_params q = {0}; (void)p; (void)q;\n'
                   '  double A[9] = {2,0,0, 0,1,0, 0,0,3}, n[3]; if (pcr_fast_eigen3x3(A, n) != PCR_OK) return 1;\n'
                   '  printf("%s %.0f %.0f %.0f\\n", pcr_version(), n[0], n[1], n[2]); return 0; }\n')
    libdir = os.path.dirname(pcr.LIB_PATH)
    exe = tmp_path / "abi_c"
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L" + libdir, "-lpcr_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("0 1 0"), r.stdout + r.stderr    # host logic only: runs without a GPU
